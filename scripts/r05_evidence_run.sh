#!/bin/bash
# The evidence run of round 5: one GPU-box session, every number of DESIGN.md section 5 from one box.
#   gpurun --timeout 2400 -- 'bash scripts/r05_evidence_run.sh'
# Outputs under gpurun_out/r05/ ; the summaries are copied to profiles/r05_* by scripts/r05_collect_profiles.sh.
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_session.sh r05 tests benchfull bench trace conv pmcconv c2 c4 pmcc4 kernels emu_c3 emu_c4 emu_c5 emu8trace f64roof rccl gloo2
