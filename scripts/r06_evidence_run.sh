#!/bin/bash
# The evidence session of round 6: ONE box, every number of DESIGN.md section 5 / profiles/r06_*.
#   gpurun --timeout 3400 -- 'bash scripts/r06_evidence_run.sh'      (then: bash scripts/r06_collect_profiles.sh here)
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_session.sh r06 tests benchfull bench trace latency kernels pmcprep covab conv c2 c4 c5 emulate emu_c4 rccl gloo2
