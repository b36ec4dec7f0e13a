#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/r04
GDHIP_BATCH_LOG=1 timeout 300 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --emulate-world 8 > gpurun_out/r04/hostlog_emu8.json 2> gpurun_out/r04/hostlog_emu8.err
python - <<'PY'
txt=open("gpurun_out/r04/hostlog_emu8.err").read().split("---- gd_density2d_batch host timeline (ms)\n")
print(len(txt)-1,"calls")
t=txt[5] if len(txt)>5 else txt[-1]
print(t[:4000])
PY
GETDIST_AMD_HOSTLOG=1 timeout 300 python scripts/host_timeline.py 8 2>&1 | grep " ms " | tail -30
