#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_native_batch.py -x -q -k weighted 2>&1 | grep -E "^E|passed|failed" | cut -c1-900 | head -20
GDHIP_BATCH_LOG=1 timeout 300 python bench.py --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/r04/hostlog.json 2> gpurun_out/r04/hostlog.err
python - <<'PY'
txt=open("gpurun_out/r04/hostlog.err").read().split("---- gd_density2d_batch host timeline (ms)\n")
print(len(txt)-1,"calls")
print(txt[6][:6000])
PY
