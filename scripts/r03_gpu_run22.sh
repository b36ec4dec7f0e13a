#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest22.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r03_pytest22.log | cut -c1-300
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench22.log 2>&1
grep "^{" gpurun_out/r03_bench22.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W1', d['ms_per_step'], d['ms_single_triangle_latency'], d['roofline']['ms_per_launch'])"
for W in 8 4 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --emulate-world $W > gpurun_out/r03_emu22_$W.log 2>&1
  grep "^{" gpurun_out/r03_emu22_$W.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W', d['n_gpus'], d['ms_per_step'])"
done
