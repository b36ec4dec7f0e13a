#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for K in 6 20 40; do
timeout 300 python bench.py --steps $K --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench12_$K.log 2>&1
grep "^{" gpurun_out/r03_bench12_$K.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K', d['steps'], d['ms_per_step'], d['ms_single_triangle_latency'], d['ms_between_step_returns'])"
done
