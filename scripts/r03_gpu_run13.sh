#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_densities.py tests/test_gpu_edge_cases.py tests/test_gpu_mutators.py -m gpu -q -x > gpurun_out/r03_pytest13.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03_pytest13.log | cut -c1-300
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench13.log 2>&1
grep "^{" gpurun_out/r03_bench13.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W1', d['ms_per_step'], d['ms_single_triangle_latency'], d['roofline']['ms_per_launch'], d['ms_between_step_returns'])"
for W in 8 4 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --emulate-world $W > gpurun_out/r03_emu13_$W.log 2>&1
  grep "^{" gpurun_out/r03_emu13_$W.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W', d['n_gpus'], d['ms_per_step'], d['ms_between_step_returns'])"
done
(cd /tmp && timeout 400 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/api13_w1 -o w1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/api13_w1.log 2>&1); echo "api w1 rc=$?"
