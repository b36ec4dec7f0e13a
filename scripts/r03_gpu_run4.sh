#!/bin/bash
# fourth GPU session: full GPU test suite on the new quantile / division / fused-convolution kernels, bench, traces
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest4.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r03_pytest4.log | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench4.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/r03_bench4.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'latency', d['ms_single_triangle_latency'], 'roof ms', d['roofline']['ms_per_launch'])"
for W in 8; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-world $W > gpurun_out/r03_emu4_$W.log 2>&1
  grep "^{" gpurun_out/r03_emu4_$W.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W', d['emulated_world'], 'ms_per_step', d['ms_per_step'])"
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench4 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench4.log 2>&1); echo "prof rc=$?"
K=$(find gpurun_out/prof_bench4 -name "*kernel_stats.csv" | head -1); head -45 "$K" | cut -c1-180
