#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_densities.py tests/test_gpu_mutators.py -m gpu -q -x > gpurun_out/r03_pytest6.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r03_pytest6.log | cut -c1-300
timeout 600 python scripts/r03_kernels.py > gpurun_out/r03_kernels6.log 2>&1; echo "kernels rc=$?"
grep "^isolated" gpurun_out/r03_kernels6.log | cut -c1-1800
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench6.log 2>&1
grep "^{" gpurun_out/r03_bench6.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'latency', d['ms_single_triangle_latency'])"
timeout 300 python scripts/host_timeline.py 2>/dev/null | tail -14
for W in 2 4 8; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-world $W > gpurun_out/r03_emu6_$W.log 2>&1
grep "^{" gpurun_out/r03_emu6_$W.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W', d['emulated_world'], 'ms_per_step', d['ms_per_step'])"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_emu8b -o emu8 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --emulate-world 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_emu8b.log 2>&1); echo "prof rc=$?"
K=$(find gpurun_out/prof_emu8b -name "*kernel_stats.csv" | head -1); head -28 "$K" | cut -c1-150
