#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_densities.py tests/test_gpu_edge_cases.py tests/test_gpu_mutators.py -m gpu -q -x > gpurun_out/r03_pytest5.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r03_pytest5.log | cut -c1-300
timeout 600 python scripts/r03_kernels.py > gpurun_out/r03_kernels5.log 2>&1; echo "kernels rc=$?"
grep "^isolated" gpurun_out/r03_kernels5.log | cut -c1-1800
for v in 0 1; do
  GETDIST_AMD_BIN_FIRST=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench5_binfirst$v.log 2>&1
  grep "^{" gpurun_out/r03_bench5_binfirst$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BIN_FIRST=$v ms_per_step', d['ms_per_step'], 'latency', d['ms_single_triangle_latency'])"
done
timeout 300 python scripts/host_timeline.py > gpurun_out/r03_host_timeline.txt 2>&1; tail -25 gpurun_out/r03_host_timeline.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-world 8 > gpurun_out/r03_emu5_8.log 2>&1
grep "^{" gpurun_out/r03_emu5_8.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W', d['emulated_world'], 'ms_per_step', d['ms_per_step'])"
GETDIST_AMD_TIMING=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --emulate-world 8 > gpurun_out/r03_emu5_8_timing.log 2>&1
grep "^{" gpurun_out/r03_emu5_8_timing.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('phase_seconds_total')))"
