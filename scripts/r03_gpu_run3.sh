#!/bin/bash
# third GPU session: covariance fix check, emulated 8-rank timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/r03_kernels.py > gpurun_out/r03_kernels3.log 2>&1; echo "kernels rc=$?"
grep "^cov" gpurun_out/r03_kernels3.log | cut -c1-330
timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_primitives.py -m gpu -q -x -k "cov or gelman or c4" > gpurun_out/r03_pytest3.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest3.log
for W in 2 4 8; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-world $W > gpurun_out/r03_emu_$W.log 2>&1; echo "emu $W rc=$?"
  grep "^{" gpurun_out/r03_emu_$W.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W', d['emulated_world'], 'ms_per_step', d['ms_per_step'])"
done
GETDIST_AMD_TIMING=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --emulate-world 8 > gpurun_out/r03_emu_8_timing.log 2>&1; echo "emu timing rc=$?"
grep "^{" gpurun_out/r03_emu_8_timing.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('phase_seconds_total'),indent=0))"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_emu8 -o emu8 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --emulate-world 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_emu8.log 2>&1); echo "prof rc=$?"
K=$(find gpurun_out/prof_emu8 -name "*kernel_stats.csv" | head -1); head -40 "$K" | cut -c1-200
