#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench16.log 2>&1
grep "^{" gpurun_out/r03_bench16.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('LDS  W1', d['ms_per_step'], d['ms_single_triangle_latency'], d['ms_between_step_returns'])"
GDHIP_CONV_ROCFFT=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench16r.log 2>&1
grep "^{" gpurun_out/r03_bench16r.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rocFFT W1', d['ms_per_step'], d['ms_single_triangle_latency'])"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench16 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench16.log 2>&1); echo "prof rc=$?"
head -30 $(find gpurun_out/prof_bench16 -name "*kernel_stats.csv" | head -1) | cut -c1-150
