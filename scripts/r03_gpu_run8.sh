#!/bin/bash
# per-rank floor work: split parameter exchange (N_eff beside the binning), two-stream convolution of a rank's share,
# partial-table reduction with the chunks' loads in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_primitives.py -m gpu -q -x -k "c3_full_shape or hist2d or packed or upscaled or chunk or shear" > gpurun_out/r03_pytest8.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r03_pytest8.log | cut -c1-300
for W in 8 4 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-world $W > gpurun_out/r03_emu8_$W.log 2>&1
  grep "^{" gpurun_out/r03_emu8_$W.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W', d['n_gpus'], d['ms_per_step'])"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench8.log 2>&1
grep "^{" gpurun_out/r03_bench8.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W1', d['ms_per_step'], d['roofline']['ms_per_launch'])"
timeout 300 python scripts/host_timeline.py 8 > gpurun_out/r03_host_timeline_w8.txt 2>&1; tail -30 gpurun_out/r03_host_timeline_w8.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_emu8c -o emu8 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --emulate-world 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_emu8c.log 2>&1); echo "prof rc=$?"
