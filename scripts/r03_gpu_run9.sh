#!/bin/bash
# host-side picture of a step: HIP runtime API trace + kernel trace (no counters), W = 1 and rank 0 of 8
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/api_w1 -o w1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/api_w1.log 2>&1); echo "api w1 rc=$?"
(cd /tmp && timeout 400 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/api_w8 -o w8 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --emulate-world 8 > $GRAFT_REPO_ROOT/gpurun_out/api_w8.log 2>&1); echo "api w8 rc=$?"
ls -la gpurun_out/api_w1 gpurun_out/api_w8
timeout 300 python scripts/host_timeline.py > gpurun_out/r03_host_timeline_w1.txt 2>&1; grep " ms " gpurun_out/r03_host_timeline_w1.txt | tail -30
