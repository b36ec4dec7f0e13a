#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_native_batch.py -x -q 2>&1 | tail -5
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_native -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_native.log 2>&1); echo "prof rc=$?"
f=$(find gpurun_out/r04/prof_native -name "*kernel_trace.csv" | head -1)
python scripts/stream_timeline.py $f 5 0.08 > gpurun_out/r04/timeline_native.txt; cat gpurun_out/r04/timeline_native.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_emu8 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --emulate-world 8 > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_emu8.log 2>&1); echo "prof rc=$?"
f=$(find gpurun_out/r04/prof_emu8 -name "*kernel_trace.csv" | head -1)
python scripts/stream_timeline.py $f 5 0.03 > gpurun_out/r04/timeline_emu8.txt; cat gpurun_out/r04/timeline_emu8.txt
