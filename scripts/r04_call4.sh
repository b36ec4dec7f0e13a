#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 300 python scripts/r04_dct_check.py 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_native_batch.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_native2.json 2> gpurun_out/r04/bench_native2.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 8 > gpurun_out/r04/emu8_native2.json 2> gpurun_out/r04/emu8_native2.err
python - <<'PY'
import json
for f in ("bench_native2","emu8_native2"):
    d=json.loads(open("gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1]); print(f, "ms_per_step=%.2f"%d["ms_per_step"], d.get("ms_single_triangle_latency"))
PY
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_native2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_native2.log 2>&1); echo "prof rc=$?"
f=$(find gpurun_out/r04/prof_native2 -name "*kernel_trace.csv" | head -1)
python scripts/stream_timeline.py $f 5 0.15 > gpurun_out/r04/timeline_native2.txt; head -70 gpurun_out/r04/timeline_native2.txt
