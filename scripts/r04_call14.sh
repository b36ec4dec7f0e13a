#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_native_batch.py -x -q 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_native8_$i.json 2> gpurun_out/r04/bench_native8_$i.err
done
GDHIP_CONV_MOMENT_ARRAYS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_native8_arrays.json 2> gpurun_out/r04/bench_native8_arrays.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 8 > gpurun_out/r04/emu8_native8.json 2> gpurun_out/r04/emu8_native8.err
python - <<'PY'
import json
for f in ("bench_native8_1","bench_native8_2","bench_native8_arrays","emu8_native8"):
    try:
        d=json.loads(open("gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1]); print(f, "ms_per_step=%.2f"%d["ms_per_step"], d.get("ms_single_triangle_latency"))
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/r04/bench_native8_1.err
