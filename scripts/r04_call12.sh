#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_native_batch.py tests/test_gpu_rccl_smoke.py -x -q 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_native7_$i.json 2> gpurun_out/r04/bench_native7_$i.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 8 > gpurun_out/r04/emu8_native7.json 2> gpurun_out/r04/emu8_native7.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 4 > gpurun_out/r04/emu4_native7.json 2> gpurun_out/r04/emu4_native7.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 2 > gpurun_out/r04/emu2_native7.json 2> gpurun_out/r04/emu2_native7.err
python - <<'PY'
import json
for f in ("bench_native7_1","bench_native7_2","emu8_native7","emu4_native7","emu2_native7"):
    try:
        d=json.loads(open("gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1]); print(f, "ms_per_step=%.2f"%d["ms_per_step"], d.get("ms_single_triangle_latency"), (d.get("roofline") or {}).get("traffic_source","")[:60], (d.get("roofline") or {}).get("frac"))
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/r04/bench_native7_1.err
