#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/r04
nproc; free -g | head -2
timeout 300 python scripts/parity_census.py --nsamples 300000 --max-pairs 60 --out gpurun_out/r04/census_small.json 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_native_batch.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "weighted_full_size" 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 8 > gpurun_out/r04/emu8_native6.json 2> gpurun_out/r04/emu8_native6.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_native6.json 2> gpurun_out/r04/bench_native6.err
python - <<'PY'
import json
for f in ("bench_native6","emu8_native6"):
    try:
        d=json.loads(open("gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1]); print(f, "ms_per_step=%.2f"%d["ms_per_step"], d.get("ms_single_triangle_latency"))
    except Exception as e: print(f, "failed", e)
PY
