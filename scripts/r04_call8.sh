#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for parts in 2 3 4 12; do
  GDHIP_KOPT_PARTS=$parts timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_parts$parts.json 2> gpurun_out/r04/bench_parts$parts.err
done
python - <<'PY'
import json
for f in ("bench_parts2","bench_parts3","bench_parts4","bench_parts12"):
    try:
        d=json.loads(open("gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1]); print(f, "ms_per_step=%.2f"%d["ms_per_step"], d.get("ms_single_triangle_latency"))
    except Exception as e: print(f, "failed", e)
PY
