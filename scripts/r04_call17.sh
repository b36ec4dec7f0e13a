#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/r04; export GETDIST_AMD_LIVE_PMC=0
timeout 300 python -m pytest tests/test_gpu_native_batch.py -x -q -k "c3_shape or equals_python" 2>&1 | tail -2
GDHIP_CONV_TAIL_AUX=1 timeout 300 python -m pytest tests/test_gpu_native_batch.py -x -q -k "c3_shape" 2>&1 | tail -2
for v in a b a b; do
 if [ $v = a ]; then unset GDHIP_CONV_TAIL_AUX; else export GDHIP_CONV_TAIL_AUX=1; fi
 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_aux_$v.json 2> gpurun_out/r04/bench_aux_$v.err
 python - <<PY
import json
d=json.loads(open("gpurun_out/r04/bench_aux_$v.json").read().strip().splitlines()[-1]); print("$v", "ms_per_step=%.2f"%d["ms_per_step"], d.get("ms_single_triangle_latency"))
PY
done
