#!/bin/bash
# round 4, GPU call 1: the native batch entry against the Python-planned route, then the bench both ways
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_native_batch.py -x -q > gpurun_out/r04/native_tests.log 2>&1
echo "native tests rc=$?" | tee -a gpurun_out/r04/native_tests.log
tail -5 gpurun_out/r04/native_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_native.json 2> gpurun_out/r04/bench_native.err
echo "bench native rc=$?"
GETDIST_AMD_NATIVE_BATCH=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_python.json 2> gpurun_out/r04/bench_python.err
echo "bench python rc=$?"
for W in 8 4 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world $W > gpurun_out/r04/emu${W}_native.json 2> gpurun_out/r04/emu${W}_native.err
done
GETDIST_AMD_NATIVE_BATCH=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 8 > gpurun_out/r04/emu8_python.json 2> gpurun_out/r04/emu8_python.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "ms_per_step=%.2f"%d["ms_per_step"], "lat=%s"%d.get("ms_single_triangle_latency"), "roof=%s"%(d.get("roofline") or {}).get("ms_per_launch"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r04/*.err
