#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_native_batch.py -q 2>&1 | grep -E "^E  |passed|failed" | cut -c1-600 | head -12
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_native5.json 2> gpurun_out/r04/bench_native5.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-world 8 > gpurun_out/r04/emu8_native5.json 2> gpurun_out/r04/emu8_native5.err
python - <<'PY'
import json
for f in ("bench_native5","emu8_native5"):
    try:
        d=json.loads(open("gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1]); print(f, "ms_per_step=%.2f"%d["ms_per_step"], d.get("ms_single_triangle_latency"))
    except Exception as e: print(f, "failed", e)
PY
tail -5 gpurun_out/r04/bench_native5.err
GDHIP_BATCH_LOG=1 timeout 300 python bench.py --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/r04/hostlog3.json 2> gpurun_out/r04/hostlog3.err
python - <<'PY'
txt=open("gpurun_out/r04/hostlog3.err").read().split("---- gd_density2d_batch host timeline (ms)\n")
print(len(txt)-1,"calls")
t=txt[6] if len(txt)>6 else txt[-1]
print("\n".join(l for l in t.splitlines() if "conv:" not in l)[:3000])
PY
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_native5 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_native5.log 2>&1); echo "prof rc=$?"
f=$(find gpurun_out/r04/prof_native5 -name "*kernel_trace.csv" | head -1)
python scripts/stream_timeline.py $f 5 0.25 > gpurun_out/r04/timeline_native5.txt; head -64 gpurun_out/r04/timeline_native5.txt
