cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04j; mkdir -p $O; export TMPDIR=/tmp GETDIST_AMD_LIVE_PMC=0
timeout 300 python bench.py > $O/r04_bench_default.json 2>/dev/null
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_native.json 2>/dev/null
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
python scripts/stream_timeline.py $(find $O/prof_bench -name "*kernel_trace.csv" | head -1) 4 0.1 > $O/r04_bench_stream_timeline.txt 2>&1
rm -rf $O/prof_bench
python - <<'PY'
import json
for f in ("r04_bench_default","r04_bench_native"):
    d=json.loads(open("gpurun_out/r04j/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d["ms_single_triangle_latency"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"])
PY
head -3 $O/r04_bench_stream_timeline.txt
