#!/bin/bash
# refresh of the step-level evidence after the 16 x 18 register transforms of the convolution (the kernel-level files of
# scripts/r04_evidence_run2.sh that this change does not touch stay): full GPU suite, the driver's bench line, default bench,
# A/B against the radix-pass column kernel, emulated ranks, the bench's kernel trace + stream timeline, the convolution
# micro-benchmark's kernel trace.  Usage on the GPU box: bash scripts/r04_evidence_run3.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 40 --warmup 5 --cpu-budget-s 120 > $O/bench_run.log 2>&1; echo "bench rc=$?"
grep "^{" $O/bench_run.log | tail -1 > $O/r04_bench.json
export GETDIST_AMD_LIVE_PMC=0
timeout 300 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err
GETDIST_AMD_NATIVE_BATCH=0 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_python_planned.json 2>/dev/null
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_native.json 2>/dev/null
GDHIP_KOPT_STREAMED=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_streamed_fixed_point.json 2>/dev/null
GDHIP_CONV_RADIX_PASSES=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_radix_pass_columns.json 2>/dev/null
python - <<'PY'
import json, subprocess, sys
out = {}
for f in ("r04_bench", "r04_bench_default", "r04_bench_python_planned", "r04_bench_native", "r04_bench_streamed_fixed_point", "r04_bench_radix_pass_columns"):
    d = json.loads(open("gpurun_out/r04g/%s.json" % f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d["ms_single_triangle_latency"], d.get("roofline", {}).get("frac"))
for W in (2, 4, 8):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--emulate-world", str(W)], capture_output=True, text=True)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out[str(W)] = dict(ms_per_step=d["ms_per_step"], value_if_all_ranks_took_this_long=d["value"], ms_between_step_returns=d["ms_between_step_returns"])
    print("W", W, d["ms_per_step"])
json.dump(out, open("gpurun_out/r04g/r04_emulate_world.json", "w"), indent=1)
PY
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
python scripts/stream_timeline.py $(find $O/prof_bench -name "*kernel_trace.csv" | head -1) 4 0.1 > $O/r04_bench_stream_timeline.txt 2>&1
rm -rf $O/prof_bench
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_conv -o conv -- python $GRAFT_REPO_ROOT/scripts/r04_conv_bench.py > $GRAFT_REPO_ROOT/$O/conv_bench.log 2>&1)
cp $(find $O/prof_conv -name "*kernel_stats.csv" | head -1) $O/r04_conv_bench_kernel_stats.csv
rm -rf $O/prof_conv
GDHIP_CONV_RADIX_PASSES=1 timeout 100 python scripts/r04_conv_bench.py >> $O/conv_bench.log 2>&1
cat $O/conv_bench.log | grep density2d
cp gpurun_out/r04_parity_2d.json gpurun_out/r04_configs.json $O/ 2>/dev/null
head -12 $O/r04_bench_kernel_stats.csv | cut -c1-50,120-220
