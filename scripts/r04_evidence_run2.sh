#!/bin/bash
# evidence run of round 4 on the final tree (after the resident fixed-point kernel): full GPU suite, smoke, the driver's
# bench line (live PMC traffic, CPU baseline), default bench, Python-planned route and streamed fixed-point kernel on the same
# box, emulated ranks, host timelines, isolated kernels, the bench's kernel trace and per-stream timeline, counter passes of
# the optimiser, the C5-shaped triangle's kernel trace.  Usage on the GPU box: bash scripts/r04_evidence_run2.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 40 --warmup 5 --cpu-budget-s 150 > $O/bench_run.log 2>&1; echo "bench rc=$?"
grep "^{" $O/bench_run.log | tail -1 > $O/r04_bench.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/r04f/r04_bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'latency', d['ms_single_triangle_latency'], 'roof', d['roofline']['ms_per_launch'], d['roofline']['frac'], d['roofline']['traffic_source'][:50], 'cpu', d['cpu_baseline']['value'])
print(json.dumps({k:v for k,v in d['cpu_baseline']['full_triangle_small_n']['parity_census'].items() if k!='per_class'})[:1000])
print(json.dumps({k:v for k,v in d['parity'].items() if k not in ('classes','note')})[:1200])
PY
export GETDIST_AMD_LIVE_PMC=0
timeout 300 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
GETDIST_AMD_NATIVE_BATCH=0 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_python_planned.json 2> $O/bench_python.err
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_native.json 2> $O/bench_native.err
GDHIP_KOPT_STREAMED=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_streamed_fixed_point.json 2> $O/bench_streamed.err
python - <<'PY'
import json, subprocess, sys, os
out = {}
for f in ("r04_bench_default", "r04_bench_python_planned", "r04_bench_native", "r04_bench_streamed_fixed_point"):
    d = json.loads(open("gpurun_out/r04f/%s.json" % f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["ms_single_triangle_latency"])
for W in (2, 4, 8):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--emulate-world", str(W)], capture_output=True, text=True)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out[str(W)] = dict(ms_per_step=d["ms_per_step"], value_if_all_ranks_took_this_long=d["value"], ms_between_step_returns=d["ms_between_step_returns"])
    print("W", W, d["ms_per_step"])
json.dump(out, open("gpurun_out/r04f/r04_emulate_world.json", "w"), indent=1)
PY
GDHIP_BATCH_LOG=1 GETDIST_AMD_HOSTLOG=1 timeout 300 python scripts/host_timeline.py 8 > $O/r04_emu8_timeline.txt 2> $O/host_timeline_w8.err
GDHIP_BATCH_LOG=1 GETDIST_AMD_HOSTLOG=1 timeout 300 python scripts/host_timeline.py > $O/r04_step_timeline.txt 2> $O/host_timeline_w1.err
timeout 600 python scripts/r04_kernels.py > $O/kernels_run.log 2>&1; echo "kernels rc=$?"; cp gpurun_out/r04_kernels.json $O/ 2>/dev/null
timeout 200 python scripts/r04_kopt_resident.py > $O/r04_kopt_resident_ab.json 2> $O/kopt_ab.err
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof bench rc=$?"
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
python scripts/stream_timeline.py $(find $O/prof_bench -name "*kernel_trace.csv" | head -1) 4 0.1 > $O/r04_bench_stream_timeline.txt 2>&1
rm -rf $O/prof_bench
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/scripts/run_configs.py c5 > $GRAFT_REPO_ROOT/$O/prof_c5.log 2>&1); echo "prof c5 rc=$?"
cp $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) $O/r04_c5_kernel_stats.csv
grep "^{" $O/prof_c5.log | tail -1 > $O/r04_config_c5.json
rm -rf $O/prof_c5
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_kopt_$ctr -o h -- python $GRAFT_REPO_ROOT/scripts/tune_kopt.py > /dev/null 2>&1)
done
python scripts/summarise_pmc_r04.py $O > $O/pmc_summary.log 2>&1; tail -12 $O/pmc_summary.log
rm -rf $O/pmc_kopt_FETCH_SIZE/*/*.db 2>/dev/null
cp gpurun_out/r04_parity_2d.json gpurun_out/r04_configs.json $O/ 2>/dev/null
ls $O
