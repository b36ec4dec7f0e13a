#!/bin/bash
# evidence run of round 4 (what profiles/r04_* were made by): full GPU suite, smoke, the driver's bench line with live PMC
# traffic and CPU baseline, native vs Python-planned route on one box, emulated ranks, kernel traces of every config, PMC
# passes, isolated kernels.  Usage on the GPU box: bash scripts/r04_evidence_run.sh [quick]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04e
export TMPDIR=/tmp
O=gpurun_out/r04e
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 $O/pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 40 --warmup 5 --cpu-budget-s 200 > $O/bench_run.log 2>&1; echo "bench rc=$?"
grep "^{" $O/bench_run.log | tail -1 > $O/r04_bench.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/r04e/r04_bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'latency', d['ms_single_triangle_latency'], 'roof', d['roofline']['ms_per_launch'], d['roofline']['frac'], d['roofline']['traffic_source'][:50], 'cpu', d['cpu_baseline']['value'])
print(json.dumps({k:v for k,v in d['cpu_baseline']['full_triangle_small_n']['parity_census'].items() if k!='per_class'}))
print(json.dumps({k:v for k,v in d['parity'].items() if k not in ('classes','note')})[:1200])
PY
timeout 300 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
GETDIST_AMD_LIVE_PMC=0 GETDIST_AMD_NATIVE_BATCH=0 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_python_planned.json 2> $O/bench_python.err
GETDIST_AMD_LIVE_PMC=0 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/r04_bench_native.json 2> $O/bench_native.err
python - <<'PY'
import json, subprocess, sys, os
out = {}
for f in ("r04_bench_default", "r04_bench_python_planned", "r04_bench_native"):
    d = json.loads(open("gpurun_out/r04e/%s.json" % f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["ms_single_triangle_latency"])
env = dict(os.environ, GETDIST_AMD_LIVE_PMC="0")
for W in (2, 4, 8):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--emulate-world", str(W)], capture_output=True, text=True, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    out[str(W)] = dict(ms_per_step=d["ms_per_step"], value_if_all_ranks_took_this_long=d["value"], ms_between_step_returns=d["ms_between_step_returns"])
    print("W", W, d["ms_per_step"])
json.dump(out, open("gpurun_out/r04e/r04_emulate_world.json", "w"), indent=1)
PY
GDHIP_BATCH_LOG=1 GETDIST_AMD_HOSTLOG=1 timeout 300 python scripts/host_timeline.py 8 > $O/host_timeline_w8.out 2> $O/host_timeline_w8.err
GDHIP_BATCH_LOG=1 GETDIST_AMD_HOSTLOG=1 timeout 300 python scripts/host_timeline.py > $O/host_timeline_w1.out 2> $O/host_timeline_w1.err
timeout 600 python scripts/r04_kernels.py > $O/kernels_run.log 2>&1; echo "kernels rc=$?"; cp gpurun_out/r04_kernels.json $O/ 2>/dev/null
# kernel traces: the bench (C3) and the other configs
export GETDIST_AMD_LIVE_PMC=0
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof bench rc=$?"
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
python scripts/stream_timeline.py $(find $O/prof_bench -name "*kernel_trace.csv" | head -1) 4 0.1 > $O/r04_bench_stream_timeline.txt 2>&1
for c in c2 c4 c5; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$c -o $c -- python $GRAFT_REPO_ROOT/scripts/run_configs.py $c > $GRAFT_REPO_ROOT/$O/prof_$c.log 2>&1); echo "prof $c rc=$?"
  cp $(find $O/prof_$c -name "*kernel_stats.csv" | head -1) $O/r04_${c}_kernel_stats.csv
  grep "^{" $O/prof_$c.log | tail -1 > $O/r04_config_$c.json
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${c}_$ctr -o h -- python $GRAFT_REPO_ROOT/scripts/run_configs.py $c > /dev/null 2>&1)
  done
done
# PMC passes of the binning kernels and of the optimiser
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_hist_$ctr -o h -- python $GRAFT_REPO_ROOT/scripts/pmc_hist2d.py > /dev/null 2>&1)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_kopt_$ctr -o h -- python $GRAFT_REPO_ROOT/scripts/tune_kopt.py > /dev/null 2>&1)
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_hist_SQ -o h -- python $GRAFT_REPO_ROOT/scripts/pmc_hist2d.py > /dev/null 2>&1)
python scripts/summarise_pmc_r04.py $O > $O/pmc_summary.log 2>&1; tail -30 $O/pmc_summary.log
ls $O | head -60
