#!/bin/bash
# evidence run of round 3 (what profiles/r03_* were made by): full GPU suite, bench line with CPU baseline, kernel trace, emulated ranks with timeline, isolated kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r03_pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 40 --warmup 5 --cpu-budget-s 200 > gpurun_out/r03_bench_run.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/r03_bench_run.log | tail -1 > gpurun_out/r03_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'latency', d['ms_single_triangle_latency'], 'roof', d['roofline']['ms_per_launch'], d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['single_process_value'])
print(d['ms_between_step_returns'])
print(json.dumps({k:v for k,v in d['cpu_baseline']['full_triangle_small_n']['parity_census'].items() if k!='per_class'}))
print(json.dumps({k:v for k,v in d['parity'].items() if k not in ('classes','note')})[:1500])"
timeout 600 python scripts/r03_kernels.py > gpurun_out/r03_kernels_run.log 2>&1; echo "kernels rc=$?"
python - <<'PY'
import json, subprocess, sys
out = {}
for W in (2, 4, 8):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--emulate-world", str(W)], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    out[str(W)] = dict(ms_per_step=d["ms_per_step"], value_if_all_ranks_took_this_long=d["value"], ms_between_step_returns=d["ms_between_step_returns"])
    print("W", W, d["ms_per_step"])
json.dump(out, open("gpurun_out/r03_emulate_world.json", "w"), indent=1)
PY
timeout 300 python scripts/host_timeline.py 8 2>&1 | grep " ms " > gpurun_out/r03_host_timeline_w8.txt; cat gpurun_out/r03_host_timeline_w8.txt
timeout 300 python scripts/host_timeline.py 2>&1 | grep " ms " > gpurun_out/r03_host_timeline_w1.txt; cat gpurun_out/r03_host_timeline_w1.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1); echo "prof rc=$?"
cp $(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/r03_bench_kernel_stats.csv
cp $(find gpurun_out/prof_bench -name "*domain_stats.csv" | head -1) gpurun_out/r03_bench_domain_stats.csv 2>/dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_emu8 -o emu8 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --emulate-world 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_emu8.log 2>&1); echo "prof8 rc=$?"
head -12 gpurun_out/r03_bench_kernel_stats.csv | cut -c1-160
