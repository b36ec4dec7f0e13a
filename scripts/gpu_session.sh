#!/bin/bash
# One parameterised wrapper for the GPU-box sessions of a round:  gpurun -- 'bash scripts/gpu_session.sh <tag> <stage>...'
# Stages write under gpurun_out/<tag>/ (merged back by gpurun); the summaries that are judged are copied to profiles/ by hand.
#   tests        pytest -m gpu (whole suite)            tests:<expr>  pytest -m gpu -k <expr>
#   conv         scripts/conv_bench.py under rocprofv3 --kernel-trace --stats (one 136-pair batch of the convolution stage)
#   bench        bench.py --steps 40 --warmup 5 --no-cpu-baseline      benchfull   bench.py (the driver's invocation)
#   trace        rocprofv3 kernel trace + stream timeline of 4 bench steps
#   c2 / c4 / c5 scripts/run_configs.py <config> (+ kernel trace for c4)
#   kernels      scripts/kernels_isolated.py (isolated O(N) kernels)
#   pmcconv      scripts/pmc_conv.sh (counter passes of the convolution kernels)
#   emulate      bench.py --emulate-world 2/4/8 (+ C5 / C4 forms when present)
#   pmcprep      scripts/pmc_kernels.py over the O(N) preparation kernels of a C3 step      latency   kernel + copy trace of a delivered triangle
#   covab        scripts/cov_ab.py (one-pass against two-pass covariance)                   ab:ENV=V  same-box A/B of an environment switch
cd "$GRAFT_REPO_ROOT" || exit 1
TAG="$1"; shift
O=gpurun_out/$TAG; mkdir -p "$O"; export TMPDIR=/tmp
prof() {  # prof <name> <cmd...>: kernel trace + stats of a command, the stats csv kept as $O/<name>_kernel_stats.csv
    local name="$1"; shift
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o "$name" -- "$@" > "$GRAFT_REPO_ROOT/$O/$name.log" 2>&1)
    cp "$(find "$O/prof_$name" -name '*kernel_stats.csv' | head -1)" "$O/${name}_kernel_stats.csv" 2>/dev/null
}
for stage in "$@"; do
    echo "==== stage $stage"
    case "$stage" in
        tests) timeout 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; tail -5 "$O/pytest.log" ;;
        tests:*) timeout 900 python -m pytest tests -m gpu -x -q -k "${stage#tests:}" > "$O/pytest_k.log" 2>&1; tail -5 "$O/pytest_k.log" ;;
        conv)
            prof conv python "$GRAFT_REPO_ROOT/scripts/conv_bench.py" 8
            grep "density2d" "$O/conv.log"; rm -rf "$O/prof_conv"
            python - "$O/conv_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.4: print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
            ;;
        bench) GETDIST_AMD_LIVE_PMC=0 timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > "$O/bench_40.json" 2> "$O/bench_40.err"
               python -c "import json,sys; d=json.loads(open('$O/bench_40.json').read().strip().splitlines()[-1]); print('delivered triangle ms', d['ms_per_step'], 'value', d['value'], 'pipelined ms', d['ms_per_step_pipelined'], 'roofline', d['roofline']['frac'], d['roofline']['ms_per_launch'])" ;;
        benchfull) timeout 900 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
               python -c "import json,sys; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], 'parity loose', d.get('parity',{}).get('n_pairs_on_loose_gate'))" ;;
        trace)
            prof bench python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline
            python scripts/stream_timeline.py "$(find "$O/prof_bench" -name '*kernel_trace.csv' | head -1)" 4 0.1 > "$O/bench_stream_timeline.txt" 2>&1
            rm -rf "$O/prof_bench"; head -30 "$O/bench_kernel_stats.csv" | cut -c1-150 ;;
        c2|c5) timeout 900 python scripts/run_configs.py "$stage" > "$O/config_$stage.json" 2> "$O/config_$stage.err"; tail -c 600 "$O/config_$stage.json" ;;
        c4)
            prof c4 python "$GRAFT_REPO_ROOT/scripts/run_configs.py" c4
            tail -3 "$O/c4.log" | cut -c1-400; rm -rf "$O/prof_c4"; head -8 "$O/c4_kernel_stats.csv" | cut -c1-170 ;;
        kernels) timeout 900 python scripts/kernels_isolated.py > "$O/kernels.log" 2>&1; cp gpurun_out/kernels_isolated.json "$O/kernels.json" 2>/dev/null; tail -3 "$O/kernels.log" ;;
        pmcconv) timeout 900 python scripts/pmc_kernels.py "$O/pmc_conv.json" -- python "$GRAFT_REPO_ROOT/scripts/conv_bench.py" 4 2>&1 | tail -16 ;;
        pmcc4) timeout 900 python scripts/pmc_kernels.py "$O/pmc_c4.json" --match k_cov,k_col_pass -- python "$GRAFT_REPO_ROOT/scripts/run_configs.py" c4 2>&1 | tail -6 ;;
        emulate)
            for W in 2 4 8; do
                GETDIST_AMD_LIVE_PMC=0 timeout 300 python bench.py --steps 30 --warmup 5 --emulate-world $W --no-cpu-baseline > "$O/emulate_w$W.json" 2> "$O/emulate_w$W.err"
                python -c "import json; d=json.loads(open('$O/emulate_w$W.json').read().strip().splitlines()[-1]); print('W', $W, 'ms_per_step', d['ms_per_step'])"
            done ;;
        emu_c3) GETDIST_AMD_LIVE_PMC=0 timeout 600 python scripts/emulate_scaling.py --nparams 50 --nsamples 10000000 > "$O/emulate_c3.json" 2> "$O/emulate_c3.err"; cat "$O/emulate_c3.err" | grep "^W=" ;;
        emu_c3_class) GETDIST_AMD_PAIR_DEAL=class GETDIST_AMD_LIVE_PMC=0 timeout 600 python scripts/emulate_scaling.py --nparams 50 --nsamples 10000000 --worlds 1,8 > "$O/emulate_c3_class.json" 2> "$O/emulate_c3_class.err"; cat "$O/emulate_c3_class.err" | grep "^W=" ;;
        emu_c5) GETDIST_AMD_LIVE_PMC=0 timeout 900 python scripts/emulate_scaling.py --nparams 200 --nsamples 2000000 --steps 8 --warmup 3 > "$O/emulate_c5.json" 2> "$O/emulate_c5.err"; cat "$O/emulate_c5.err" | grep "^W=" ;;
        emu_c4) for W in 1 2 4 8; do timeout 300 python scripts/gelman_rubin_multi_gpu.py --emulate-world $W > "$O/emulate_c4_w$W.json" 2> "$O/emulate_c4_w$W.err"; tail -1 "$O/emulate_c4_w$W.json" | cut -c1-300; done ;;
        emu8trace)
            (cd /tmp && GDHIP_BATCH_LOG=1 GETDIST_AMD_HOSTLOG=1 GETDIST_AMD_LIVE_PMC=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_emu8" -o emu8 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 3 --emulate-world 8 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$O/emu8.log" 2> "$GRAFT_REPO_ROOT/$O/emu8.err")
            python scripts/stream_timeline.py "$(find "$O/prof_emu8" -name '*kernel_trace.csv' | head -1)" 6 0.02 > "$O/emu8_stream_timeline.txt" 2>&1
            cp "$(find "$O/prof_emu8" -name '*kernel_stats.csv' | head -1)" "$O/emu8_kernel_stats.csv"; rm -rf "$O/prof_emu8"
            head -120 "$O/emu8_stream_timeline.txt" | cut -c1-120; grep -v WARNING "$O/emu8.err" | tail -60 | cut -c1-120 ;;
        f64roof) (cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o lds_atomic_f64_roof "$GRAFT_REPO_ROOT/scripts/micro/lds_atomic_f64_roof.hip" && ./lds_atomic_f64_roof 400) > "$O/lds_atomic_f64_roof.txt" 2>&1; cat "$O/lds_atomic_f64_roof.txt" ;;
        pmcw) timeout 900 python scripts/pmc_kernels.py "$O/pmc_weighted.json" --match k_wpart,k_hist2d -- python "$GRAFT_REPO_ROOT/scripts/weighted_binning.py" 2>&1 | tail -8
              grep -E "conflict|valu" "$O/pmc_weighted.json" | head -20 ;;
        gloo2) GETDIST_AMD_LIVE_PMC=0 timeout 600 python bench.py --gpus 2 --backend gloo --share-device --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_gloo2_shared_gpu.json" 2> "$O/bench_gloo2.err"; tail -c 900 "$O/bench_gloo2_shared_gpu.json"; tail -3 "$O/bench_gloo2.err" | cut -c1-300 ;;
        pmcprep)  # counter traffic of the O(N) preparation kernels of a C3 step (review item: <= 18 GB per step)
              timeout 1500 python scripts/pmc_kernels.py "$O/pmc_prep.json" --match k_col_,k_cov_,k_qlin,k_qsel,k_autocov,k_kde_lag,k_prebin,k_bucket,k_sum_partials -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -14 ;;
        latency)  # one delivered triangle on the time axis: pipeline stages and result copies
              (cd /tmp && GETDIST_AMD_LIVE_PMC=0 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lat_$TAG -o lat -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$O/lat.log" 2>&1)
              python scripts/latency_timeline.py /tmp/lat_$TAG 5 > "$O/latency_timeline.txt" 2>&1; head -14 "$O/latency_timeline.txt" ;;
        covab) timeout 600 python scripts/cov_ab.py > "$O/cov_ab.json" 2> "$O/cov_ab.err"; grep -E "N[0-9]|ms_" "$O/cov_ab.json" ;;
        ab:*)  # same-box A/B of one environment switch: ab:NAME=VALUE  (three alternating pairs of 30-step runs)
              SW="${stage#ab:}"
              for rep in 1 2 3; do
                  for side in base "$SW"; do
                      if [ "$side" = base ]; then envs="GDAMD_AB=base"; else envs="$SW"; fi
                      env $envs GETDIST_AMD_LIVE_PMC=0 timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > "$O/ab_tmp.json" 2> "$O/ab_tmp.err"
                      python -c "import json; d=json.loads(open('$O/ab_tmp.json').read().strip().splitlines()[-1]); print('$side', 'delivered', round(d['ms_per_step'],3), 'pipelined', round(d['ms_per_step_pipelined'],3))" | tee -a "$O/ab_${SW%%=*}.txt"
                  done
              done ;;
        rccl) timeout 300 python scripts/nccl_smoke.py > "$O/rccl_smoke.log" 2>&1; tail -5 "$O/rccl_smoke.log" ;;
        *) echo "unknown stage $stage" ;;
    esac
done
