cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06e; mkdir -p $O
run() { tag=$1; shift; env "$@" GETDIST_AMD_LIVE_PMC=0 timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/ab_$tag.json 2> $O/ab_$tag.err; python -c "import json; d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]); print('$tag', 'ms_per_step', round(d['ms_per_step'],3), 'pipelined', round(d['ms_per_step_pipelined'],3))" || tail -5 $O/ab_$tag.err; }
run deferred X=1
run joined GDHIP_BATCH_SHEAR_DEFERRED=0
run deferred2 X=1
run joined2 GDHIP_BATCH_SHEAR_DEFERRED=0
timeout 900 python -m pytest tests -m gpu -x -q -k "native_batch or c3_full_shape or density_2d" 2>&1 | tail -4
export TMPDIR=/tmp
(cd /tmp && GETDIST_AMD_LIVE_PMC=0 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lat -o lat -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/lat.log 2>&1)
python scripts/latency_timeline.py /tmp/lat 5 2>&1 | tee $O/latency_timeline.txt | head -16
