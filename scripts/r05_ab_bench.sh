cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05ab; export GETDIST_AMD_LIVE_PMC=0
cp getdist_amd/csrc/libgdhip.so /tmp/new.so
for round in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then cp gpurun_ab/libgdhip_prev.so getdist_amd/csrc/libgdhip.so; else cp /tmp/new.so getdist_amd/csrc/libgdhip.so; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', round(d['ms_per_step'],3), round(d['ms_single_triangle_latency'],2))"
  done
done
cp /tmp/new.so getdist_amd/csrc/libgdhip.so
python -m pytest tests -m gpu -x -q -k "native or c3_full or density_2d" 2>&1 | tail -2
