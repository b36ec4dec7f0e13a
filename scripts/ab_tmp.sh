#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06n
for rep in 1 2 3; do
 for side in base GDHIP_KDE_LAG_BLOCKS_PER_CU=4 GDHIP_KDE_LAG_BLOCKS_PER_CU=2 GDHIP_KDE_LAG_BLOCKS_PER_CU=6; do
  if [ "$side" = base ]; then envs="GDAMD_AB=base"; else envs="$side"; fi
  env $envs GETDIST_AMD_LIVE_PMC=0 timeout 400 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r06n/ab.json 2> gpurun_out/r06n/ab.err
  python -c "import json; d=json.loads(open('gpurun_out/r06n/ab.json').read().strip().splitlines()[-1]); print('$side', 'delivered', round(d['ms_per_step'],3), 'pipelined', round(d['ms_per_step_pipelined'],3))"
 done
done
