#!/bin/bash
# Same-box comparison of several environment settings of a C3 step: ab_matrix.sh <tag> <reps> "ENV=V ENV2=V" "..." ...
# ("base" = no switch).  Alternates the settings, `reps` rounds of 30-step runs; one line per run in gpurun_out/<tag>/ab_matrix.txt.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG="$1"; REPS="$2"; shift 2
O=gpurun_out/$TAG; mkdir -p "$O"
for rep in $(seq 1 "$REPS"); do
    for cfg in "$@"; do
        if [ "$cfg" = base ]; then envs="GDAMD_AB=base"; else envs="$cfg"; fi
        env $envs GETDIST_AMD_LIVE_PMC=0 timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > "$O/abm_tmp.json" 2> "$O/abm_tmp.err"
        python -c "import json; d=json.loads(open('$O/abm_tmp.json').read().strip().splitlines()[-1]); print('%-60s delivered %.3f pipelined %.3f' % ('$cfg', d['ms_per_step'], d['ms_per_step_pipelined']))" | tee -a "$O/ab_matrix.txt"
    done
done
