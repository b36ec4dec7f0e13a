#!/bin/bash
# Same-box A/B of one library switch on the C3 bench:  gpurun -- 'bash scripts/r05_ab_bench.sh VAR[=VALUE] [inverse] [notests]'
# (the variable set = the old behaviour; with `inverse` = the new one).  Alternating runs, three of each.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05ab; export GETDIST_AMD_LIVE_PMC=0
VAR="${1:-GDHIP_MAIN_STREAM_NORMAL}"; VAL=1
case "$VAR" in *=*) VAL="${VAR#*=}"; VAR="${VAR%%=*}";; esac
for round in 1 2 3; do
  for which in new old; do
    if [ "$2" = inverse ]; then  # the variable set = the NEW behaviour
      if [ $which = new ]; then export "$VAR"="$VAL"; else unset "$VAR"; fi
    else
      if [ $which = old ]; then export "$VAR"="$VAL"; else unset "$VAR"; fi
    fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', round(d['ms_per_step'],3), round(d['ms_single_triangle_latency'],2))"
  done
done
unset "$VAR"
[ "$3" = notests ] && exit 0
python -m pytest tests -m gpu -x -q -k "native or c3_full or density_2d or smoke" 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
