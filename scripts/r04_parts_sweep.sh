export GETDIST_AMD_LIVE_PMC=0
for P in 2 1 3 2; do echo -n "parts $P: "; GDHIP_KOPT_PARTS=$P timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_single_triangle_latency"])'; done
