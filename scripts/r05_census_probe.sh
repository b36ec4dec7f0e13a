cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05l
cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/memory/memory.limit_in_bytes 2>/dev/null; grep -E "MemTotal|MemAvailable" /proc/meminfo; nproc; df -h /dev/shm | tail -1
GETDIST_AMD_LIVE_PMC=0 GETDIST_AMD_CENSUS_WORKERS=16 GETDIST_AMD_CENSUS_MAX_PAIRS=128 timeout 800 python bench.py --steps 5 --warmup 2 > gpurun_out/r05l/bench_probe.json 2> gpurun_out/r05l/bench_probe.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05l/bench_probe.json').read().strip().splitlines()[-1])
f=d['parity'].get('full_size_census')
print({k:v for k,v in (f or {}).items() if k not in ('per_class','loose_pairs')})
print('ms', d['ms_per_step'], 'cpu', d['cpu_baseline']['value'])
PY
