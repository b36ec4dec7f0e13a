mkdir -p gpurun_out; cd /root/repo
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | cut -c1-420
GETDIST_AMD_LAZY_RESULTS=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('eager', d['ms_per_step'], d['ms_single_triangle_latency'])"
timeout 700 python -m pytest tests -m gpu -q -x > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_pytest_gpu.log
python -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
