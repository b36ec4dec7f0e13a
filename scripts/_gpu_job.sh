mkdir -p gpurun_out; cd /root/repo
timeout 300 python -m pytest tests/test_gpu_primitives.py -q -x -k "minmax or sheared" 2>&1 | tail -3
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('bench', d['ms_per_step'], d['ms_single_triangle_latency'])"
timeout 700 python -m pytest tests -m gpu -q -x > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_pytest_gpu.log
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r02_prof_bench.json 2> gpurun_out/r02_prof_bench.err; echo "prof rc=$?"
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_bench_kernel_stats.csv \;
find /tmp/prof -name "*kernel_trace.csv" -exec cp {} gpurun_out/r02_bench_kernel_trace.csv \;
