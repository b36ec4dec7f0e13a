mkdir -p gpurun_out; cd /root/repo
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_pytest_gpu.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | cut -c1-330
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r02_prof_bench.json 2> gpurun_out/r02_prof_bench.err; echo "prof rc=$?"
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_bench_kernel_stats.csv \;
find /tmp/prof -name "*kernel_trace.csv" -exec cp {} gpurun_out/r02_bench_kernel_trace.csv \;
for W in 8 2; do timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --emulate-world $W > gpurun_out/r02_emu$W.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r02_emu$W.json'));print($W, d['ms_per_step'])"; done
