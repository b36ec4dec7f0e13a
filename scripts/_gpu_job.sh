mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_pytest_gpu.log
timeout 500 python bench.py --steps 20 --warmup 5 --cpu-budget-s 120 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r02_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r02_prof_bench.json 2> gpurun_out/r02_prof_bench.err; echo "prof rc=$?"
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_bench_kernel_stats.csv \;
find /tmp/prof -name "*domain_stats.csv" -exec cp {} gpurun_out/r02_bench_domain_stats.csv \;
timeout 200 python scripts/kernel_bench.py > gpurun_out/r02_kernel_bench.txt 2>&1; tail -3 gpurun_out/r02_kernel_bench.txt | cut -c1-200
for W in 2 4 8; do timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --emulate-world $W > gpurun_out/r02_emu$W.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r02_emu$W.json'));print($W, d['ms_per_step'])"; done
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o k -- python scripts/tune_kopt.py > /dev/null 2>&1
  f=$(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1); cp "$f" gpurun_out/r02_pmc_kopt_$C.csv
done
ls gpurun_out | head -30
