mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r02_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-budget-s 120 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/r02_bench.json
python -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
