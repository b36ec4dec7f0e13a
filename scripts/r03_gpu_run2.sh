#!/bin/bash
# second GPU session of round 3: full GPU test suite, kernel A/B, PMC passes of the binning kernels, bench with CPU baseline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest2.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r03_pytest2.log
timeout 600 python scripts/r03_kernels.py > gpurun_out/r03_kernels2.log 2>&1; echo "kernels rc=$?"
grep "^cov\|^u8" gpurun_out/r03_kernels2.log | cut -c1-400
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o h -- python $GRAFT_REPO_ROOT/scripts/pmc_hist2d.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ -o h -- python $GRAFT_REPO_ROOT/scripts/pmc_hist2d.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ.log 2>&1); echo "pmc SQ rc=$?"
F=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1); S=$(find gpurun_out/pmc_SQ -name "*counter_collection.csv" | head -1)
python scripts/summarise_pmc.py "$F" "$W" "$S" > gpurun_out/r03_pmc_summary.txt 2>&1; echo "summarise rc=$?"
cp "$F" gpurun_out/r03_pmc_hist2d_FETCH_SIZE.csv; cp "$W" gpurun_out/r03_pmc_hist2d_WRITE_SIZE.csv; cp "$S" gpurun_out/r03_pmc_hist2d_SQ.csv
cp profiles/r03_pmc_hist2d.json gpurun_out/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-budget-s 200 > gpurun_out/r03_bench2.log 2>&1; echo "bench rc=$?"
tail -c 2500 gpurun_out/r03_bench2.log
