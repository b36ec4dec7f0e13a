#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_native_batch.py -x -q -k two_objects 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | head
timeout 1500 python scripts/parity_census.py --out gpurun_out/r04/r04_parity_census_1e7.json > gpurun_out/r04/census_full.log 2>&1; echo "census rc=$?"
tail -40 gpurun_out/r04/census_full.log | cut -c1-500
