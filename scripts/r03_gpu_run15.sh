#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "lds_convolution" > gpurun_out/r03_pytest15.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r03_pytest15.log | cut -c1-250
