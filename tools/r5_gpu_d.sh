#!/bin/bash
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOTDIR
mkdir -p gpurun_out/r5f
timeout 600 python -m pytest tests/test_gpu_realnvp.py tests/test_gpu_steep.py tests/test_gpu_headline_parity.py tests/test_gpu_concurrency.py -q -m gpu -k "realnvp" 2>&1 | grep -v "^    \|^$" | tail -30 | cut -c1-400 | tee gpurun_out/r5f/realnvp_test.txt
