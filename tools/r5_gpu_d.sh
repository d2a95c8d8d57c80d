#!/bin/bash
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOTDIR
mkdir -p gpurun_out/r5f
timeout 300 python -m pytest tests/test_gpu_concurrency.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r5f/concurrency_test.txt
timeout 200 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r5f/bench.json
