#!/bin/bash
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOTDIR
mkdir -p gpurun_out/r5f
timeout 600 python -m pytest tests/test_gpu_context.py -q -m gpu 2>&1 | tail -40 | cut -c1-300 | tee gpurun_out/r5f/context_test.txt
