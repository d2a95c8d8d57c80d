#!/bin/bash
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOTDIR
timeout 600 python -m pytest tests/test_gpu_grads.py tests/test_gpu_kernels.py tests/test_gpu_bin_index.py -q -x -m gpu 2>&1 | tail -3
bash tools/collect_profiles.sh r5 2>&1 | tail -3 | cut -c1-600
