#!/bin/bash
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOTDIR
mkdir -p gpurun_out/r5f
timeout 400 python tests/probes/hog_all_families.py 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5f/hog_all_families.txt
