#!/bin/bash
# round 5, first GPU call: the new bin-index tests, the whole suite with per-test durations, the bench line
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/r5a
mkdir -p $OUT
cd $ROOTDIR
export NFA_PARITY_LOG=$OUT/parity.jsonl
timeout 600 python -m pytest tests/test_gpu_bin_index.py -q -x > $OUT/bins.log 2>&1
tail -5 $OUT/bins.log
timeout 1100 python -m pytest tests -m gpu -q --durations=80 --deselect tests/test_gpu_bin_index.py > $OUT/suite.log 2>&1
tail -100 $OUT/suite.log | head -95
unset NFA_PARITY_LOG
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
