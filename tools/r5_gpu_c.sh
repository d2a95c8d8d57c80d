#!/bin/bash
# round 5, third GPU call: the whole GPU suite (as the driver runs it) with durations, parity log kept
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/r5c
mkdir -p $OUT
cd $ROOTDIR
export NFA_PARITY_LOG=$OUT/parity.jsonl
rm -f gpurun_out/parity_report.jsonl
timeout 1150 python -m pytest tests -m gpu -q -x --durations=40 > $OUT/suite.log 2>&1
tail -60 $OUT/suite.log
cp gpurun_out/parity_report.jsonl $OUT/parity_report.jsonl 2>/dev/null
