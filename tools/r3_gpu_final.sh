#!/bin/bash
# One GPU call at the end of round 3: the whole -m gpu suite (no -x: every failure shows), the K9 / K5d backward
# timings, the bench line.  Outputs under gpurun_out/r3_final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3_final
mkdir -p $O
export TMPDIR=/tmp
timeout 540 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log
tail -5 $O/gpu_suite.log
timeout 150 python tools/k9_bwd_micro.py > $O/k9_backward.txt 2>&1; echo "rc=$?" >> $O/k9_backward.txt
cat $O/k9_backward.txt
timeout 240 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cat $O/bench.json | cut -c1-600
