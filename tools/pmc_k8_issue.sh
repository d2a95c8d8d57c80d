#!/bin/bash
# Matrix-pipe and VALU occupancy of the K8 launch from hardware counters (separate --pmc passes,
# --kernel-trace only):  tools/pmc_k8_issue.sh   -> gpurun_out/pmc_k8_issue/summary.txt
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/pmc_k8_issue
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-consistency --skip-k1-roofline --skip-graph --skip-mfma-ceiling"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $BENCH > $OUT/p$i.log 2>&1
done
python $ROOTDIR/tools/pmc_report.py $OUT rqs_resnet_kernel > $OUT/summary.txt 2>&1
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/summary.txt
