#!/bin/bash
# Matrix-pipe / VALU / wait accounting of the K8h launch from hardware counters (separate --pmc
# passes, --kernel-trace only):  tools/pmc_k8h_issue.sh [kernel-name-pattern]
#   -> gpurun_out/pmc_k8h_issue/summary.txt
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
PAT=${1:-rqs_resnet_f16_kernel}
OUT=$ROOTDIR/gpurun_out/pmc_k8h_issue
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --skip-k1-roofline --skip-graph --skip-mfma-ceiling"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $BENCH > $OUT/p$i.log 2>&1
done
python $ROOTDIR/tools/pmc_report.py $OUT $PAT > $OUT/summary.txt 2>&1
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/summary.txt
