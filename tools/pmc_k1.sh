#!/bin/bash
# PMC passes over the K1 micro-benchmark (one rocprofv3 run per counter group; counters are
# collected WITHOUT any tracing domain other than --kernel-trace, as the pool requires).
#   tools/pmc_k1.sh <tag> [k1_micro args...]     (run on the GPU box from the repo root)
set -u
TAG=$1; shift
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $ROOTDIR/tools/k1_micro.py --reps 8 "$@" > $OUT/p$i.log 2>&1
done
ls -R $OUT | head -30
