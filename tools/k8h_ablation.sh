#!/bin/bash
# Times the product build and every build_variants/abl_*.so (tools/build_variant.sh ... -DNFA_ABL_*) on the
# bench workload: what the kernel's time is made of.  Output: gpurun_out/k8h_ablation.txt
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/k8h_ablation.txt
mkdir -p $ROOTDIR/gpurun_out
: > $OUT
export NFA_K8H_NOREDO=1
python $ROOTDIR/tools/k8h_time.py 262144 32768 >> $OUT 2>&1
for v in $ROOTDIR/build_variants/*.so; do
  NFLOWS_AMD_LIB=$v timeout 120 python $ROOTDIR/tools/k8h_time.py 262144 32768 2>&1 | grep -E "rows|Error|error" >> $OUT
done
cat $OUT
