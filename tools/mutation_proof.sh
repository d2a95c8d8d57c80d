#!/bin/bash
# tools/mutation_proof.sh  (round 4)
# Proves that tests/test_gpu_steep.py SEES the defect it was written for: builds the library with the round-3
# Newton-slope defect restored in both f16 whole-layer kernels (-DNFA_MUTATION_NEWTON_SLOPE: `in_h * t5` for
# `in_w * t5`, csrc/rqs_fused8.hpp) and runs the steep-spline tests against that build.  Expected: the K8h / K8s
# cases FAIL (inverse error ratios far above 2 x), every other engine passes.  Build here (no GPU needed):
#     tools/mutation_proof.sh build
# run on the GPU box (the variant travels with the snapshot):
#     tools/mutation_proof.sh run      -> gpurun_out/steep_mutation_proof.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
V=$R/build_variants/newton_mutant.so
if [ "${1:-build}" == "build" ]; then
  set -e
  mkdir -p $R/build_variants
  HF="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Wno-pass-failed -I$R/include -I$R/nflows_amd/csrc"
  make -C $R/nflows_amd/csrc -s
  for s in rqs_resnet_f16 rqs_resnet_f16s; do
    /opt/rocm/bin/hipcc $HF -DNFA_MUTATION_NEWTON_SLOPE -c $R/nflows_amd/csrc/$s.hip -o /tmp/mut_$s.o &
  done
  wait
  SRCS=$(sed -n 's/^SRCS *:= *//p' $R/nflows_amd/csrc/Makefile)
  OBJS=""
  for s in $SRCS; do
    case $s in
      rqs_resnet_f16.hip|rqs_resnet_f16s.hip) OBJS="$OBJS /tmp/mut_${s%.hip}.o";;
      *) OBJS="$OBJS $R/nflows_amd/csrc/${s%.hip}.o";;
    esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V $OBJS
  echo built $V
else
  mkdir -p $R/gpurun_out
  OUT=$R/gpurun_out/steep_mutation_proof.txt
  echo "# NFLOWS_AMD_LIB=build_variants/newton_mutant.so (csrc/rqs_fused8.hpp built with -DNFA_MUTATION_NEWTON_SLOPE)" > $OUT
  echo "# python -m pytest tests/test_gpu_steep.py -q -m gpu -k steep_coupling_flow    (expected: k8h_* / k8s_* FAIL)" >> $OUT
  cd $R && NFLOWS_AMD_LIB=$V python -m pytest tests/test_gpu_steep.py -q -m gpu -k steep_coupling_flow -p no:cacheprovider 2>&1 \
    | grep -E "^(FAILED|PASSED|ERROR)|passed|failed|^E +AssertionError" | cut -c1-300 >> $OUT
  tail -25 $OUT
fi
