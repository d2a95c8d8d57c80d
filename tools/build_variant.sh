#!/bin/bash
# tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]  -> build_variants/<name>.so
# (experiments only: one source of nflows_amd/csrc recompiled with extra -D flags, linked with the
# regular objects; select it at run time with NFLOWS_AMD_LIB=build_variants/<name>.so)
# The object list is the Makefile's SRCS, so a variant always exports every symbol _native.load() asks for.
set -e
R=/root/repo
N=$1; SRC=$2; shift; shift
mkdir -p $R/build_variants
HF="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Wno-pass-failed -I$R/include -I$R/nflows_amd/csrc"
make -C $R/nflows_amd/csrc -s
/opt/rocm/bin/hipcc $HF "$@" -c $R/nflows_amd/csrc/$SRC -o /tmp/var_$N.o 2>/tmp/build_$N.log || { tail -5 /tmp/build_$N.log; exit 1; }
SRCS=$(sed -n 's/^SRCS *:= *//p' $R/nflows_amd/csrc/Makefile)
OBJS=""
for s in $SRCS; do
  if [ "$s" == "$SRC" ]; then OBJS="$OBJS /tmp/var_$N.o"; else OBJS="$OBJS $R/nflows_amd/csrc/${s%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/$N.so $OBJS
echo built $R/build_variants/$N.so
