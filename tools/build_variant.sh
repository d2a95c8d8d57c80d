#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...]  -> build_variants/<name>.so (experiments only)
set -e
R=/root/repo
N=$1; shift
mkdir -p $R/build_variants
HF="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I$R/include -I$R/nflows_amd/csrc"
/opt/rocm/bin/hipcc $HF "$@" -c $R/nflows_amd/csrc/rqs.hip -o /tmp/rqs_$N.o 2>/tmp/build_$N.log || { tail -5 /tmp/build_$N.log; exit 1; }
[ -f $R/nflows_amd/csrc/misc.o ] || make -C $R/nflows_amd/csrc -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/$N.so /tmp/rqs_$N.o $R/nflows_amd/csrc/misc.o $R/nflows_amd/csrc/rqs_bwd.o
