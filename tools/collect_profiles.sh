#!/bin/bash
# Regenerates everything under profiles/ that bench.py's JSON line refers to (run on the GPU box
# from the repo root; results land in gpurun_out/profiles_<tag>/ and are copied into profiles/
# by hand afterwards):
#   1. PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, --kernel-trace only) for K8h (default
#      path) and K1 (--path k1)                      -> k8h_pmc_traffic.json, k1_pmc_traffic.json
#   2. rocprofv3 --kernel-trace --stats of the bench command -> <tag>_kernel_stats_bench.csv
#   3. the bench line itself (reads the fresh traffic files) -> <tag>_bench_1gpu.json
#   tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r1}
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --skip-mfma-ceiling"

for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_k8/$c -o p -- $BENCH --skip-k1-roofline > $OUT/pmc_k8_$c.log 2>&1
  # (SKIP_K1=1: the K1 kernel has not changed since its last collection)
  [ "${SKIP_K1:-0}" = "1" ] || timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_k1/$c -o p -- $BENCH --path k1 > $OUT/pmc_k1_$c.log 2>&1
done
python $ROOTDIR/tools/pmc_traffic.py $OUT/pmc_k8 $ROOTDIR/profiles/k8h_pmc_traffic.json rqs_resnet_f16_kernel 157286400 \
  "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --skip-k1-roofline"
[ "${SKIP_K1:-0}" = "1" ] || python $ROOTDIR/tools/pmc_traffic.py $OUT/pmc_k1 $ROOTDIR/profiles/k1_pmc_traffic.json rqs_coupling_wavetile 907018240 \
  "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --path k1"
cp $ROOTDIR/profiles/k8h_pmc_traffic.json $ROOTDIR/profiles/k1_pmc_traffic.json $OUT/
# the raw counter CSVs are large; keep only the summaries
rm -rf $OUT/pmc_k8 $OUT/pmc_k1

timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- python $ROOTDIR/bench.py --no-cpu-baseline --skip-extra --skip-mfma-ceiling > $OUT/stats_bench.log 2>&1
DB=$(find $OUT/stats -name '*.db' | head -1)
python $ROOTDIR/tools/rocprof_summary.py "$DB" $OUT/${TAG}_kernel_stats_bench.csv
rm -rf $OUT/stats

cd $ROOTDIR
timeout 600 python bench.py > $OUT/${TAG}_bench_1gpu.json 2> $OUT/bench.err
tail -c 2500 $OUT/${TAG}_bench_1gpu.json
