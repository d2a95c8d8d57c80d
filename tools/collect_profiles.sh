#!/bin/bash
# Regenerates everything under profiles/ that bench.py's JSON line refers to (run on the GPU box
# from the repo root; results land in gpurun_out/profiles_<tag>/ and are copied into profiles/
# by hand afterwards):
#   1. PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, --kernel-trace only) for the whole-layer kernel of every
#      engine -- K8x (f16x3, the headline), K8h (f16x2), K8 (bf16x3) -- and K1 (--path k1)
#                                                    -> k8x_ / k8h_ / k8_pmc_traffic.json, k1_pmc_traffic.json
#   2. rocprofv3 --kernel-trace --stats of the bench command -> <tag>_kernel_stats_bench.csv
#   3. the bench line itself (reads the fresh traffic files) -> <tag>_bench_1gpu.json
#   tools/collect_profiles.sh <tag>          (SKIP_K1=1: the K1 kernel has not changed since its last collection)
set -u
TAG=${1:-r1}
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --skip-mfma-ceiling --skip-graph"
IO=135266304   # 262 144 rows x (256 B in + 256 B out + 4 B log-determinant)

# engine : counter file : kernel-name pattern : algorithmic bytes per launch (rows' I/O + 32 layers' packed weights)
for spec in "f16x3:k8x:rqs_resnet_f16x3_kernel:$((IO + 32 * 1007616))" "f16x2:k8h:rqs_resnet_f16_kernel:$((IO + 32 * 688128))" \
            "bf16x3:k8:rqs_resnet_kernel:$((IO + 32 * 1007616))"; do
  IFS=: read ENGINE NAME PATTERN BYTES <<< "$spec"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$NAME/$c -o p -- $BENCH --engine $ENGINE --skip-k1-roofline > $OUT/pmc_${NAME}_$c.log 2>&1
  done
  python $ROOTDIR/tools/pmc_traffic.py $OUT/pmc_$NAME $ROOTDIR/profiles/${NAME}_pmc_traffic.json $PATTERN $BYTES \
    "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --skip-k1-roofline --engine $ENGINE"
  cp $ROOTDIR/profiles/${NAME}_pmc_traffic.json $OUT/
  rm -rf $OUT/pmc_$NAME      # (the raw counter CSVs are large; keep only the summaries)
done
if [ "${SKIP_K1:-0}" != "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_k1/$c -o p -- $BENCH --path k1 > $OUT/pmc_k1_$c.log 2>&1
  done
  python $ROOTDIR/tools/pmc_traffic.py $OUT/pmc_k1 $ROOTDIR/profiles/k1_pmc_traffic.json rqs_coupling_wavetile 907018240 \
    "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --path k1"
  cp $ROOTDIR/profiles/k1_pmc_traffic.json $OUT/
  rm -rf $OUT/pmc_k1
fi

# the same command as the driver's (default steps), extras included: every engine's whole-layer kernel appears in the summary
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- python $ROOTDIR/bench.py --no-cpu-baseline --skip-mfma-ceiling --skip-graph > $OUT/stats_bench.log 2>&1
DB=$(find $OUT/stats -name '*.db' | head -1)
python $ROOTDIR/tools/rocprof_summary.py "$DB" $OUT/${TAG}_kernel_stats_bench.csv
rm -rf $OUT/stats

cd $ROOTDIR
timeout 900 python bench.py > $OUT/${TAG}_bench_1gpu.json 2> $OUT/bench.err
tail -c 1500 $OUT/${TAG}_bench_1gpu.json
