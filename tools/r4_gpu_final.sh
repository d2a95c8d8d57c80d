set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/profiles_r4b
mkdir -p $OUT
cd $ROOTDIR
timeout 150 python bench.py > $OUT/r4_bench_1gpu.json 2> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
timeout 110 rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- python $ROOTDIR/bench.py --no-cpu-baseline --skip-extra --skip-mfma-ceiling > $OUT/stats_bench.log 2>&1
DB=$(find $OUT/stats -name '*.db' | head -1)
python $ROOTDIR/tools/rocprof_summary.py "$DB" $OUT/r4_kernel_stats_bench.csv > /dev/null 2>&1
rm -rf $OUT/stats
timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/tstats -o k -- python $ROOTDIR/tools/train_probe.py 65536 > $OUT/train_probe.log 2>&1
DB=$(find $OUT/tstats -name '*.db' | head -1)
python $ROOTDIR/tools/rocprof_summary.py "$DB" $OUT/r4_train_kernel_stats.csv > /dev/null 2>&1
rm -rf $OUT/tstats
cd $ROOTDIR
timeout 40 python tools/bins_time.py 262144 20 24 32 2>&1 | grep "K=" > $OUT/bins_time_20_32.txt
tail -c 300 $OUT/r4_bench_1gpu.json; tail -2 $OUT/train_probe.log; cat $OUT/bins_time_20_32.txt
