cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/t3; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o k -- python $R/tools/train_probe.py 65536 > $O/stats.log 2>&1
DB=$(find $O/stats -name '*.db' | head -1)
python $R/tools/rocprof_summary.py "$DB" $O/train_kernel_stats.csv; rm -rf $O/stats
