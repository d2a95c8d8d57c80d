#!/bin/bash
# SQ counter passes over the whole-layer kernel of one engine (rocprofv3 --pmc, kernel-trace only): where the waves' cycles go.
#   tools/sq_counters.sh <engine> <kernel-name pattern>      -> gpurun_out/r6/sq_<engine>.txt
ENGINE=$1; PATTERN=$2
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/r6
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --skip-mfma-ceiling --skip-graph --skip-k1-roofline --engine $ENGINE"
: > $OUT/sq_$ENGINE.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD"; do
  rm -rf $OUT/sq_tmp
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/sq_tmp -o p -- $BENCH > $OUT/sq_tmp.log 2>&1
  python - "$PATTERN" $OUT <<'PY' >> $OUT/sq_$ENGINE.txt
import sys, glob, csv, collections
pat, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/sq_tmp/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("%-32s %.4e per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
done
rm -rf $OUT/sq_tmp $OUT/sq_tmp.log
cat $OUT/sq_$ENGINE.txt
