#!/bin/bash
# Round 6, the bounded K8h energy experiment: cache policy of the weight stream's LDS-DMA loads.
# Variants are built by hand (csrc/k8h_common.hpp: -DNFA_K8H_DMA_AUX=2|16|18 on rqs_resnet_f16.hip, linked to
# tools/bin/libnfa_k8h_aux<v>.so); this script times the bench step, samples clock and socket power, and takes the
# kernel's HBM fetch bytes (rocprofv3 --pmc FETCH_SIZE) for each.  Output: gpurun_out/r6/k8h_energy_experiment.txt
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/r6
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in default 2 16 18; do
  if [ $v = default ]; then unset NFLOWS_AMD_LIB; else export NFLOWS_AMD_LIB=$ROOTDIR/tools/bin/libnfa_k8h_aux$v.so; fi
  python $ROOTDIR/bench.py --engine f16x2 --steps 40 --warmup 5 --no-cpu-baseline --skip-extra --skip-consistency --skip-k1-roofline --skip-graph --skip-mfma-ceiling > $OUT/energy_$v.json 2> $OUT/energy_$v.err
  rm -rf $OUT/pmc_energy_$v
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_energy_$v -o p -- python $ROOTDIR/bench.py --engine f16x2 --steps 3 --warmup 1 --no-cpu-baseline --skip-extra --skip-consistency --skip-k1-roofline --skip-graph --skip-mfma-ceiling > $OUT/pmc_energy_$v.log 2>&1
  python - $v $OUT <<'PY'
import sys, json, glob, csv
v, out = sys.argv[1], sys.argv[2]
r = json.load(open("%s/energy_%s.json" % (out, v)))
ms = r["roofline"]["avg_launch_ms"]
cp = r["steady_state"]["clock_and_power"]
fetch = []
for f in glob.glob("%s/pmc_energy_%s/**/*counter_collection.csv" % (out, v), recursive=True):
    for row in csv.DictReader(open(f)):
        if "rqs_resnet_f16_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == "FETCH_SIZE":
            fetch.append(float(row["Counter_Value"]))
kib = sum(fetch) / max(1, len(fetch))
w = cp["socket_power_w_median"]
line = "aux=%-8s launch %.3f ms  sclk %s MHz  socket %.0f W  energy/launch %.2f J  (%.1f uJ/sample)  HBM fetch %.3f GB/launch (x2-corrected, %d launches)" % (
    v, ms, cp["sclk_mhz_median"], w, w * ms * 1e-3, w * ms * 1e-3 / 262144 * 1e6, kib * 1024 * 2 / 1e9, len(fetch))
print(line)
open("%s/k8h_energy_experiment.txt" % out, "a").write(line + "\n")
PY
  rm -rf $OUT/pmc_energy_$v
done
