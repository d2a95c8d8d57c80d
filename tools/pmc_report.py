#!/usr/bin/env python3
"""Summarise tools/pmc_k1.sh output: per-counter mean over the K1 dispatches."""
import csv, glob, os, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "rqs_coupling"
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(d, "p*", "*counter_collection.csv"))):
    vals = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta = (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Grid_Size"), row.get("Scratch_Size"))
    for k, v in vals.items():
        agg[k] = sum(v) / len(v)
print("dispatch meta (vgpr, sgpr, lds, grid, scratch):", meta)
for k, v in agg.items():
    print("%-24s %16.1f" % (k, v))
for f in sorted(glob.glob(os.path.join(d, "p1", "*kernel_trace.csv"))):
    durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    if durs: print("kernel duration under PMC pass 1: mean %.1f us over %d" % (sum(durs) / len(durs) / 1e3, len(durs)))
