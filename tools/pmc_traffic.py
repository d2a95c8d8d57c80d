#!/usr/bin/env python3
"""HBM traffic per K1 launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in
SEPARATE runs, as MI355X_MICROARCH.md prescribes: they do not fit one pass).

    python tools/pmc_traffic.py <dir with fetch/ and write/ csv outputs> profiles/k1_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB-like units of
1024 B; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
streaming read, so the read side is doubled; WRITE_SIZE is used as reported (uncalibrated in the
guide; it matches the algorithmic write bytes here within 7 %).  Only full-batch launches of the
pipelined kernel are averaged."""
import csv
import glob
import json
import os
import sys


def mean_counter(d, counter, pat="rqs_coupling_pipelined", min_grid=512 * 256):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"] and row["Counter_Name"] == counter and int(row["Grid_Size"]) >= min_grid:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main(d, out, pat="rqs_coupling_pipelined", algorithmic=None, command=None):
    fetch, nf = mean_counter(d, "FETCH_SIZE", pat)
    write, nw = mean_counter(d, "WRITE_SIZE", pat)
    res = {"kernel": pat, "launches_averaged": [nf, nw],
           "FETCH_SIZE_raw_KiB": fetch, "WRITE_SIZE_raw_KiB": write,
           "fetch_bytes_corrected_x2": None if fetch is None else fetch * 1024 * 2,
           "write_bytes": None if write is None else write * 1024,
           "hbm_bytes_per_launch": None if fetch is None or write is None else fetch * 2048 + write * 1024,
           "algorithmic_bytes_per_launch": algorithmic if algorithmic is not None
           else 4 * (65536 * 64 * 2 + 65536 * 32 * 23 + 65536),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `%s`; "
                     "gfx950 x2 correction on reads" % (command or "python bench.py --steps 3 --warmup 1 --no-cpu-baseline")}
    try:  # the digest of the kernel's sources: bench.py quotes these counters only while they are unchanged
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        res["kernel_source_sha256"] = bench.kernel_source_digest(os.path.basename(out))
    except Exception as e:
        res["kernel_source_sha256"] = None
        res["digest_error"] = repr(e)
    try:  # an explanatory note written into the previous file by hand travels along
        note = json.load(open(out)).get("note")
        if note:
            res["note"] = note
    except (OSError, ValueError):
        pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    # pmc_traffic.py <dir> <out.json> [kernel-name-substring] [algorithmic-bytes] [command]
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4] or ["rqs_coupling_pipelined"]),
         algorithmic=int(sys.argv[4]) if len(sys.argv) > 4 else None,
         command=sys.argv[5] if len(sys.argv) > 5 else None)
