#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats` result (rocpd sqlite .db) into the small CSV that
is committed under profiles/ (per-kernel calls, total / average / min / max duration in us).
nflows_amd kernels are additionally broken down by grid size, so that the full-batch launches of
the timed region are not averaged with the small launches of the consistency check.

    python tools/rocprof_summary.py gpurun_out/prof_x/k_results.db profiles/r1_x_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    per_grid = c.execute(
        "select name || ' [grid ' || grid_x || ' x wg ' || workgroup_x || ', vgpr ' || vgpr_count || "
        "', lds ' || lds_size || ']', count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels where name like '%nfa::%' group by name, grid_x order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1.0
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([name, calls, round(tot / 1e3, 3), round(avg / 1e3, 3), round(mn / 1e3, 3),
                        round(mx / 1e3, 3), round(100.0 * tot / total, 3)])
        w.writerow([])
        w.writerow(["nflows_amd kernels by launch shape", "calls", "total_us", "avg_us", "min_us", "max_us", ""])
        for name, calls, tot, avg, mn, mx in per_grid:
            w.writerow([name, calls, round(tot / 1e3, 3), round(avg / 1e3, 3), round(mn / 1e3, 3),
                        round(mx / 1e3, 3), ""])
    print("wrote", out, "(%d kernels)" % len(rows))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
