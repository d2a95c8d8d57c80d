#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats` result (rocpd sqlite .db) into the small CSV that
is committed under profiles/ (per-kernel calls, total / average / min / max duration in us).

    python tools/rocprof_summary.py gpurun_out/prof_x/k_results.db profiles/r1_x_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    try:
        rows = c.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
            "from kernels group by name order by sum(duration) desc").fetchall()
        scale = 1e-3  # ns -> us
    except sqlite3.OperationalError:
        rows = [(n, calls, tot, avg, None, None) for n, calls, tot, avg, _ in
                c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")]
        scale = 1.0
    total = sum(r[2] for r in rows) or 1.0
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([name, calls, round(tot * scale, 3), round(avg * scale, 3),
                        "" if mn is None else round(mn * scale, 3),
                        "" if mx is None else round(mx * scale, 3), round(100.0 * tot / total, 3)])
    print("wrote", out, "(%d kernels)" % len(rows))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
