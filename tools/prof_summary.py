#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a text table: per-kernel calls, total,
average, min, max duration. Usage: python tools/prof_summary.py <results.db> [top_n] > profiles/<name>.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace summary of %s" % sys.argv[1])
    print("# total kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    print("%-72s %6s %10s %6s %10s %10s %10s %5s %5s %5s %7s %9s %5s" % (
        "kernel", "calls", "total_ms", "pct", "avg_us", "min_us", "max_us", "vgpr", "agpr", "sgpr", "lds", "grid_x", "wg_x"))
    for r in rows[:top]:
        print("%-72s %6d %10.3f %6.2f %10.2f %10.2f %10.2f %5d %5d %5d %7d %9d %5d" % (
            r[0][:72], r[1], r[2] / 1e6, 100.0 * r[2] / total, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], r[9], r[10], r[11]))


if __name__ == "__main__":
    main()
