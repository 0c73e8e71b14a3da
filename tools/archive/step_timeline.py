#!/usr/bin/env python
"""Kernel timeline of ONE steady-state step from a rocprofv3 --kernel-trace database: the dispatches between two consecutive launches of
an anchor kernel (default k_assoc_a: once per 4-output training step), with the idle time of the device between them.
python tools/step_timeline.py <results.db> [anchor substring] [which occurrence]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_assoc_a"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
    short = lambda n: (re.search(r"k_\w+(<[^>]*>)?", n) or re.search(r"\w+", n)).group(0)
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    a, b = idx[which], idx[which + 1]
    t0, busy_until, busy, idle = rows[a][0], rows[a][0], 0.0, 0.0
    for r in rows[a:b]:
        gap = max(0.0, (r[0] - busy_until) / 1e3)
        idle += gap
        busy_until = max(busy_until, r[1])
        print("%9.1f %8.1f us  (idle before %6.1f)  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap, short(r[2])))
    period = (rows[b][0] - t0) / 1e3
    idle += max(0.0, (rows[b][0] - busy_until) / 1e3)
    print("step period %.1f us, %d dispatches, sum of kernel times %.1f us, device idle %.1f us" % (
        period, b - a, sum((r[1] - r[0]) for r in rows[a:b]) / 1e3, idle))


if __name__ == "__main__":
    main()
