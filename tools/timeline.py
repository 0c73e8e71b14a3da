#!/usr/bin/env python
"""Print the kernel timeline of a few steady-state windows from a rocprofv3 --kernel-trace database (which kernels overlap,
how long each waits): python tools/timeline.py <results.db> [first_window] [n_windows]."""
import sqlite3
import sys
import re


def main():
    db = sqlite3.connect(sys.argv[1])
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    nwin = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = db.execute("select d.start, d.end, s.kernel_name, %s from %s d join %s s on d.kernel_id = s.id order by d.start" % (
        "d." + qcol if qcol else "0", kd, ks)).fetchall()
    short = lambda n: (re.search(r"k_\w+(<[^>]*>)?", n) or re.search(r"\w+", n)).group(0)
    s1 = [i for i, r in enumerate(rows) if "k_stage1" in r[2]]
    if len(s1) < first + nwin + 1:
        first = max(0, len(s1) - nwin - 1)
    a, b = s1[first], s1[first + nwin]
    t0 = rows[a][0]
    print("window period: %.1f us" % ((rows[b][0] - t0) / nwin / 1e3))
    for r in rows[a:b]:
        print("%9.1f %9.1f  %7.1f us  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], short(r[2])))


if __name__ == "__main__":
    main()
