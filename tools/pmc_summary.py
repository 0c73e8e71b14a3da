#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel from *_counter_collection.csv files under a directory.
Usage: python tools/pmc_summary.py <dir> [kernel-substring ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    subs = sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if subs and not any(s in k for s in subs):
                continue
            key = (f, row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[key] = k
        for key, v in per_dispatch.items():
            m = re.search(r"(k_\w+)", names[key])
            acc[m.group(1) if m else names[key][:60]][key[2]].append(v)
    for k in sorted(acc):
        print(k)
        for cname in sorted(acc[k]):
            vals = acc[k][cname]
            print("    %-32s avg %.4g over %d dispatches" % (cname, sum(vals) / len(vals), len(vals)))


if __name__ == "__main__":
    main()
