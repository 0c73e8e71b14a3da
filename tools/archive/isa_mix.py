#!/usr/bin/env python
"""Instruction mix of the large basic blocks (the hot loops) of one kernel in a hipcc -S listing.
Usage: python tools/isa_mix.py listing.s kernel_name_substring"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:end]:
        l = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, cur)); cur = []; name = l
        elif l and not l.startswith(";") and not l.startswith("."):
            cur.append(l.split()[0])
    blocks.append((name, cur))
    for name, b in blocks:
        if len(b) < 150:
            continue
        c = collections.Counter(b)
        grp = lambda f: sum(v for k, v in c.items() if f(k))
        print("%s %d instructions: valu %d mfma %d lds %d vmem %d salu %d" % (
            name, len(b), grp(lambda k: k.startswith("v_") and not k.startswith("v_mfma") and not k.startswith("v_accvgpr")),
            grp(lambda k: k.startswith("v_mfma")), grp(lambda k: k.startswith("ds_")),
            grp(lambda k: k.startswith(("global_", "buffer_", "scratch_", "flat_"))), grp(lambda k: k.startswith("s_"))))
        print("   ", ", ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])[:30]))


if __name__ == "__main__":
    main()
