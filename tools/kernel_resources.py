#!/usr/bin/env python
"""VGPRs / LDS / spills / workgroup size of the kernels of a built libgenie_hip.so whose (mangled) name contains a pattern:
python tools/kernel_resources.py [lib.so] [pattern ...]   (llvm-objdump --offloading + llvm-readelf --notes; no GPU needed)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def resources(so):
    d = tempfile.mkdtemp()
    local = os.path.join(d, "lib.so")
    os.symlink(os.path.abspath(so), local)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", local], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = glob.glob(os.path.join(d, "*gfx950*"))
    if not co:
        raise SystemExit("no gfx950 code object in " + so)
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co[0]], stdout=subprocess.PIPE, text=True).stdout
    out = []
    for blk in txt.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        out.append((name.group(1), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("vgpr_spill_count"),
                    g("max_flat_workgroup_size")))
    return out


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "genie_amd", "lib", "libgenie_hip.so")
    pats = [a for a in sys.argv[1:] if not a.endswith(".so")]
    print("%-90s %5s %5s %7s %6s %5s" % ("kernel", "vgpr", "sgpr", "lds", "spill", "wg"))
    for name, v, s, l, sp, wg in resources(so):
        if not pats or any(p in name for p in pats):
            print("%-90s %5d %5d %7d %6d %5d" % (name[:90], v, s, l, sp, wg))
