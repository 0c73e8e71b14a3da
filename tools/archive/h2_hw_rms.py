#!/usr/bin/env python
"""On the hardware: rms / max error of x_latent (stage 1 + stage 2) against the fp64 oracle for the f16x2 stage 1 (default) and
the fp32-MFMA stage 1 (GENIE_S1=f32), on the golden fixtures; the counterpart of tests/test_h2_numerics_cpu.py's emulation.
Usage: python tools/h2_hw_rms.py   (spawns itself once per stage-1 form)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def child():
    import torch
    from genie_amd import engine
    from tests.util import Case
    dev = "cuda:0"
    for name in ("o1_20x500", "cfg1_20x500", "odd_33x257"):
        c = Case(name)
        ref = c.oracle_forward(torch.float64, structured=False)
        f32 = c.oracle_forward(torch.float32, structured=False)
        sta, src = c.tables()
        hp = engine.HipPath(c.S, c.G, engine.csr_from_table(sta), engine.csr_from_table(src),
                            grid_order=engine.morton_order(c.x_grid.numpy()), device=dev)
        hp.set_weights({k: v.to(dev) for k, v in c.weights.items()})
        _, _, h0, h1 = hp.da_stage1(c.Slice.to(dev), c.Mask.to(dev), debug=True)
        xl, bip = hp.da_stage2_bipartite(c.Mask.to(dev), c.edge_attr.to(dev), want_x_latent=True)
        out = []
        for k, v in (("h1", h1), ("x_latent", xl), ("bip", bip)):
            d = (v.cpu().double() - ref[k]).abs()
            d32 = (f32[k].double() - ref[k]).abs()
            out.append("%s rms %.2e max %.2e (torch fp32 on the CPU: rms %.2e max %.2e)" % (k, (d ** 2).mean().sqrt(), d.max(), (d32 ** 2).mean().sqrt(), d32.max()))
        print("  %-12s %s" % (name, " | ".join(out)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for form in ("default", "f32"):
            env = dict(os.environ)
            if form == "f32":
                env["GENIE_S1"] = "f32"
            print("stage 1 = %s" % ("k_stage1_h2 (two fp16 pieces, three products)" if form == "default" else "k_stage1 (fp32 MFMA)"))
            sys.stdout.flush()
            subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
