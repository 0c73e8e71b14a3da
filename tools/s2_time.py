#!/usr/bin/env python
"""Stage-2 kernel alone (HIP events over repeated launches after one stage 1), for A/B runs of kernel variants selected by
environment variables (GENIE_BPC2 = workgroups per CU, GENIE_S2_WGMAP = work map; read by -DGENIE_TUNING=1 builds only:
GENIE_LIB_PATH=genie_amd/lib/libgenie_tune.so).
Usage: python tools/s2_time.py [config] [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from genie_amd import engine, synthetic  # noqa: E402
from tests.util import Case  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_200x10k"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    S, G, n_picks, L, nq = synthetic.CONFIGS[cfg]
    geom = synthetic.Geometry(S, G, L=L, n_query=10, seed=1)
    win = synthetic.make_window(geom, n_picks, seed=2)
    dev = "cuda:0"
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G),
                        grid_order=engine.sfc_order(geom.x_grid), device=dev, sta_order=engine.sfc_order(geom.locs))
    hp.set_weights({k: v.to(dev) for k, v in Case("cfg1_20x500").weights.items()})
    Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
    ea = torch.from_numpy(geom.edge_attr()).to(dev)
    hp.set_static_edge_attr(ea)
    # stage 2 is timed where it runs in the path: right after a stage 1 that has just written c / wu / wv (a back-to-back loop
    # of stage 2 alone re-reads cache-warm rows and ranked k_stage2_lds 12 % ahead of k_stage2_fast; in sequence it is 6 % behind)
    ms, ms1 = [], []
    for k in range(iters + 10):
        ea_, e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        ea_.record()
        hp.da_stage1(Slice, Mask)
        e0.record()
        hp.da_stage2_partials_range(Mask, ea, 0, G)
        e1.record()
        torch.cuda.synchronize()
        if k >= 10:
            ms.append(e0.elapsed_time(e1))
            ms1.append(ea_.elapsed_time(e0))
    bip = hp.bipartite_readout()
    tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("GENIE_") and k != "GENIE_LIB_PATH")
    print("stage 2 %s [%s]: median %.4f ms after stage 1 (stage 1 incl. split %.4f ms)  (checksum %.6f)"
          % (cfg, tag or "defaults", sorted(ms)[len(ms) // 2], sorted(ms1)[len(ms1) // 2], float(bip.double().sum())))


if __name__ == "__main__":
    main()
