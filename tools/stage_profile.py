#!/usr/bin/env python
"""Run only the HIP path kernels (no read-out heads) on one synthetic config, for rocprofv3 passes.
Usage: python tools/stage_profile.py [config] [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from genie_amd import engine, graph, synthetic  # noqa: E402
from tests.util import Case  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_200x10k"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    S, G, n_picks, L, nq = synthetic.CONFIGS[cfg]
    geom = synthetic.Geometry(S, G, L=L, n_query=10, seed=1)
    win = synthetic.make_window(geom, n_picks, seed=2)
    dev = "cuda:0"
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G),
                        grid_order=engine.sfc_order(geom.x_grid), device=dev, sta_order=engine.sfc_order(geom.locs))
    hp.set_weights({k: v.to(dev) for k, v in Case("cfg1_20x500").weights.items()})
    Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
    ea, pos = torch.from_numpy(geom.edge_attr()).to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
    hp.set_static_edge_attr(ea)
    for _ in range(iters):
        hp.path_fwd(Slice, Mask, ea, pos)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hp.path_fwd(Slice, Mask, ea, pos)
    e1.record()
    torch.cuda.synchronize()
    print("path_fwd %s: %.3f ms/window" % (cfg, e0.elapsed_time(e1) / iters))


if __name__ == "__main__":
    main()
