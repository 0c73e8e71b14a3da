#!/usr/bin/env python
"""Time one training step of the path at config 2 (a-8 first pass): forward_fixed_source in train() mode + backward + Adam."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from genie_amd import module, synthetic  # noqa

def main():
    S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
    geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
    dev = "cuda:0"
    torch.manual_seed(0)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
    locs, xg = torch.from_numpy(geom.locs).float().to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                             torch.from_numpy(geom.edge_attr()).to(dev), locs, xg)
    win = synthetic.make_window(geom, n_picks, seed=2)
    Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
    xq, tq = torch.from_numpy(geom.x_query).float().to(dev), torch.from_numpy(geom.t_query).float().to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    net.train()
    def step():
        opt.zero_grad(set_to_none=True)
        y, x = net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
        loss = (y ** 2).mean() + (x ** 2).mean()
        loss.backward()
        opt.step()
        return float(loss.detach())
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): l = step()
    torch.cuda.synchronize()
    print("train step (forward + backward + Adam), config 2: %.1f ms, loss %.3e, peak mem %.1f GB" % (
        (time.perf_counter() - t0) / 10 * 1e3, l, torch.cuda.max_memory_allocated() / 2 ** 30))
    net.eval()
    with torch.no_grad():
        y, x = net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
    print("eval forward after the updates: finite", bool(torch.isfinite(y).all() and torch.isfinite(x).all()))

if __name__ == "__main__":
    main()
