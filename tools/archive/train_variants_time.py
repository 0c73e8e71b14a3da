#!/usr/bin/env python
"""Training step of forward_fixed_source at config 3 (200 x 10 000) under the three model definitions: ms per step (forward + loss +
backward + Adam) and a finiteness / non-zero check of the static-term gradient columns. Usage: python tools/train_variants_time.py"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import module, synthetic  # noqa


def main():
    S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
    geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
    dev = "cuda:0"
    locs, xg = torch.from_numpy(geom.locs).float().to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
    win = synthetic.make_window(geom, n_picks, seed=2)
    Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
    xq, tq = torch.from_numpy(geom.x_query).float().to(dev), torch.from_numpy(geom.t_query).float().to(dev)
    for name, kw in (("default", {}), ("use_updated_model_definition", dict(use_updated_model_definition=True)),
                     ("use_absolute_pos", dict(use_absolute_pos=True))):
        torch.manual_seed(0)
        net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev, **kw)
        net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                                 torch.from_numpy(geom.edge_attr()).to(dev), locs, xg)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        net.train()

        def step():
            opt.zero_grad(set_to_none=True)
            y, x = net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
            loss = 0.1 * (y ** 2).mean() + 0.4 * (x ** 2).mean()
            loss.backward()
            opt.step()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            l = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        p = dict(net.named_parameters())
        extra = ""
        if "use_updated_model_definition" in kw:
            g1 = p["DataAggregation.l1_t1_2.weight"].grad[:, 60:64]; g2 = p["DataAggregation.l2_t2_2.weight"].grad[:, 90:94]
            extra = ", static columns: max |grad| %.3e / %.3e, finite %s" % (float(g1.abs().max()), float(g2.abs().max()),
                                                                              bool(torch.isfinite(g1).all() and torch.isfinite(g2).all()))
        if "use_absolute_pos" in kw:
            g1 = p["DataAggregation.init_trns.weight"].grad[:, 4:10]
            extra = ", static columns: max |grad| %.3e, finite %s" % (float(g1.abs().max()), bool(torch.isfinite(g1).all()))
        print("%-30s %.2f ms per step, loss %.4e%s" % (name, dt, float(l), extra), flush=True)
        del net, opt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
