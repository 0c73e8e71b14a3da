#!/usr/bin/env python
"""Gradients of the rough-cotangent stress case (33 x 257, scaled `o1` weights) against the structured oracle run in fp64: how far
is the HIP training step from the fp64 truth, and how far is the fp32 oracle itself? (PReLU makes the gradient a discontinuous
function of the pre-activations: two fp32 evaluations of the same forward differ by sign flips of near-zero pre-activations.)"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import graph, module, synthetic  # noqa
from oracle import genie_oracle as O  # noqa
from tests.util import Case  # noqa
DEV = "cuda:0"
S, G, Q = 33, 257, 100
geom = synthetic.Geometry(S, G, L=200e3, n_query=Q, seed=3)
win = synthetic.make_window(geom, 700, seed=4)
w0 = Case("o1_20x500").weights
rng = np.random.default_rng(5)
cy, cx = torch.from_numpy(rng.normal(0, 1, (G, 9)).astype(np.float32)), torch.from_numpy(rng.normal(0, 1, (Q, 9)).astype(np.float32))
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
net.train()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), t(geom.locs), t(geom.x_grid))
y, x = net.forward_fixed_source(t(win["Slice"]), t(win["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query), t(geom.t_query))
((y[:, :, 0] * cy.to(DEV)).sum() + (x[:, :, 0] * cx.to(DEV)).sum()).backward()


def oracle(dt):
    w = {k: v.clone().to(dt).requires_grad_(True) for k, v in w0.items()}
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    yo, xo = O.forward_fixed_source_structured(w, c(win["Slice"]), c(win["Mask"]), graph.neighbour_table(geom.A_sta_sta, S),
                                               graph.neighbour_table(geom.A_src_src, G), c(geom.edge_attr()), torch.from_numpy(geom.A_src_src),
                                               c(geom.x_grid), c(geom.x_query), c(geom.t_query), S, G)
    ((yo[:, :, 0] * cy.to(dt)).sum() + (xo[:, :, 0] * cx.to(dt)).sum()).backward()
    return {k: w[k].grad for k in module.TRAIN_PATH_PARAMS}


g64, g32 = oracle(torch.float64), oracle(torch.float32)
worst = [0.0, 0.0, 0.0]
for k in module.TRAIN_PATH_PARAMS:
    sc = float(g64[k].abs().max())
    gh = net.get_parameter(k).grad.cpu().double()
    e = [float((gh - g64[k]).abs().max()) / sc, float((g32[k].double() - g64[k]).abs().max()) / sc, float((gh - g32[k].double()).abs().max()) / sc]
    worst = [max(a, b) for a, b in zip(worst, e)]
    if max(e) > 5e-5:
        print("%-45s |g| %.3e  HIP-fp64 %.2e  oracle32-fp64 %.2e  HIP-oracle32 %.2e" % (k, sc, *e))
print("worst relative deviation: HIP vs fp64 %.2e, fp32 oracle vs fp64 %.2e, HIP vs fp32 oracle %.2e" % tuple(worst))
