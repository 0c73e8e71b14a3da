#!/usr/bin/env python
"""Per-parameter gradient comparison of the HIP training step (forward_fixed_source in train() mode) against the structured
oracle's autograd on the CPU: prints max |diff| / scale for every path parameter. Usage: train_grad_check.py [S G Q]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import graph, module, synthetic  # noqa
from tests.util import Case  # noqa
from oracle import genie_oracle as O  # noqa

S, G, Q = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (40, 300, 70)
dev = "cuda:0"
geom = synthetic.Geometry(S, G, L=200e3, n_query=Q, seed=3)
win = synthetic.make_window(geom, max(50, S * 20), seed=4)
w0 = Case(os.environ.get("CASE", "tiny_6x40")).weights
rng = np.random.default_rng(5)
lbl, lbl_q = torch.from_numpy(rng.random((G, 9)).astype(np.float32)), torch.from_numpy(rng.random((Q, 9)).astype(np.float32))
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
net.train()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), t(geom.locs), t(geom.x_grid))
y, x = net.forward_fixed_source(t(win["Slice"]), t(win["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query), t(geom.t_query))
mse = torch.nn.functional.mse_loss
cy, cx = torch.from_numpy(rng.normal(0, 1, (G, 9)).astype(np.float32)), torch.from_numpy(rng.normal(0, 1, (Q, 9)).astype(np.float32))
loss = (y[:, :, 0] * cy.to(dev)).sum() + (x[:, :, 0] * cx.to(dev)).sum()
loss.backward()
torch.cuda.synchronize()
w = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
yo, xo = O.forward_fixed_source_structured(w, c(win["Slice"]), c(win["Mask"]), graph.neighbour_table(geom.A_sta_sta, S),
                                           graph.neighbour_table(geom.A_src_src, G), c(geom.edge_attr()), torch.from_numpy(geom.A_src_src),
                                           c(geom.x_grid), c(geom.x_query), c(geom.t_query), S, G)
lo = (yo[:, :, 0] * cy).sum() + (xo[:, :, 0] * cx).sum()
lo.backward()
print("loss hip %.8g oracle %.8g | max|y-yo| %.2e max|x-xo| %.2e" % (float(loss), float(lo), float((y.cpu() - yo).abs().max()), float((x.cpu() - xo).abs().max())))
bad = 0
for k, p in net.named_parameters():
    if w[k].grad is None:
        continue
    if p.grad is None:
        print("%-55s MISSING" % k); bad += 1
        continue
    sc = max(1e-12, float(w[k].grad.abs().max()))
    err = float((p.grad.cpu() - w[k].grad).abs().max())
    flag = "" if err <= 2e-5 * sc else ("  <-- BAD" if err > 2e-4 * sc else "  <-- marginal")
    bad += flag.endswith("BAD")
    print("%-55s |g| %.3e err %.3e rel %.2e%s" % (k, sc, err, err / sc, flag))
print("bad:", bad)
