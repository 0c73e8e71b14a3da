#!/usr/bin/env python
"""Debug: d r of genie_tail_train_bwd against the oracle's autograd of the tail (r -> y, x) on the CPU."""
import os, sys, ctypes
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import graph, module, synthetic, engine, _lib  # noqa
from tests.util import Case  # noqa
from oracle import genie_oracle as O  # noqa
S, G, Q = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (40, 300, 70)
dev = "cuda:0"
geom = synthetic.Geometry(S, G, L=200e3, n_query=Q, seed=3)
win = synthetic.make_window(geom, max(50, S * 20), seed=4)
w0 = Case(os.environ.get("CASE", "tiny_6x40")).weights
rng = np.random.default_rng(5)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S), engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G),
                    grid_order=engine.sfc_order(geom.x_grid), device=dev, sta_order=engine.sfc_order(geom.locs))
hp.set_weights({k: v.to(dev) for k, v in w0.items()})
Sl, Mk, ea = t(win["Slice"]), t(win["Mask"]), t(geom.edge_attr())
pos, xq, tq = t(geom.x_grid), t(geom.x_query), t(geom.t_query)
knn = engine.knn_device(pos, xq, 10)
y, x, xs, ylat, xl, save, tsave = hp.path_train_fwd(Sl, Mk, ea, pos, xq, knn, tq)
cy, cx = torch.from_numpy(rng.normal(0, 1, (G, 9)).astype(np.float32)), torch.from_numpy(rng.normal(0, 1, (Q, 9)).astype(np.float32))
rp, re = hp.reverse_query_table(knn)
scratch = torch.empty(int(hp.lib.genie_tail_train_scratch_floats(hp.ctx, Q)), dtype=torch.float32, device=dev)
d_r = torch.zeros((G, 32), dtype=torch.float32, device=dev)
blob = torch.empty(int(hp.lib.genie_train_grad_floats()), dtype=torch.float32, device=dev)
P = engine._ptr
_lib.check(hp.lib.genie_tail_train_bwd(hp.ctx, P(pos), P(xq), P(knn), P(rp), P(re), Q, 10, P(tq.reshape(-1)), 9, P(tsave), P(cy.to(dev)), P(cx.to(dev)),
                                       None, None, P(scratch), P(d_r), P(blob), engine._stream()), "tail bwd")
torch.cuda.synchronize()
r = tsave[:32 * G].view(G, 32)[:, :30].cpu().clone().requires_grad_(True)
w = {k: v.clone() for k, v in w0.items()}
bip = O.act(O.linear(r, w, "Bipartite_ReadIn.fc2"), w, "Bipartite_ReadIn.activate2")
A = torch.from_numpy(geom.A_src_src); xg = torch.from_numpy(geom.x_grid).float()
sa = bip
for k in (1, 2, 3):
    sa = O.spatial_aggregation(w, sa, A, xg, "SpatialAggregation%d" % k)
yo = O.temporal_attention(w, O.spatial_direct(w, sa), torch.from_numpy(geom.t_query).float())
edges = torch.stack([knn.cpu().long().reshape(-1), torch.arange(Q).repeat_interleave(10)])
xo = O.temporal_attention(w, O.spatial_attention(w, sa, torch.from_numpy(geom.x_query).float(), xg, edge_index=edges), torch.from_numpy(geom.t_query).float())
((yo[:, :, 0] * cy).sum() + (xo[:, :, 0] * cx).sum()).backward()
err = (d_r[:, :30].cpu() - r.grad).abs()
print("max|y-yo| %.2e  |d_r| max %.3e  err max %.3e at %s ; pad cols max %.2e" % (float((y[:, :, 0].cpu() - yo[:, :, 0]).abs().max()), float(r.grad.abs().max()),
      float(err.max()), np.unravel_index(int(err.argmax()), err.shape), float(d_r[:, 30:].abs().max())))
print("rows with err > 1e-4*max:", (err.max(1)[0] > 1e-4 * float(r.grad.abs().max())).nonzero().reshape(-1)[:40].tolist())
print("cols err:", err.max(0)[0].tolist())
