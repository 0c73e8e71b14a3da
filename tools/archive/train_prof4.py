#!/usr/bin/env python
"""torch profiler table of the reference's 4-output training step at config 3 (which kernels the step goes to)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import graph, module, synthetic, train as gtrain  # noqa
S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
dev = "cuda:0"
torch.manual_seed(0)
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
locs, xg, xq, tq = t(geom.locs), t(geom.x_grid), t(geom.x_query), t(geom.t_query)
net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), locs, xg)
smp = synthetic.training_sample(geom, int(sys.argv[1]) if len(sys.argv) > 1 else 4000, n_src=4, seed=3, window=0)
net._sta_tab = graph.neighbour_table(geom.A_sta_sta, S).long().to(dev)
net._src_tab = graph.neighbour_table(geom.A_src_src, G).long().to(dev)
net.A_edges_p, net.A_edges_s = t(smp["A_edges_p"]).long(), t(smp["A_edges_s"]).long()
net.dt_partition, net.tlatent = t(smp["dt_partition"]), t(smp["tlatent"])
args4 = (t(smp["Slice"]), t(smp["Mask"]), t(smp["tpick"]), t(smp["ipick"]).long(), t(smp["phase_label"]), locs, xg, xq,
         t(smp["x_query_src"]), tq, t(smp["tq_sample"]), t(smp["trv_out_q"]))
lab4 = (t(smp["Lbls"]), t(smp["Lbls_query"]), t(smp["pick_lbls"]))
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
net.train()
def step():
    opt.zero_grad(set_to_none=True)
    loss = gtrain.reference_loss(net.forward_fixed(*args4), lab4, 1)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print("4-output step: %.2f ms, picks %d" % ((time.perf_counter() - t0) / 5 * 1e3, len(smp["tpick"])))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=60))
