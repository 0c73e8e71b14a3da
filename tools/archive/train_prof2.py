#!/usr/bin/env python
"""torch profiler table of one training step at config 2 (which kernels the 17 ms go to)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import module, synthetic  # noqa
S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
dev = "cuda:0"
torch.manual_seed(0)
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
locs, xg = torch.from_numpy(geom.locs).float().to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), torch.from_numpy(geom.edge_attr()).to(dev), locs, xg)
win = synthetic.make_window(geom, n_picks, seed=2)
Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
xq, tq = torch.from_numpy(geom.x_query).float().to(dev), torch.from_numpy(geom.t_query).float().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
net.train()
def step():
    opt.zero_grad(set_to_none=True)
    y, x = net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
    ((y ** 2).mean() + (x ** 2).mean()).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=70))
