import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from genie_amd import module, synthetic
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
net.train()
def step():
    net.zero_grad(set_to_none=True)
    y, x = net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
    ((y ** 2).mean() + (x ** 2).mean()).backward()
for _ in range(3): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::sum", "aten::mm", "aten::copy_", "aten::cat", "aten::addmm", "aten::mul", "aten::add", "aten::index", "aten::index_add_", "aten::max", "aten::pad", "aten::constant_pad_nd", "aten::slice_backward", "aten::fill_", "aten::zero_")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:40]:
    print("%-22s %8.2f ms/step  x%-3d %s" % (e.key, e.self_device_time_total / 3e3, e.count // 3, str(e.input_shapes)[:110]))
