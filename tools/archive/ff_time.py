#!/usr/bin/env python
"""forward_fixed (4 outputs: source branch + association heads, module.py:963-997) at config 2, eval mode: total and the
association part; torch profiler table of one call."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import graph, module, synthetic  # noqa

S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
dev = "cuda:0"
torch.manual_seed(0)
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev).eval()
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
net.set_adjacencies_from_positions(t(geom.locs), t(geom.x_grid), t(geom.edge_attr()))
smp = synthetic.training_sample(geom, int(os.environ.get("N_PICKS", "5000")), n_src=8, seed=3)
net.A_edges_p, net.A_edges_s = t(smp["A_edges_p"], torch.long), t(smp["A_edges_s"], torch.long)
net.dt_partition, net.tlatent = t(smp["dt_partition"]), t(smp["tlatent"])
args = (t(smp["Slice"]), t(smp["Mask"]), t(smp["tpick"]), t(smp["ipick"], torch.long), t(smp["phase_label"]), t(geom.locs), t(geom.x_grid),
        t(geom.x_query), t(smp["x_query_src"]), t(geom.t_query), t(smp["tq_sample"]), t(smp["trv_out_q"]))
with torch.no_grad():
    for _ in range(3): out = net.forward_fixed(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = net.forward_fixed(*args)
    torch.cuda.synchronize()
    print("forward_fixed, config 2, %d picks, 8 candidate sources: %.2f ms" % (len(smp["tpick"]), (time.perf_counter() - t0) / 10 * 1e3))
    for _ in range(3): net.forward_fixed_source(*args[:8], args[9])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net.forward_fixed_source(*args[:8], args[9])
    torch.cuda.synchronize()
    print("forward_fixed_source (single stream): %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net.forward_fixed(*args); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
