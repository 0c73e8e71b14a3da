import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from genie_amd import synthetic
S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
dev = "cuda:0"
net = bench.build_model(geom, dev)
win = synthetic.make_window(geom, n_picks, seed=2)
dS, dM = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
locs = torch.from_numpy(geom.locs).float().to(dev); xg = torch.from_numpy(geom.x_grid).float().to(dev)
xq = torch.from_numpy(geom.x_query).float().to(dev); tq = torch.from_numpy(geom.t_query).float().to(dev)
with torch.no_grad():
    for _ in range(300): net.forward_fixed_source_pipelined(dS, dM, None, None, None, locs, xg, xq, tq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): net.forward_fixed_source_pipelined(dS, dM, None, None, None, locs, xg, xq, tq)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
print("host enqueue %.3f ms/window, total %.3f ms/window" % (t_enq / 300 * 1e3, t_all / 300 * 1e3))
