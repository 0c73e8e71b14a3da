#!/usr/bin/env python
"""Standalone cost of the G-sized tail at config 2: per-window calls vs genie_tail_batched over 8 windows (run under
rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import ctypes, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from genie_amd import _lib, module, synthetic  # noqa
from genie_amd.engine import _ptr  # noqa

S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
dev = "cuda:0"
torch.manual_seed(0)
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
net.eval()
locs, xg = torch.from_numpy(geom.locs).float().to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), torch.from_numpy(geom.edge_attr()).to(dev), locs, xg)
win = synthetic.make_window(geom, n_picks, seed=2)
Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
xq, tq = torch.from_numpy(geom.x_query).float().to(dev), torch.from_numpy(geom.t_query).float().to(dev)
hp = net._hip
if os.environ.get("TAIL") == "f32":
    hp.set_tail_precision(False)
NB = int(os.environ.get("NB", "8"))
with torch.no_grad():
    net.window_batch = NB
    for _ in range(NB):
        net.push_window(Slice, Mask)
    y, x, ev = net.flush_windows(xg, xq, tq)
    torch.cuda.synchronize()
    knn = net.SpatialAttention.query_table(xq, xg, 10)
    tqf = tq.reshape(-1).contiguous()
    xs_o = torch.empty((NB, G, 30), device=dev); yo = torch.empty((NB, G, tqf.numel(), 1), device=dev)
    xo = torch.empty((NB, xq.shape[0], tqf.numel(), 1), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def batched():
        _lib.check(hp.lib.genie_tail_batched(hp.ctx, 0, NB, _ptr(xg), _ptr(xq), _ptr(knn), xq.shape[0], 10, _ptr(tqf), tqf.numel(),
                                             _ptr(xs_o), _ptr(yo), _ptr(xo), hp._ws_ptr, st), "tail")
    bip = torch.empty((G, 15), device=dev); xs1 = torch.empty((G, 30), device=dev)
    def single():
        _lib.check(hp.lib.genie_bipartite_readout(hp.ctx, _ptr(bip), hp._ws_ptr, st), "bip")
        _lib.check(hp.lib.genie_spatial_agg3_fwd(hp.ctx, _ptr(bip), _ptr(xg), _ptr(xs1), hp._ws_ptr, st), "sa")
        hp.readout_grid(xs1, tq); hp.readout_query(xs1, xg, xq, knn, tq)
    for label in ("MFMA tiles",):
        for name, f, per in (("per-window tail", single, 1), ("batched tail x%d" % NB, batched, NB)):
            for _ in range(5): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): f()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 30
            print("%-14s %-18s %.1f us per call, %.1f us per window" % (label, name, dt * 1e6, dt / per * 1e6))
    print("n_query", xq.shape[0], "T", tqf.numel())
