import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from genie_amd import engine, synthetic
from tests.util import Case
S, G = 200, 10000
geom = synthetic.Geometry(S, G, L=300e3, n_query=G, seed=1)
dev = "cuda:0"
hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S), engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G), grid_order=engine.morton_order(geom.x_grid), device=dev)
hp.set_weights({k: v.to(dev) for k, v in Case("cfg1_20x500").weights.items()})
pos = torch.from_numpy(geom.x_grid).float().to(dev)
xs = torch.randn(G, 30, device=dev)
d = torch.cdist(pos.double(), pos.double())
knn_all = d.topk(10, largest=False).indices.int().contiguous()
for T in (10, 1):
    tq = torch.arange(T, device=dev).float()
    for nq in (256, 4096, 8192, 10000, 16384 if False else 10000):
        xq = pos[:nq].contiguous(); knn = knn_all[:nq].contiguous()
        for f, name in ((lambda: hp.readout_query(xs, pos, xq, knn, tq), "query"), (lambda: hp.readout_grid(xs[:nq].contiguous() if False else xs, tq), "grid")):
            if name == "grid" and nq != 10000: continue
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            print("T=%d %s nq=%d: %.1f us" % (T, name, nq, e0.elapsed_time(e1) / 20 * 1e3))
print("--- slim tail mode")
hp.lib.genie_set_tail_mode(hp.ctx, 1)
tq = torch.arange(10, device=dev).float()
xq = pos.contiguous(); knn = knn_all.contiguous()
for f, name in ((lambda: hp.readout_query(xs, pos, xq, knn, tq), "query"), (lambda: hp.readout_grid(xs, tq), "grid")):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("slim T=10 %s nq=10000: %.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3))
