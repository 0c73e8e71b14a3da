"""Where the per-sample context rebuild of `forward()` spends its time (train_GENIE_model.py:1722-1786: a new station subset, grid and
product edge lists per sample): python tools/rebuild_time.py [S G n_samples] -> per-call times and a cProfile of one rebuild."""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from genie_amd import graph, module, synthetic  # noqa: E402

S, G, N = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (200, 10000, 4)
dev = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
torch.manual_seed(0)
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
net.train()


def sample(i):
    geom = synthetic.Geometry(S - (i % 3), G, L=300e3, n_query=2000, seed=100 + i)
    smp = synthetic.training_sample(geom, 3000, n_src=4, seed=3 + i)
    A1, A2, A3, A4 = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, geom.n_sta, G, device=dev)
    ea = graph.GraphEdges(x=t(geom.edge_attr()), edge_index=A3)
    eaf = graph.GraphEdges(x=ea.x, edge_index=A3.flip(0).contiguous())
    return (t(smp["Slice"]), t(smp["Mask"]), A1, A2, ea, eaf, A4, torch.from_numpy(geom.A_src_src).to(dev), t(smp["A_edges_p"]).long(),
            t(smp["A_edges_s"]).long(), t(smp["dt_partition"]), t(smp["tlatent"]), t(smp["tpick"]), t(smp["ipick"]).long(), t(smp["phase_label"]),
            t(geom.locs), t(geom.x_grid), t(geom.x_query), t(smp["x_query_src"]), t(geom.t_query), t(smp["tq_sample"]), t(smp["trv_out_q"]))


samples = [sample(i) for i in range(N)]
torch.cuda.synchronize()
for rep in range(2):
    for i, s in enumerate(samples):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = net(*s)
        loss = sum(o.sum() for o in out)
        loss.backward()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = net(*s)                       # same graphs again: cache hit
        loss = sum(o.sum() for o in out)
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("rep %d sample %d: step with rebuild %.2f ms, same graphs again %.2f ms" % (rep, i, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
net(*samples[1])
torch.cuda.synchronize()          # the profile below starts with an idle device, as the timed steps above do
pr = cProfile.Profile()
for i in (0, 1, 2, 0, 1, 2):      # six forwards, each on another graph than the one before, each from an idle device
    pr.enable()
    out = net(*samples[i])
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
if len(sys.argv) > 4:
    sys.exit(0)

# ---- phase timers (each phase bracketed by device synchronisation)
from genie_amd import engine  # noqa: E402


def timed(name, fn, acc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return r


acc = {}
for rep in range(3):
    for s in samples:
        A1, A2, ea, eaf, A4, A_src, locs, xg = s[2], s[3], s[4], s[5], s[6], s[7], s[15], s[16]
        n_sta, n_grid = locs.shape[0], xg.shape[0]
        tabs = timed("base_tables_from_product", lambda: graph.base_tables_from_product(A1, A2, n_sta, n_grid), acc)
        src_from_A = timed("csr_from_edges(A_src)", lambda: engine.csr_from_edges(A_src, n_grid), acc)
        src_csr = engine.csr_from_table(tabs[1])
        timed("torch.equal csr", lambda: torch.equal(src_from_A[0].cpu(), src_csr[0].cpu()) and torch.equal(src_from_A[1].cpu(), src_csr[1].cpu()), acc)
        order = timed("sfc_order x2 (host)", lambda: (engine.sfc_order(xg.cpu().numpy()), engine.sfc_order(locs.cpu().numpy())), acc)
        old = net._hip
        net._hip = None
        timed("old context destroy", lambda: old.__del__() if old is not None else None, acc)
        del old
        hp = timed("HipPath.__init__", lambda: engine.HipPath(n_sta, n_grid, engine.csr_from_table(tabs[0]), src_csr, grid_order=order[0],
                                                              scale_rel=net.scale_rel, device=dev, sta_order=order[1]), acc)
        net._hip = hp
        net._configure_engine()
        net._edge_attr = ea.x
        timed("set_static_edge_attr", lambda: hp.set_static_edge_attr(ea.x), acc)
        timed("sync_weights (upload, pack, range guard)", lambda: hp.sync_weights(net._path_params, None), acc)
        timed("first path_train_fwd (tables on first use)", lambda: net._path_train(s[0], s[1], xg, s[17], s[19]), acc)
        timed("second path_train_fwd", lambda: net._path_train(s[0], s[1], xg, s[17], s[19]), acc)
for k, v in acc.items():
    print("%-48s %s" % (k, " ".join("%7.2f" % x for x in v)))

# ---- per-C-entry-point times of one rebuild + first / second training step on the new context (host time incl. a device sync after each call)
import ctypes  # noqa: E402
from genie_amd import _lib  # noqa: E402

lib = _lib.load()


class _Timed(object):
    def __init__(self, name, fn, log):
        self.name, self.fn, self.log = name, fn, log

    def __call__(self, *a):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = self.fn(*a)
        torch.cuda.synchronize()
        self.log.append((self.name, (time.perf_counter() - t0) * 1e3))
        return r


class _Proxy(object):
    def __init__(self, lib, log):
        self._lib, self._log = lib, log

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        return _Timed(name, fn, self._log) if name.startswith("genie_") else fn


log = []
engine._lib.load = lambda: _Proxy(lib, log)
for k, s in enumerate(samples[:2]):
    log.clear()
    t0 = time.perf_counter()
    out = net(*s)
    sum(o.sum() for o in out).backward()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    agg = {}
    for n, ms in log:
        a = agg.setdefault(n, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] = max(a[2], ms)
    print("--- sample %d with rebuild (every C call followed by a sync): %.2f ms wall, %.2f ms inside C calls" % (k, total, sum(m for _, m in log)))
    for n, (cnt, ms, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print("   %-40s x%-4d %7.3f ms (max %.3f)" % (n, cnt, ms, mx))
    log.clear()
    out = net(*s)
    sum(o.sum() for o in out).backward()
    torch.cuda.synchronize()
    agg = {}
    for n, ms in log:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += ms
    print("   second step on the same graphs: " + ", ".join("%s %.2f" % (n, ms) for n, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]))
