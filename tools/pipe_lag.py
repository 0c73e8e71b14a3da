#!/usr/bin/env python
"""Experiment: where should the G-sized tail of window i run? (a) right after stage 2 of window i (overlaps stage 1 of
i+1: the shipped forward_pipelined), (b) after stage 1 of window i+1 (overlaps stage 2 of i+1), (c) no overlap."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from genie_amd import _lib, engine, synthetic  # noqa: E402
from genie_amd.engine import _ptr  # noqa: E402
from tests.util import Case  # noqa: E402


def main():
    cfg = "cfg2_200x10k"
    S, G, n_picks, L, nq = synthetic.CONFIGS[cfg]
    geom = synthetic.Geometry(S, G, L=L, n_query=G, seed=1)
    win = synthetic.make_window(geom, n_picks, seed=2)
    dev = "cuda:0"
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G),
                        grid_order=engine.morton_order(geom.x_grid), device=dev)
    hp.set_weights({k: v.to(dev) for k, v in Case("cfg1_20x500").weights.items()})
    Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
    ea, pos = torch.from_numpy(geom.edge_attr()).to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
    xq = pos.clone()
    from genie_amd.module import knn_query_edges  # noqa
    d = torch.cdist(xq.double(), pos.double())
    knn = d.topk(10, largest=False).indices.int().contiguous()
    tq = torch.arange(-4.5, 5.0, 1.0, device=dev)[:10].float()
    side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("SIDE_PRIO", "0")))
    main_s = torch.cuda.current_stream(dev)
    lib, ctx, ws = hp.lib, hp.ctx, hp._ws_ptr

    def s1(st):
        _lib.check(lib.genie_da_stage1(ctx, _ptr(Slice), _ptr(Mask), ws, st), "s1")

    def s2(st):
        _lib.check(lib.genie_da_stage2_partials(ctx, _ptr(Mask), _ptr(ea), None, ws, st), "s2")

    def tail(stream):
        with torch.cuda.stream(stream):
            ss = ctypes.c_void_p(stream.cuda_stream)
            bip = torch.empty((G, 15), dtype=torch.float32, device=dev)
            xs = torch.empty((G, 30), dtype=torch.float32, device=dev)
            _lib.check(lib.genie_bipartite_readout(ctx, _ptr(bip), ws, ss), "bip")
            _lib.check(lib.genie_spatial_agg3_fwd(ctx, _ptr(bip), _ptr(pos), _ptr(xs), ws, ss), "sa")
            y = hp.readout_grid(xs, tq)
            x = hp.readout_query(xs, pos, xq, knn, tq)
        return y, x

    def run(mode, n=200):
        slot = 0
        ev_tail = [None, None]
        pending = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            lib.genie_set_slot(ctx, slot)
            st = ctypes.c_void_p(main_s.cuda_stream)
            if ev_tail[slot] is not None:
                main_s.wait_event(ev_tail[slot])
            s1(st)
            if mode == "b" and pending is not None:
                e = torch.cuda.Event(); e.record(main_s)
                side.wait_event(e)
                lib.genie_set_slot(ctx, pending)
                tail(side)
                d_ = torch.cuda.Event(); d_.record(side)
                ev_tail[pending] = d_
                lib.genie_set_slot(ctx, slot)
            s2(st)
            if mode == "a":
                e = torch.cuda.Event(); e.record(main_s)
                side.wait_event(e)
                tail(side)
                d_ = torch.cuda.Event(); d_.record(side)
                ev_tail[slot] = d_
            elif mode == "b":
                pending = slot
            else:
                tail(main_s)
            slot ^= 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for mode in ("c", "a", "b", "a", "b"):
        run(mode, 20)
        print(mode, "%.4f ms/window" % run(mode), flush=True)


if __name__ == "__main__":
    main()
