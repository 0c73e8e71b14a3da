#!/usr/bin/env python
"""Does stage 2 of window i overlap stage 1 of window i + 1 when they run on two streams (stage 1 is bound by vector / matrix
issue, stage 2 by the vector-memory path)? Sequential pairs vs the two-stream schedule, HIP events over N windows.
Usage: python tools/overlap_probe.py [config] [windows]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import _lib, engine, synthetic  # noqa: E402
from tests.util import Case  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_200x10k"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    S, G, n_picks, L, nq = synthetic.CONFIGS[cfg]
    geom = synthetic.Geometry(S, G, L=L, n_query=10, seed=1)
    win = synthetic.make_window(geom, n_picks, seed=2)
    dev = "cuda:0"
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G),
                        grid_order=engine.sfc_order(geom.x_grid), device=dev, sta_order=engine.sfc_order(geom.locs))
    hp.set_weights({k: v.to(dev) for k, v in Case("cfg1_20x500").weights.items()})
    Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
    ea = torch.from_numpy(geom.edge_attr()).to(dev)
    hp.set_static_edge_attr(ea)
    slot = lambda k: _lib.check(hp.lib.genie_set_slot(hp.ctx, k), "genie_set_slot")
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()

    def sequential(count):
        for i in range(count):
            slot(i % 2)
            hp.da_stage1(Slice, Mask)
            hp.da_stage2_partials_range(Mask, ea, 0, G)

    def two_streams(count):
        done1 = [None, None]      # stage 1 of the window in slot k finished (main stream)
        done2 = [None, None]      # stage 2 of the window in slot k finished (side stream)
        for i in range(count):
            k = i % 2
            if done2[k] is not None:
                main_s.wait_event(done2[k])            # the rows of slot k are free again
            slot(k)
            hp.da_stage1(Slice, Mask)
            done1[k] = torch.cuda.Event(); done1[k].record(main_s)
            side.wait_event(done1[k])
            with torch.cuda.stream(side):
                slot(k)
                hp.da_stage2_partials_range(Mask, ea, 0, G)
                done2[k] = torch.cuda.Event(); done2[k].record(side)
        main_s.wait_stream(side)

    for name, fn in (("sequential", sequential), ("two streams", two_streams), ("sequential", sequential), ("two streams", two_streams)):
        fn(300)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(n)
        e1.record()
        torch.cuda.synchronize()
        print("%-12s %.4f ms per window (stage 1 + stage 2, %s)" % (name, e0.elapsed_time(e1) / n, cfg))
    slot(0)


if __name__ == "__main__":
    main()
