#!/usr/bin/env python
"""Measurement lines for the two graph / model variants of the hot path, same shape as
config 2 (200 stations / 10 000 grid nodes / 50 000 picks): one JSON line each with ms per window, picks/s, the CPU oracle
timed on the same window and the max-abs difference of (y, x), and the time per product node relative to the default model on
the full product graph (same single-stream loop). Usage: python tools/bench_variants.py [default] [edges] [abspos] [subgraph]"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from genie_amd import graph, module, synthetic  # noqa: E402


def time_gpu(step, n_settle=300, n=200):
    for _ in range(n_settle):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def main():
    which = sys.argv[1:] or ["default", "edges", "abspos", "subgraph"]
    ns_default = None
    from oracle import genie_oracle as O
    S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
    geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
    win = synthetic.make_window(geom, n_picks, seed=2)
    dev = "cuda:0"
    locs, xg = torch.from_numpy(geom.locs).float().to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
    xq, tq = torch.from_numpy(geom.x_query).float().to(dev), torch.from_numpy(geom.t_query).float().to(dev)
    for v in which:
        torch.manual_seed(0)
        if v in ("default", "edges", "abspos"):        # Cartesian product graph: the default model, a-9 (DataAggregationEdges), use_absolute_pos
            net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev, use_updated_model_definition=(v == "edges"),
                                                        use_absolute_pos=(v == "abspos")).eval()
            net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                                     torch.from_numpy(geom.edge_attr()).to(dev), locs, xg)
            Slice, Mask = torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"])
            A1, A2, Ap, Asis = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
            ea = torch.from_numpy(geom.edge_attr())
            n_prod = S * G
            label = {"default": "default model definition", "edges": "use_updated_model_definition=True (DataAggregationEdges)",
                     "abspos": "use_absolute_pos=True"}[v] + ", full product graph"
        else:                   # f-4: use_subgraph, every source node keeps its 60 nearest stations
            net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev).eval()
            d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
            keep = np.zeros(d.shape, dtype=bool)
            keep[np.arange(G)[:, None], np.argsort(d, axis=1)[:, :60]] = True
            src_i, sta_i = np.nonzero(keep)
            pairs = np.stack((sta_i, src_i))
            A1, A2, Ap = graph.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
            Asis = torch.from_numpy(pairs).long()
            rows = src_i * S + sta_i
            Slice, Mask = torch.from_numpy(win["Slice"][rows]), torch.from_numpy(win["Mask"][rows])
            ea = torch.from_numpy(geom.edge_attr()[rows])
            ge = graph.GraphEdges(x=ea.to(dev), edge_index=Ap.to(dev))
            net.set_adjacencies(A1.to(dev), A2.to(dev), ge, ge, Asis.to(dev), torch.from_numpy(geom.A_src_src).to(dev),
                                None, None, None, None, locs, xg)
            n_prod, label = int(pairs.shape[1]), "use_subgraph=True, 60 nearest stations per source node (irregular product graph)"
        dS, dM = Slice.to(dev), Mask.to(dev)
        with torch.no_grad():
            dt, (y, x) = time_gpu(lambda: net.forward_fixed_source(dS, dM, None, None, None, locs, xg, xq, tq))
            # independent windows as in the apply loop: P-sized kernels per window, one G-sized tail per 16 windows on a side stream
            net.window_batch = 16

            def batch16():
                for _ in range(16):
                    net.push_window(dS, dM)
                return net.flush_windows(xg, xq, tq)
            dtb, (yb, xb, _ev) = time_gpu(batch16, n_settle=20, n=15)
            net._hip.wait_tails()
            torch.cuda.synchronize()
            dtb /= 16.0
            same = bool(torch.equal(yb[3], y) and torch.equal(xb[3], x))
            w = {k: t.detach().cpu() for k, t in net.state_dict().items()}
            kw = {}
            if v == "edges":
                kw["pos_rel"] = (O.edge_pos_features(torch.from_numpy(geom.locs).float(), A1, Asis[0]),
                                 O.edge_pos_features(torch.from_numpy(geom.x_grid).float(), A2, Asis[1]))
            if v == "abspos":
                Slice = O.absolute_pos_inputs(Slice, torch.from_numpy(geom.locs).float(), torch.from_numpy(geom.x_grid).float(), Asis)
            t0 = time.perf_counter()
            yc, xc = O.forward_fixed_source(w, Slice, Mask, A1, A2, ea, Ap, torch.from_numpy(geom.A_src_src),
                                            torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(),
                                            torch.from_numpy(geom.t_query).float(), **kw)
            cdt = time.perf_counter() - t0
        ns = dt * 1e9 / n_prod
        if v == "default":
            ns_default = ns
        print(json.dumps({
            "variant": v, "workload": "200 stations / 10000 grid nodes / 50000 picks per window, " + label, "n_product_nodes": n_prod,
            "ms_per_window": round(dt * 1e3, 4), "picks_per_s": round(n_picks / dt, 1), "single_stream": True,
            "ms_per_window_batched_tails": round(dtb * 1e3, 4), "picks_per_s_batched_tails": round(n_picks / dtb, 1),
            "batched_equals_single_stream_bitwise": same,
            "ns_per_product_node": round(ns, 4), "vs_default_per_product_node": round(ns / ns_default, 3) if ns_default else None,
            "cpu_oracle_s_per_window": round(cdt, 2), "cpu_cores": int(torch.get_num_threads()),
            "max_abs_y_vs_cpu": float((y.cpu() - yc).abs().max()), "max_abs_x_vs_cpu": float((x.cpu() - xc).abs().max())}), flush=True)
        del net
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
