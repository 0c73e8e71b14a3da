#!/usr/bin/env python
"""Stage times on an irregular product graph (use_subgraph, 200 x 10 000 with the 60 nearest stations per source node): HIP-event
medians of stage 1 and stage 2 (+ the segmented Bipartite sum), with the f16x2 kernels (k_stage1_h2<PCSR>, k_stage2_h2p) and with
stage_precision="f32" (k_stage1_pcsr, k_stage2_pcsr). Usage: python tools/s2p_time.py"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import engine, graph, module, synthetic  # noqa


def main():
    S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
    geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
    win = synthetic.make_window(geom, n_picks, seed=2)
    dev = "cuda:0"
    locs, xg = torch.from_numpy(geom.locs).float().to(dev), torch.from_numpy(geom.x_grid).float().to(dev)
    d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
    keep = np.zeros(d.shape, dtype=bool)
    keep[np.arange(G)[:, None], np.argsort(d, axis=1)[:, :60]] = True
    src_i, sta_i = np.nonzero(keep)
    pairs = np.stack((sta_i, src_i))
    A1, A2, Ap = graph.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
    rows = src_i * S + sta_i
    Slice, Mask = torch.from_numpy(win["Slice"][rows]).to(dev), torch.from_numpy(win["Mask"][rows]).to(dev)
    ea = torch.from_numpy(geom.edge_attr()[rows]).to(dev)
    ge = graph.GraphEdges(x=ea, edge_index=Ap.to(dev))
    out = {}
    for mode in ("auto", "f32"):
        engine.STAGE_PRECISION = mode
        torch.manual_seed(0)
        net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev).eval()
        net.set_adjacencies(A1.to(dev), A2.to(dev), ge, ge, torch.from_numpy(pairs).long().to(dev), torch.from_numpy(geom.A_src_src).to(dev),
                            None, None, None, None, locs, xg)
        hp = net._hip
        with torch.no_grad():
            net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, torch.from_numpy(geom.x_query).float().to(dev),
                                     torch.from_numpy(geom.t_query).float().to(dev))
            t1, t2 = [], []
            for i in range(60):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(); hp.da_stage1(Slice, Mask); e[1].record()
                xl, bip = hp.da_stage2_bipartite(Mask, ea, want_x_latent=(i == 59)); e[2].record()
                torch.cuda.synchronize()
                if i >= 10:
                    t1.append(e[0].elapsed_time(e[1])); t2.append(e[1].elapsed_time(e[2]))
        out[mode] = (xl, bip)
        print("%s: stage 1 %.4f ms, stage 2 + segmented sum %.4f ms (P = %d product nodes)" % (mode, np.median(t1), np.median(t2), rows.size), flush=True)
    print("x_latent equal bit for bit: %s; max |d Bipartite| %.3e (scale %.3e)" % (
        bool(torch.equal(out["auto"][0], out["f32"][0])) if False else "n/a (stage 1 differs)",
        float((out["auto"][1] - out["f32"][1]).abs().max()), float(out["f32"][1].abs().max())))


if __name__ == "__main__":
    main()
