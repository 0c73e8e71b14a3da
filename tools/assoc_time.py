#!/usr/bin/env python
"""Time forward_fixed (4 outputs) at config 2 and its PyTorch-ROCm association heads separately."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from genie_amd import module, synthetic, engine as _engine  # noqa

def main():
    S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
    geom = synthetic.Geometry(S, G, L=L, n_query=nq, seed=1)
    dev = "cuda:0"
    torch.manual_seed(0)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev).eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                             torch.from_numpy(geom.edge_attr()).to(dev), torch.from_numpy(geom.locs).float().to(dev),
                             torch.from_numpy(geom.x_grid).float().to(dev))
    win = synthetic.make_window(geom, n_picks, seed=2)
    Slice, Mask = torch.from_numpy(win["Slice"]).to(dev), torch.from_numpy(win["Mask"]).to(dev)
    xg = torch.from_numpy(geom.x_grid).float().to(dev)
    P = S * G
    from genie_amd import graph
    net._sta_tab = graph.neighbour_table(geom.A_sta_sta, S).long().to(dev)
    net._src_tab = graph.neighbour_table(geom.A_src_src, G).long().to(dev)
    with torch.no_grad():
        x_spatial, x_latent, _ = net._path(Slice, Mask, xg, want_x_latent=True)
        y_latent = net.SpatialDirect(x_spatial)
        mask_out = (torch.rand(G, 1, device=dev) > 0.5).float()
        outs = {}
        for name, hip in (("torch gathers", None), ("HIP neighbour means", net._hip), ("HIP means + blocked GEMMs", net._hip)):
            if name == "HIP neighbour means":
                os.environ["GENIE_ASSOC_PLAIN"] = "1"
            else:
                os.environ.pop("GENIE_ASSOC_PLAIN", None)
            def heads():
                s, m1 = net.BipartiteGraphReadOutOperator(y_latent, net._edge_attr, mask_out, S)
                return net.DataAggregationAssociationPhase(s, x_latent, m1, Mask, net._sta_tab, net._src_tab, S, G, hip=hip)
            for _ in range(2): outs[name] = heads()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): heads()
            torch.cuda.synchronize()
            print("P-sized association heads, %s: %.2f ms" % (name, (time.perf_counter() - t0) / 5 * 1e3))
        net._hip.sync_weights(net._path_params)
        for _ in range(2): outs["genie_assoc_fwd"] = net._hip.assoc_fwd(y_latent, mask_out, x_latent, Mask, net._edge_attr)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net._hip.assoc_fwd(y_latent, mask_out, x_latent, Mask, net._edge_attr)
        torch.cuda.synchronize()
        print("P-sized association heads, genie_assoc_fwd (HIP): %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
        a_ = outs["torch gathers"]
        for k in ("HIP neighbour means", "HIP means + blocked GEMMs", "genie_assoc_fwd"):
            print("%s: max|diff| %.3e  max|ref| %.3e" % (k, float((a_ - outs[k]).abs().max()), float(a_.abs().max())))

if __name__ == "__main__":
    main()
