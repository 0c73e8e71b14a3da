#!/usr/bin/env python
"""Stage-by-stage comparison of the eval forward_fixed (HIP) with the oracle on an association fixture: assoc_dbg.py <name>"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import graph, module, engine as E  # noqa
from oracle import genie_oracle as O  # noqa
name = sys.argv[1] if len(sys.argv) > 1 else "assoc_20x60"
z = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
w = O.weights_from_npz(z)
S, G = int(z["n_sta"]), int(z["n_grid"])
dev = "cuda:0"
c = lambda k, dt=torch.float32: torch.from_numpy(np.asarray(z[k])).to(dt)
t = lambda k, dt=torch.float32: c(k, dt).to(dev)
A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(z["A_sta_sta"], z["A_src_src"], S, G)
o = O.forward_fixed_source(w, c("Slice"), c("Mask"), A_in_sta, A_in_src, c("edge_attr"), A_src_in_prod, c("A_src_src", torch.long), c("x_grid"),
                           c("x_query"), c("t_query"), full=True)
x_src_o = O.spatial_attention(w, o["sa3"], c("x_query_src"), c("x_grid"))
mask_out = 1.0 * (o["y"][:, :, 0].max(1, keepdim=True)[0] > 0.01)
s0, m1 = O.bipartite_read_out(w, o["y_latent"], c("edge_attr"), mask_out, S)
s_o = O.data_aggregation_association(w, s0, o["x_latent"], m1, c("Mask"), A_in_sta, A_in_src)
tl = c("tlatent")
ap_o = O.local_slice_collapse(w, c("A_edges_p", torch.long), c("dt_partition"), c("tpick"), c("ipick", torch.long), c("phase_label"), s_o, tl[:, 0:1], "LocalSliceLgCollapseP")
as_o = O.local_slice_collapse(w, c("A_edges_s", torch.long), c("dt_partition"), c("tpick"), c("ipick", torch.long), c("phase_label"), s_o, tl[:, 1:2], "LocalSliceLgCollapseS")
arv_o = O.station_source_attention(w, c("x_query_src").shape[0], c("tq_sample"), x_src_o, c("trv_out_q"), ap_o, as_o, c("tpick"), c("ipick", torch.long), c("phase_label"))
print("oracle vs fixture arv_p %.2e" % float((arv_o[:, :, 0:1] - c("arv_p")).abs().max()))
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
net.load_state_dict({k: v.clone() for k, v in w.items()}, strict=True)
net.eval()
ea = graph.GraphEdges(x=t("edge_attr"), edge_index=A_src_in_prod.to(dev))
net.set_adjacencies(A_in_sta.to(dev), A_in_src.to(dev), ea, ea, A_src_in_sta.to(dev), t("A_src_src", torch.long), t("A_edges_p", torch.long),
                    t("A_edges_s", torch.long), t("dt_partition"), t("tlatent"), t("locs"), t("x_grid"))
hp = net._hip
d = lambda a, b: float((a.cpu() - b).abs().max())
with torch.no_grad():
    xs, xl, _ = net._path(t("Slice"), t("Mask"), t("x_grid"), want_x_latent=True)
    y, ylat = hp.readout_grid_latent(xs, t("t_query"))
    print("x_spatial %.2e x_latent %.2e y %.2e y_latent %.2e" % (d(xs, o["sa3"]), d(xl, o["x_latent"]), d(y, o["y"]), d(ylat, o["y_latent"])))
    knn_src = E.knn_device(t("x_grid"), t("x_query_src"), 10)
    x_src = hp.spatial_attention(xs, t("x_grid"), t("x_query_src"), knn_src, t("t_query"))
    print("x_src %.2e" % d(x_src, x_src_o))
    mo = 1.0 * (y[:, :, 0].max(1, keepdim=True)[0] > 0.01)
    s = hp.assoc_fwd(ylat, mo, xl, t("Mask"), t("edge_attr"))
    print("assoc s %.2e (scale %.2e)" % (d(s, s_o), float(s_o.abs().max())))
    # heads fed with ORACLE inputs, so that every stage is judged on its own
    a32 = lambda k: t(k, torch.long).to(torch.int32).contiguous()
    ap = hp.lslc_fwd(0, s_o.to(dev), a32("A_edges_p"), t("dt_partition"), t("tpick"), a32("ipick"), t("phase_label"), t("tlatent"), 0, 15.0)
    as_ = hp.lslc_fwd(1, s_o.to(dev), a32("A_edges_s"), t("dt_partition"), t("tpick"), a32("ipick"), t("phase_label"), t("tlatent"), 1, 15.0)
    print("lslc p %.2e s %.2e" % (d(ap, ap_o), d(as_, as_o)))
    arv = hp.arrivals_fwd(t("tq_sample"), x_src_o.to(dev), t("trv_out_q"), ap_o.to(dev), as_o.to(dev), t("tpick"), t("ipick", torch.long), t("phase_label"), 15.0)
    e = (arv.cpu() - arv_o).abs()
    print("arrivals %.2e" % float(e.max()), "worst (src, pick):", np.unravel_index(int(e.max(2)[0].argmax()), e.shape[:2]),
          "ipick there:", int(z["ipick"][int(e.max(2)[0].max(0)[0].argmax())]))
    bad = (e.max(2)[0].max(0)[0] > 1e-4).nonzero().reshape(-1)
    print("bad picks:", bad.numel(), "stations of bad picks:", sorted(set(z["ipick"][bad.numpy()].tolist())))
    ip = z["ipick"]
    order = np.argsort(ip, kind="stable")
    pos_in_sta = {int(b): int(np.where(order[ip[order] == ip[b]] == b)[0][0]) for b in bad.numpy()}
    print("positions of the bad picks in their station list:", sorted(pos_in_sta.values()))
    print("station 3 count:", int((ip == 3).sum()), "error by source:", e.max(2)[0].max(1)[0].tolist())
