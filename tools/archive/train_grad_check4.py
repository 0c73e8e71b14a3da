#!/usr/bin/env python
"""Per-parameter gradients of the 4-output training call (net(*22 tensors) in train() mode) against the oracle's autograd on an
association fixture: train_grad_check4.py [fixture]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from genie_amd import graph, module  # noqa
from oracle import genie_oracle as O  # noqa
name = sys.argv[1] if len(sys.argv) > 1 else "assoc_20x60"
z = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
w0 = O.weights_from_npz(z)
S, G = int(z["n_sta"]), int(z["n_grid"])
DEV = "cuda:0"
t = lambda k, dt=torch.float32, dev=DEV: torch.from_numpy(np.asarray(z[k])).to(dt).to(dev)
net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
net.train()
A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(z["A_sta_sta"], z["A_src_src"], S, G)
ea = graph.GraphEdges(x=t("edge_attr"), edge_index=A_src_in_prod.to(DEV))
ea_flip = graph.GraphEdges(x=t("edge_attr"), edge_index=A_src_in_prod.flip(0).contiguous().to(DEV))
graphs = (A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea_flip, A_src_in_sta.to(DEV), t("A_src_src", torch.long),
          t("A_edges_p", torch.long), t("A_edges_s", torch.long), t("dt_partition"), t("tlatent"))
tail = (t("tpick"), t("ipick", torch.long), t("phase_label"), t("locs"), t("x_grid"), t("x_query"), t("x_query_src"),
        t("t_query"), t("tq_sample"), t("trv_out_q"))
outs = net(t("Slice"), t("Mask"), *graphs, *tail)
for o, k in zip(outs, ("y", "x", "arv_p", "arv_s")):
    print(k, "vs fixture %.2e" % float((o.detach().cpu() - torch.from_numpy(z[k])).abs().max()))
g = torch.Generator().manual_seed(11)
coef = [torch.randn(o.shape, generator=g) for o in outs]
sum((o * c_.to(DEV)).sum() for o, c_ in zip(outs, coef)).backward()
c = lambda k, dt=torch.float32: t(k, dt, "cpu")
w = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
ref = O.forward_fixed(w, c("Slice"), c("Mask"), A_in_sta, A_in_src, c("edge_attr"), A_src_in_prod, c("A_src_src", torch.long),
                      c("A_edges_p", torch.long), c("A_edges_s", torch.long), c("dt_partition"), c("tlatent"), c("tpick"),
                      c("ipick", torch.long), c("phase_label"), c("x_grid"), c("x_query"), c("x_query_src"), c("t_query"),
                      c("tq_sample"), c("trv_out_q"), S)
sum((o * c_).sum() for o, c_ in zip(ref, coef)).backward()
bad = 0
for k, p in net.named_parameters():
    if w[k].grad is None:
        continue
    if p.grad is None:
        print("%-60s MISSING" % k); bad += 1; continue
    sc = max(1e-12, float(w[k].grad.abs().max()))
    err = float((p.grad.cpu() - w[k].grad).abs().max())
    flag = "" if err <= 2e-5 * sc else ("  <-- BAD" if err > 2e-4 * sc else "  <-- marginal")
    bad += flag.endswith("BAD")
    if flag or k.startswith(("Bipartite", "DataAggregationAssoc")):
        print("%-60s |g| %.3e err %.3e rel %.2e%s" % (k, sc, err, err / sc, flag))
print("bad:", bad)
