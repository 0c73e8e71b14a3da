#!/usr/bin/env python
"""max|y - ref|, max|x - ref| of the HIP forward_fixed_source on the golden fixtures, against the reference's fp32 and fp64 runs.
Usage: python tools/fixture_err.py [fixture ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genie_amd import graph, module  # noqa: E402
from tests.util import Case, max_abs  # noqa: E402

dev = "cuda:0"
for name in sys.argv[1:] or ["tiny_6x40", "cfg1_20x500", "odd_33x257", "o1_20x500"]:
    c = Case(name)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=dev)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(dev), edge_index=A_src_in_prod.to(dev))
    net.set_adjacencies(A_in_sta.to(dev), A_in_src.to(dev), ea, ea, A_src_in_sta.to(dev), c.A_src_src.to(dev),
                        None, None, None, None, c.locs.float().to(dev), c.x_grid.float().to(dev))
    if os.environ.get("TAIL") == "f32":
        net._hip.set_tail_precision(False)
    with torch.no_grad():
        y, x = net.forward_fixed_source(c.Slice.to(dev), c.Mask.to(dev), None, None, None, c.locs.float().to(dev),
                                        c.x_grid.float().to(dev), c.x_query.float().to(dev), c.t_query.float().to(dev))
    y, x = y.cpu(), x.cpu()
    print("%-14s max|y| %.3g  vs fp32 ref: y %.3e x %.3e   vs fp64 ref: y %.3e x %.3e   (fp32 ref vs fp64 ref: y %.3e x %.3e)" % (
        name, float(c.ref("y").abs().max()), max_abs(y, c.ref("y")), max_abs(x, c.ref("x")), max_abs(y, c.ref("y64")), max_abs(x, c.ref("x64")),
        max_abs(c.ref("y"), c.ref("y64")), max_abs(c.ref("x"), c.ref("x64"))))
