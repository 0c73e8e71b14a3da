"""Shared helpers for the parity tests: golden-fixture loading and oracle invocation."""
import os

import numpy as np
import torch

from genie_amd import graph as G
from oracle import genie_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["tiny_6x40", "cfg1_20x500", "odd_33x257"]
EDGES_CASES = ["edges_12x60", "edges_7x13"]      # use_updated_model_definition class (DataAggregationEdges)
SUBGRAPH_CASES = ["subgraph_14x50"]               # use_subgraph: irregular product graph
ABSPOS_CASES = ["abspos_12x60", "abspos_7x13"]    # use_absolute_pos: positions appended to the inputs


class Case(object):
    """One golden fixture: inputs, graphs, weights (state_dict names) and reference outputs."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.z = z
        self.name = name
        self.S, self.G = int(z["n_sta"]), int(z["n_grid"])
        self.row_stride = int(z["row_stride"])
        self.Slice = torch.from_numpy(z["Slice"].astype(np.float32))
        self.Mask = torch.from_numpy(z["Mask"].astype(np.float32))
        self.edge_attr = torch.from_numpy(z["edge_attr"].astype(np.float32))
        self.A_sta_sta = torch.from_numpy(z["A_sta_sta"]).long()
        self.A_src_src = torch.from_numpy(z["A_src_src"]).long()
        self.locs = torch.from_numpy(z["locs"])
        self.x_grid = torch.from_numpy(z["x_grid"])
        self.x_query = torch.from_numpy(z["x_query"])
        self.t_query = torch.from_numpy(z["t_query"])
        self.weights = O.weights_from_npz(z, torch.float32)
        self.edges_variant = self.weights["DataAggregation.l1_t1_2.weight"].shape[1] == 68
        self.abspos_variant = self.weights["DataAggregation.init_trns.weight"].shape[1] == 14

    def product_edges(self):
        if "pairs" in self.z.files:      # irregular product graph (use_subgraph, process_utils.py:744-849)
            pairs = torch.from_numpy(self.z["pairs"]).long()
            A1, A2, A_src_in_prod = G.subgraph_product_edges(self.A_sta_sta, self.A_src_src, pairs.numpy())
            return A1, A2, A_src_in_prod, pairs
        return G.cartesian_product_edges(self.A_sta_sta, self.A_src_src, self.S, self.G)

    def tables(self):
        return G.neighbour_table(self.A_sta_sta, self.S), G.neighbour_table(self.A_src_src, self.G)

    def ref(self, key):
        return torch.from_numpy(np.asarray(self.z[key]))

    def strided(self, t):
        """Apply the fixture's row stride to a [P, C] tensor."""
        if t.shape[0] == self.S * self.G and self.row_stride > 1:
            return t[:: self.row_stride]
        return t

    def oracle_forward(self, dtype=torch.float32, structured=False):
        w = {k: v.to(dtype) for k, v in self.weights.items()}
        args = dict(full=True)
        if structured and not self.edges_variant and not self.abspos_variant:
            sta_nbr, src_nbr = self.tables()
            return O.forward_fixed_source_structured(
                w, self.Slice.to(dtype), self.Mask.to(dtype), sta_nbr, src_nbr, self.edge_attr.to(dtype),
                self.A_src_src, self.x_grid.to(dtype), self.x_query.to(dtype), self.t_query.to(dtype),
                self.S, self.G, **args)
        A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = self.product_edges()
        Slice = self.Slice.to(dtype)
        if self.abspos_variant:
            Slice = O.absolute_pos_inputs(Slice, self.locs.to(dtype), self.x_grid.to(dtype), A_src_in_sta)
        if self.edges_variant:
            assert not structured
            args["pos_rel"] = (O.edge_pos_features(self.locs.to(dtype), A_in_sta, A_src_in_sta[0]),
                               O.edge_pos_features(self.x_grid.to(dtype), A_in_src, A_src_in_sta[1]))
        return O.forward_fixed_source(
            w, Slice, self.Mask.to(dtype), A_in_sta, A_in_src, self.edge_attr.to(dtype),
            A_src_in_prod, self.A_src_src, self.x_grid.to(dtype), self.x_query.to(dtype),
            self.t_query.to(dtype), **args)


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max())
