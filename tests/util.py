"""Shared helpers for the parity tests: golden-fixture loading and oracle invocation."""
import os

import numpy as np
import torch

from genie_amd import graph as G
from oracle import genie_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["tiny_6x40", "cfg1_20x500", "cfg1e_20x500", "odd_33x257", "o1_20x500", "s2000_2000x24"]
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


def oracle_bipartite_for_nodes(w, geom, picks, sample, t0=0.0):
    """Bipartite read-in output [len(sample), 15] of the source nodes `sample` of a LARGE problem, by the oracle's arithmetic
    on their two-hop source neighbourhood only (DataAggregation needs the neighbours' `v`, which needs THEIR neighbours' h0):
    module.py:85-98, :224-229 restated on the base kNN tables. Inputs are embedded for just those nodes
    (genie_amd.synthetic.make_slice_mask with g_slice)."""
    from genie_amd import synthetic
    S, Gn = geom.n_sta, geom.n_grid
    src_tab = G.neighbour_table(geom.A_src_src, Gn).numpy()
    sta_nbr = G.neighbour_table(geom.A_sta_sta, S)
    sample = np.asarray(sample, dtype=np.int64)
    n1 = np.unique(np.concatenate([sample, src_tab[sample].reshape(-1)]))
    n2 = np.unique(np.concatenate([n1, src_tab[n1].reshape(-1)]))
    pos2 = -np.ones(Gn, dtype=np.int64)
    pos2[n2] = np.arange(n2.size)
    Sl, Mk = synthetic.make_slice_mask(geom, picks, t0, g_slice=n2)
    Sl, Mk = torch.from_numpy(Sl), torch.from_numpy(Mk)
    pre = "DataAggregation"
    h0 = O.act(O.linear(torch.cat((Sl, Mk), -1), w, pre + ".init_trns"), w, pre + ".activate").view(n2.size, S, -1)
    i1 = torch.from_numpy(pos2[n1])
    nb1 = torch.from_numpy(pos2[src_tab[n1]])                                  # [n1, kp] positions in n2
    M1 = Mk.view(n2.size, S, -1)[i1].reshape(n1.size * S, -1)
    h0_1 = h0[i1]
    a1 = O._gather_mean_sta(O.act(h0_1, w, pre + ".activate11"), sta_nbr).reshape(n1.size * S, -1)
    a2 = O.act(h0, w, pre + ".activate12")[nb1].mean(dim=1).reshape(n1.size * S, -1)
    h0f = h0_1.reshape(n1.size * S, -1)
    h1 = O.act(torch.cat((O.linear(torch.cat((h0f, a1, M1), 1), w, pre + ".l1_t1_2"),
                          O.linear(torch.cat((h0f, a2, M1), 1), w, pre + ".l1_t2_2")), 1), w, pre + ".activate1")
    u = O.act(O.linear(h1, w, pre + ".l2_t1_1"), w, pre + ".activate21").view(n1.size, S, -1)
    v = O.act(O.linear(h1, w, pre + ".l2_t2_1"), w, pre + ".activate22").view(n1.size, S, -1)
    pos1 = -np.ones(Gn, dtype=np.int64)
    pos1[n1] = np.arange(n1.size)
    i0 = torch.from_numpy(pos1[sample])
    nb0 = torch.from_numpy(pos1[src_tab[sample]])
    ns = sample.size
    M0 = M1.view(n1.size, S, -1)[i0].reshape(ns * S, -1)
    h1_0 = h1.view(n1.size, S, -1)[i0].reshape(ns * S, -1)
    b1 = O._gather_mean_sta(u[i0], sta_nbr).reshape(ns * S, -1)
    b2 = v[nb0].mean(dim=1).reshape(ns * S, -1)
    x_latent = O.act(torch.cat((O.linear(torch.cat((h1_0, b1, M0), 1), w, pre + ".l2_t1_2"),
                                O.linear(torch.cat((h1_0, b2, M0), 1), w, pre + ".l2_t2_2")), 1), w, pre + ".activate2")
    ea = torch.from_numpy(geom.edge_attr(sample))
    return O.bipartite_read_in_structured(w, x_latent, ea, M0, S, ns), x_latent
