"""CPU: the 4-output forward_fixed (association heads, module.py:963-997) — oracle restatement against the golden
vector produced by the reference's own forward_fixed (tests/golden/assoc_7x45.npz)."""
import os

import numpy as np
import pytest
import torch

from genie_amd import graph
from oracle import genie_oracle as O
from tests.util import GOLDEN_DIR, max_abs


# round 3: 20 stations (uniform degree 8: the pipelined kernels), 270 picks on one station, one station without picks; `_nonull`:
# no candidate source with |stime| < 2 eps, so `edge_index[0].max()` (module.py:762-763) is a real pick, not the null pick
ASSOC_CASES = ["assoc_7x45", "assoc_20x60", "assoc_20x60_nonull"]


def load(name="assoc_7x45"):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    w = O.weights_from_npz(z)
    t = lambda k, dt=torch.float32: torch.from_numpy(np.asarray(z[k])).to(dt)
    return z, w, t


@pytest.mark.parametrize("name", ASSOC_CASES)
def test_oracle_forward_fixed_matches_reference(name):
    z, w, t = load(name)
    S, G = int(z["n_sta"]), int(z["n_grid"])
    A_in_sta, A_in_src, A_src_in_prod, _ = graph.cartesian_product_edges(z["A_sta_sta"], z["A_src_src"], S, G)
    y, x, arv_p, arv_s = O.forward_fixed(
        w, t("Slice"), t("Mask"), A_in_sta, A_in_src, t("edge_attr"), A_src_in_prod, t("A_src_src", torch.long),
        t("A_edges_p", torch.long), t("A_edges_s", torch.long), t("dt_partition"), t("tlatent"), t("tpick"), t("ipick", torch.long),
        t("phase_label"), t("x_grid"), t("x_query"), t("x_query_src"), t("t_query"), t("tq_sample"), t("trv_out_q"), S)
    assert max_abs(y, t("y")) <= 1e-6 and max_abs(x, t("x")) <= 1e-6
    assert arv_p.shape == tuple(z["arv_p"].shape) and arv_s.shape == tuple(z["arv_s"].shape)
    assert max_abs(arv_p, t("arv_p")) <= 2e-6
    assert max_abs(arv_s, t("arv_s")) <= 2e-6
    assert float(t("arv_p").abs().max()) > 1e-2          # non-trivial fixture


@pytest.mark.parametrize("name", ASSOC_CASES)
def test_product_heads_match_reference_given_oracle_front(name):
    """The PyTorch association heads shipped in genie_amd/module.py (CPU run: what training steps differentiate), fed with the
    oracle's front-end intermediates, reproduce the reference's arv_p / arv_s."""
    from genie_amd import module
    z, w, t = load(name)
    S, G = int(z["n_sta"]), int(z["n_grid"])
    A_in_sta, A_in_src, A_src_in_prod, _ = graph.cartesian_product_edges(z["A_sta_sta"], z["A_src_src"], S, G)
    o = O.forward_fixed_source(w, t("Slice"), t("Mask"), A_in_sta, A_in_src, t("edge_attr"), A_src_in_prod,
                               t("A_src_src", torch.long), t("x_grid"), t("x_query"), t("t_query"), full=True)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu")
    net.load_state_dict({k: v.clone() for k, v in w.items()}, strict=True)
    sta_tab = graph.neighbour_table(z["A_sta_sta"], S).long()
    src_tab = graph.neighbour_table(z["A_src_src"], G).long()
    with torch.no_grad():
        x_src = net.SpatialAttention(o["sa3"], t("x_query_src"), t("x_grid"))
        mask_out = 1.0 * (o["y"][:, :, 0].max(1, keepdim=True)[0] > 0.01)
        s, m1 = net.BipartiteGraphReadOutOperator(o["y_latent"], t("edge_attr"), mask_out, S)
        s = net.DataAggregationAssociationPhase(s, o["x_latent"], m1, t("Mask"), sta_tab, src_tab, S, G)
        tl = t("tlatent")
        arv_p = net.LocalSliceLgCollapseP(t("A_edges_p", torch.long), t("dt_partition"), t("tpick"), t("ipick", torch.long),
                                          t("phase_label"), s, tl[:, 0:1])
        arv_s = net.LocalSliceLgCollapseS(t("A_edges_s", torch.long), t("dt_partition"), t("tpick"), t("ipick", torch.long),
                                          t("phase_label"), s, tl[:, 1:2])
        arv = net.Arrivals(int(z["tq_sample"].shape[0]), t("tq_sample"), x_src, t("trv_out_q"), arv_p, arv_s, t("tpick"), t("ipick", torch.long), t("phase_label"))
    assert max_abs(arv[:, :, 0:1], t("arv_p")) <= 2e-6
    assert max_abs(arv[:, :, 1:2], t("arv_s")) <= 2e-6


def test_time_pointers_match_the_reference_tables():
    """genie_amd.graph.time_pointers restates utils.py:602-622; the fixture's tables came from the reference function itself."""
    z, w, t = load()
    S, G = int(z["n_sta"]), int(z["n_grid"])
    trv = np.asarray(z["tlatent"]).reshape(G, S, 2)
    ep, es, dtp = graph.time_pointers(trv, max_t=float(z["max_t"]), dt=3.0 / 5.0, k=10, win=6.0)
    assert np.allclose(dtp, z["dt_partition"])
    assert ep.shape == z["A_edges_p"].shape and np.array_equal(ep, z["A_edges_p"]) and np.array_equal(es, z["A_edges_s"])


def test_station_pick_pairs_match_the_reference_construction():
    """module.station_pick_pairs against the statement of module.py:703-713 (numpy meshgrid per station)."""
    from genie_amd import module
    rng = np.random.default_rng(9)
    for n, n_sta in ((1, 3), (17, 4), (200, 23), (64, 1)):
        ip = rng.integers(0, n_sta, n)
        lists = [np.where(ip == u)[0] for u in np.unique(ip)]
        pairs = [np.stack(np.meshgrid(l, np.concatenate((l, [n])), indexing="ij"), 0).reshape(2, -1) for l in lists]
        want = np.ascontiguousarray(np.hstack(pairs)[::-1])
        got = module.station_pick_pairs(torch.from_numpy(ip))
        assert np.array_equal(got.numpy(), want), n
