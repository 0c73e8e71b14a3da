"""CPU: host logic — state_dict compatibility, graph layout contract, C-ABI symbols, failure behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from genie_amd import _lib, engine, graph, module, synthetic
from tests.util import Case, GOLDEN_CASES


def test_state_dict_keys_and_shapes_match_reference():
    """The golden fixtures carry the reference model's full state_dict (158 tensors, 65 311 parameters)."""
    c = Case("tiny_6x40")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu")
    sd = net.state_dict()
    ref = c.weights
    assert list(sd.keys()) == list(ref.keys())
    assert len(sd) == 158
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert sum(p.numel() for p in net.parameters()) == 65311
    net.load_state_dict({k: v.clone() for k, v in ref.items()}, strict=True)


def test_state_dict_of_updated_model_definition_matches_reference():
    """`use_updated_model_definition: True` (config.yaml:95): the fixture holds the state_dict of the reference class of
    module.py:1022 (DataAggregationEdges: l1_t?_2 [30,68], l2_t?_2 [15,98])."""
    c = Case("edges_12x60")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu", use_updated_model_definition=True)
    sd = net.state_dict()
    assert list(sd.keys()) == list(c.weights.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(c.weights[k].shape), k
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    view = module._split_edge_columns({k: v for k, v in c.weights.items()})
    W = c.weights["DataAggregation.l2_t1_2.weight"]
    assert torch.equal(view["DataAggregation.l2_t1_2.weight"], torch.cat((W[:, :90], W[:, 94:]), 1))
    assert torch.equal(view["DataAggregation.l2_t1_2.weight_pos"], W[:, 90:94])


def test_state_dict_with_absolute_pos_matches_reference():
    """`use_absolute_pos: True` (config.yaml:92): init_trns [30,14], l1_t2_1 [30,10], association init_trns [30,56]."""
    c = Case("abspos_12x60")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu", use_absolute_pos=True)
    sd = net.state_dict()
    assert list(sd.keys()) == list(c.weights.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(c.weights[k].shape), k
    view = module._split_abs_columns(c.weights)
    W = c.weights["DataAggregation.init_trns.weight"]
    assert torch.equal(view["DataAggregation.init_trns.weight"], torch.cat((W[:, :4], W[:, 10:]), 1))
    assert torch.equal(view["DataAggregation.init_trns.weight_abs"], W[:, 4:10])


def test_subgraph_product_edges_match_the_reference_builder():
    """genie_amd.graph.subgraph_product_edges against the output of the reference's extract_inputs_adjacencies_subgraph
    (process_utils.py:744-849; fixture written by oracle/make_golden.py --subgraph): same edge sets, same A_src_in_prod."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subgraph_builder_14x50.npz"))
    A1, A2, Ap = graph.subgraph_product_edges(z["A_sta_sta"], z["A_src_src"], z["A_src_in_sta"])
    assert set(map(tuple, A1.numpy().T.tolist())) == set(map(tuple, z["A_prod_sta_sta"].T.tolist()))
    assert set(map(tuple, A2.numpy().T.tolist())) == set(map(tuple, z["A_prod_src_src"].T.tolist()))
    assert np.array_equal(Ap.numpy(), z["A_src_in_prod"])
    assert z["A_src_in_sta"].shape[1] < 14 * 50


def test_library_exports_every_declared_symbol(repo_root):
    _lib.build()
    lib = _lib.load()
    header = open(os.path.join(repo_root, "include", "genie_hip.h")).read()
    declared = set(re.findall(r"\b(genie_[a-z0-9_]+)\s*\(", header))
    declared.discard("genie_ctx")
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.genie_version() >= 100
    names = [lib.genie_weights_name(i).decode() for i in range(lib.genie_weights_count())]
    ref = Case("tiny_6x40").weights
    edges = module._split_edge_columns(Case("edges_12x60").weights)      # registry view of the DataAggregationEdges weights
    for n_, i in zip(names, range(len(names))):
        if n_.endswith(".weight_pos"):      # the 4 edge-feature columns of the use_updated_model_definition variant
            assert edges[n_].numel() == lib.genie_weights_numel(i)
            continue
        if n_.endswith(".weight_abs"):      # the 6 absolute-position columns of init_trns (use_absolute_pos)
            assert module._split_abs_columns(Case("abspos_12x60").weights)[n_].numel() == lib.genie_weights_numel(i)
            continue
        assert n_ in ref, n_
        assert ref[n_].numel() == lib.genie_weights_numel(i)
        if not n_.startswith(("BipartiteGraphReadOutOperator.", "DataAggregationAssociationPhase.")):
            assert lib.genie_weights_numel(i) == edges[n_].numel()     # (the Edges class has wider association heads: not served)
        assert lib.genie_weights_offset(i) % 4 == 0


def test_cartesian_edges_roundtrip():
    geom = synthetic.Geometry(7, 31, L=50e3, n_query=5, seed=3)
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, 7, 31)
    assert A_in_sta.shape == (2, 7 * 31 * geom.k_sta) and A_in_src.shape == (2, 7 * 31 * geom.k_spc)
    sta_nbr, src_nbr = graph.base_tables_from_product(A_in_sta, A_in_src, 7, 31)
    assert torch.equal(sta_nbr, graph.neighbour_table(geom.A_sta_sta, 7))
    assert torch.equal(src_nbr, graph.neighbour_table(geom.A_src_src, 31))
    # every product edge stays inside one source node (sta graph) / one station (src graph)
    assert torch.equal(A_in_sta[0] // 7, A_in_sta[1] // 7)
    assert torch.equal(A_in_src[0] % 7, A_in_src[1] % 7)
    assert torch.equal(A_src_in_prod[1], torch.arange(7 * 31) // 7)
    assert torch.equal(A_src_in_sta[0], torch.arange(7 * 31) % 7)
    bad = A_in_sta.clone()
    bad[0, 5] = (bad[0, 5] + 7) % (7 * 31)
    with pytest.raises(ValueError):
        graph.base_tables_from_product(bad, A_in_src, 7, 31)


def test_base_tables_refuse_what_they_cannot_cut():
    """graph.base_tables_from_product on host lists: a base graph without any edge (the randomized sweep drew 3 stations whose edges were
    all dropped: an empty `max()` crashed here until round 5), lists that are not a multiple of the node counts, non-uniform degrees and
    the deferred form on host tensors all raise ValueError -- the signal `set_adjacencies` takes its general path on."""
    geom = synthetic.Geometry(6, 40, L=60e3, n_query=4, seed=3)
    A1, A2, _, _ = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, 6, 40)
    sta, src = graph.base_tables_from_product(A1, A2, 6, 40)
    assert sta.shape[0] == 6 and src.shape[0] == 40
    empty = torch.zeros((2, 0), dtype=torch.long)
    for args in ((empty, A2), (A1, empty), (A1[:, :-1], A2)):
        with pytest.raises(ValueError):
            graph.base_tables_from_product(*args, 6, 40)
    ragged = np.ascontiguousarray(geom.A_sta_sta[:, 1:])                      # one station with one neighbour fewer
    B1, _, _, _ = graph.cartesian_product_edges(ragged, geom.A_src_src, 6, 40)
    with pytest.raises(ValueError):
        graph.base_tables_from_product(B1, A2, 6, 40)
    with pytest.raises(ValueError):
        graph.base_tables_from_product(A1, A2, 6, 40, defer=True)             # (device lists only)
    t0, dt, dev_t, n = engine.time_partition(np.arange(-3.0, 9.0, 0.6))
    assert dev_t is None and n == 20 and t0 == -3.0 and abs(dt - 0.6) < 1e-12
    assert engine.time_partition((1.0, 2.0, None, 7)) == (1.0, 2.0, None, 7)
    t0, dt, dev_t, n = engine.time_partition(torch.arange(5, dtype=torch.float32) * 0.25)
    assert (t0, dt, dev_t, n) == (0.0, 0.25, None, 5)


def test_knn_graph_matches_bruteforce():
    rng = np.random.default_rng(0)
    x = rng.random((60, 3))
    A = graph.knn_graph(x, 5)
    d = ((x[:, None] - x[None]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    want = np.argsort(d, axis=1)[:, :5]
    assert np.array_equal(A[0].reshape(60, 5), want)
    assert np.array_equal(A[1], np.repeat(np.arange(60), 5))


def test_csr_and_morton():
    ei = torch.tensor([[3, 1, 2, 0, 1], [0, 0, 2, 2, 1]])
    rowptr, col = engine.csr_from_edges(ei, 4)
    assert rowptr.tolist() == [0, 2, 3, 5, 5] and col.tolist() == [3, 1, 1, 2, 0]
    order = engine.morton_order(np.random.default_rng(1).random((100, 3)))
    assert sorted(order.tolist()) == list(range(100))


def test_synthetic_window_semantics():
    geom = synthetic.Geometry(12, 50, L=80e3, n_query=10, seed=5)
    win = synthetic.make_window(geom, 300, seed=6)
    S, G = 12, 50
    assert win["Slice"].shape == (S * G, 4) and win["Mask"].shape == (S * G, 4)
    assert win["n_picks"] == 300
    assert np.array_equal(win["Mask"], (win["Slice"] > 0.01).astype(np.float32))
    # brute-force check of feature 0 (nearest pick of any phase to the theoretical P arrival)
    P = win["P"]
    tt = geom.travel_times()
    for g, s in [(0, 0), (17, 5), (49, 11)]:
        t = P[P[:, 1] == s, 0]
        want = np.exp(-0.5 * np.min(np.abs(t - tt[g, s, 0])) ** 2 / 9.0) if t.size else 0.0
        assert abs(win["Slice"][g * S + s, 0] - want) < 1e-6
    # phase-restricted features never exceed the any-phase ones
    assert (win["Slice"][:, 2] <= win["Slice"][:, 0] + 1e-7).all()
    assert (win["Slice"][:, 3] <= win["Slice"][:, 1] + 1e-7).all()


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    c = Case("tiny_6x40")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu")
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr, edge_index=A_src_in_prod)
    with pytest.raises(_lib.GenieHipError):
        net.set_adjacencies(A_in_sta, A_in_src, ea, ea, A_src_in_sta, c.A_src_src, None, None, None, None,
                            c.locs.float(), c.x_grid.float())
    with pytest.raises(RuntimeError):
        net.forward_fixed_source(c.Slice, c.Mask, None, None, None, c.locs.float(), c.x_grid.float(),
                                 c.x_query.float(), c.t_query.float())


def test_readout_heads_match_oracle_cpu():
    """The PyTorch restatements of the read-out heads (tests/restatements.py: SpatialDirect / TemporalAttention / SpatialAttention on
    the product module's parameters) against the golden vectors, fed with the reference's own sa3."""
    from tests import restatements as R
    for name in GOLDEN_CASES:
        c = Case(name)
        net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device="cpu")
        net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
        R.attach(net)
        with torch.no_grad():
            sa3 = c.ref("sa3")
            y = net.TemporalAttention(net.SpatialDirect(sa3), c.t_query.float())
            xq = net.SpatialAttention(sa3, c.x_query.float(), c.x_grid.float())
            x = net.TemporalAttention(xq, c.t_query.float())
        rel = lambda k: max(1.0, float(c.ref(k).abs().max()))      # relative to max|ref| on the fixture with O(1) outputs
        assert float((y - c.ref("y")).abs().max()) < 1e-6 * rel("y")
        if "xq" in c.z.files:
            assert float((xq - c.ref("xq")).abs().max()) < 2e-6 * rel("xq")
        assert float((x - c.ref("x")).abs().max()) < 1e-6 * rel("x")


def test_space_filling_curve_orders():
    """engine.morton_order / hilbert_order / sfc_order: permutations; one common scale for all axes (a nearly flat point set is
    ordered by its two long axes, not by the noise of the short one); on a full lattice consecutive Hilbert points are lattice
    neighbours."""
    import numpy as np
    from genie_amd import engine
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(0, 300e3, 500), rng.uniform(0, 300e3, 500), rng.uniform(-1e3, 1e3, 500)], axis=1)
    for f in (engine.morton_order, engine.hilbert_order, engine.sfc_order):
        p = np.asarray(f(pts))
        assert p.dtype == np.int32 and sorted(p.tolist()) == list(range(500))
    flat = pts.copy()
    flat[:, 2] = 0.0
    for f in (engine.morton_order, engine.hilbert_order):
        a, b = np.asarray(f(pts)), np.asarray(f(flat))
        assert (a == b).mean() > 0.9          # +-1 km of elevation over 300 km moves (almost) nothing
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), axis=-1).reshape(-1, 3).astype(float)
    h = g[np.asarray(engine.hilbert_order(g, bits=3))]
    assert np.all(np.abs(np.diff(h, axis=0)).sum(1) == 1)
    with pytest.raises(ValueError):
        engine.morton_order(np.zeros((5, 2)))


def test_query_knn_cache_cannot_return_a_stale_table(monkeypatch):
    """SpatialAttention.query_edges caches the kNN table of the last query set by (address, version, shape) and holds the
    tensors, so a query set freed and re-allocated at the same address, or edited in place, is never served the old table."""
    import gc
    from genie_amd import module
    from tests import restatements as R
    monkeypatch.setattr(module, "knn_query_edges", R.knn_query_edges)       # (the product's search is a HIP kernel: GPU tensors only)
    sa = module.SpatialAttention(30, 30, 3, 15)
    g = torch.Generator().manual_seed(3)
    xc = torch.rand(200, 3, generator=g) * 1e5
    seen = []
    for trial in range(4):                       # same shape every time: the allocator tends to hand back the same block
        xq = torch.rand(50, 3, generator=g) * 1e5
        tab = sa.query_table(xq, xc, 10).clone()
        want = module.knn_query_edges(xc, xq, 10)[0].view(50, 10).int()
        assert torch.equal(tab, want), trial
        seen.append(xq.data_ptr())
        del xq
        gc.collect()
    xq = torch.rand(50, 3, generator=g) * 1e5
    t1 = sa.query_table(xq, xc, 10).clone()
    assert sa.query_table(xq, xc, 10).data_ptr() == sa._edge_cache["table"].data_ptr()      # hit
    xq.mul_(-1.0).add_(1e5)                       # in-place edit: same address, new version
    t2 = sa.query_table(xq, xc, 10)
    assert torch.equal(t2, module.knn_query_edges(xc, xq, 10)[0].view(50, 10).int()) and not torch.equal(t1, t2)
    sa.invalidate_query_cache()
    assert sa._edge_cache == {}


def test_subset_oracle_matches_the_full_structured_oracle():
    """tests.util.oracle_bipartite_for_nodes (the oracle on the two-hop neighbourhood of a few source nodes, used to check
    config-4-sized runs) against the full structured oracle on a problem small enough to evaluate whole."""
    from genie_amd import graph, synthetic
    from oracle import genie_oracle as O
    from tests.util import Case, max_abs, oracle_bipartite_for_nodes
    S, G = 23, 300
    geom = synthetic.Geometry(S, G, L=150e3, n_query=5, seed=7)
    P = synthetic.make_picks(geom, 500, seed=8)
    Slice, Mask = synthetic.make_slice_mask(geom, P, 0.0)
    w = Case("odd_33x257").weights
    sta, src = graph.neighbour_table(geom.A_sta_sta, S), graph.neighbour_table(geom.A_src_src, G)
    xl = O.data_aggregation_structured(w, torch.from_numpy(Slice), torch.from_numpy(Mask), sta, src, S, G)
    bip = O.bipartite_read_in_structured(w, xl, torch.from_numpy(geom.edge_attr()), torch.from_numpy(Mask), S, G)
    sample = np.array([5, 17, 299, 120])
    b, x = oracle_bipartite_for_nodes(w, geom, P, sample)
    assert max_abs(b, bip[sample]) <= 1e-6 * max(1.0, float(bip.abs().max()))
    assert max_abs(x.view(4, S, -1), xl.view(G, S, -1)[sample]) <= 1e-6


def test_subgraph_pairs_match_the_reference_builder_on_cpu():
    """engine.subgraph_pairs_device (torch ops, any device): the product nodes of `use_subgraph` for the geometry of
    tests/golden/subgraph_builder_14x50.npz equal the reference builder's A_src_in_sta (process_utils.py:775-794)."""
    import numpy as np
    import torch
    from genie_amd import engine, synthetic
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subgraph_builder_14x50.npz"))
    geom = synthetic.Geometry(14, 50, L=80e3, n_query=21, seed=71)
    pairs = engine.subgraph_pairs_device(torch.from_numpy(geom.locs), torch.from_numpy(geom.x_grid), max_deg_offset=0.15,
                                         k_nearest_pairs=6, scale_deg=110e3)
    assert np.array_equal(pairs.numpy(), z["A_src_in_sta"])


def test_sharded_drop_in_keeps_the_reference_signatures_and_state_dict():
    """The multi-GPU form is the SAME class with two more keyword arguments (`process_group`, `shard`): positional signatures of the
    reference's methods (module.py:908, :941, :963, :999) and the 158 state_dict keys are unchanged; what is not sharded says so before it
    touches a GPU; a sharded model without adjacencies refuses like an unsharded one."""
    import inspect
    ref_args = {
        "set_adjacencies": ["A_in_sta", "A_in_src", "A_src_in_edges", "A_Lg_in_src", "A_src_in_sta", "A_src", "A_edges_p", "A_edges_s",
                            "dt_partition", "tlatent", "pos_loc", "pos_src"],
        "forward_fixed_source": ["Slice", "Mask", "tpick", "ipick", "phase_label", "locs_use_cart", "x_temp_cuda_cart", "x_query_cart", "t_query"],
        "forward_fixed": ["Slice", "Mask", "tpick", "ipick", "phase_label", "locs_use_cart", "x_temp_cuda_cart", "x_query_cart",
                          "x_query_src_cart", "t_query", "tq_sample", "trv_out_q"],
    }
    cls = module.GCN_Detection_Network_extended
    for name, args in ref_args.items():
        got = [p for p in inspect.signature(getattr(cls, name)).parameters if p != "self" and not p.startswith("_")]
        assert got == args, (name, got)
    assert len([p for p in inspect.signature(cls.forward).parameters if p != "self"]) == 22
    plain = cls(lambda x: x, lambda x: x, device="cpu")
    net = cls(lambda x: x, lambda x: x, device="cpu", shard=(1, 4))
    assert net.is_sharded and not plain.is_sharded and net.shard_plan is None
    assert list(net.state_dict().keys()) == list(plain.state_dict().keys()) and len(net.state_dict()) == 158
    z = torch.zeros(4, 4)
    with pytest.raises(RuntimeError, match="set_adjacencies"):
        net.forward_fixed_source(z, z, None, None, None, None, z, z, z)
    net._hip = object()                      # (pretend the adjacencies are set: the refusals below come before any use of the context)
    for call in (lambda: net.forward_fixed(z, z, None, None, None, None, z, z, z, z, None, None), lambda: net.push_window(z, z),
                 lambda: net(*([z] * 22))):
        with pytest.raises(NotImplementedError, match="source-node-sharded"):
            call()
    net._hip = None
    with pytest.raises(ValueError):
        cls(lambda x: x, lambda x: x, device="cpu", shard=(4, 4))
