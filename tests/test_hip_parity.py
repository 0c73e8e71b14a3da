"""GPU: parity of the HIP path (through the C ABI) against the oracle, the golden vectors and
size-independent properties. fp32 tolerances are written next to every assert:
  * outputs (y, x): 1e-5 absolute (BASELINE.json north_star);
  * intermediates: 1e-5 x max(1, max|ref|) (the station sum already drifts 1e-5 between two
    correct fp32 evaluations at S=200)."""
import numpy as np
import pytest
import torch

from genie_amd import engine, graph, module, synthetic
from tests.util import ABSPOS_CASES, EDGES_CASES, GOLDEN_CASES, SUBGRAPH_CASES, Case, max_abs

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rel_tol(ref, scale=1e-5):
    return scale * max(1.0, float(ref.abs().max()))


def make_engine(c, order="morton"):
    sta_nbr, src_nbr = c.tables()
    go = engine.morton_order(c.x_grid.numpy()) if order == "morton" else None
    hp = engine.HipPath(c.S, c.G, engine.csr_from_table(sta_nbr), engine.csr_from_table(src_nbr), grid_order=go, device=DEV)
    hp.set_weights({k: v.to(DEV) for k, v in c.weights.items()})
    return hp


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_stages_match_golden_intermediates(name):
    """Every stage of the HIP path against the reference's own intermediates (golden fixtures)."""
    c = Case(name)
    hp = make_engine(c)
    Slice, Mask, h0, h1 = hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV), debug=True)
    x_latent, bip = hp.da_stage2_bipartite(Mask, c.edge_attr.to(DEV), want_x_latent=True)
    got = {"h0": h0, "h1": h1, "x_latent": x_latent, "bip": bip}
    # stage 1 stores u / v projected through the neighbour-mean columns of l2_t1_2 / l2_t2_2, and the node-local
    # part of layer 2 (c) instead of h1
    w = c.weights
    W1, W2 = w["DataAggregation.l2_t1_2.weight"], w["DataAggregation.l2_t2_2.weight"]
    for key, which, W in (("u", 1, W1), ("v", 2, W2)):
        proj = hp.export(which).cpu()
        if key in c.z.files:
            ref = c.ref(key) @ W[:, 60:90].T
            assert max_abs(c.strided(proj), ref) <= rel_tol(ref), ("projected " + key, max_abs(c.strided(proj), ref))
    if "h1" in c.z.files and c.row_stride == 1:
        cc = hp.export(0).cpu()
        M = c.Mask
        ref = torch.cat((c.ref("h1") @ W1[:, 0:60].T + M @ W1[:, 90:94].T + w["DataAggregation.l2_t1_2.bias"],
                         c.ref("h1") @ W2[:, 0:60].T + M @ W2[:, 90:94].T + w["DataAggregation.l2_t2_2.bias"]), dim=1)
        assert max_abs(cc, ref) <= rel_tol(ref), ("c", max_abs(cc, ref))
    pos = c.x_grid.float().to(DEV)
    got["sa1"] = hp.spatial_agg(1, bip, pos)
    got["sa2"] = hp.spatial_agg(2, got["sa1"], pos)
    got["sa3"] = hp.spatial_agg(3, got["sa2"], pos)
    torch.cuda.synchronize()
    oracle = c.oracle_forward(torch.float32, structured=True)
    for k, v in got.items():
        v = v.cpu()
        o = oracle[k]
        assert max_abs(v, o) <= rel_tol(o), ("vs oracle", k, max_abs(v, o))
        if k in c.z.files:
            ref = c.ref(k)
            assert max_abs(c.strided(v), ref) <= rel_tol(ref), ("vs golden", k, max_abs(c.strided(v), ref))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_forward_fixed_source_drop_in(name):
    """The reference call sequence: set_adjacencies(product edge lists) then forward_fixed_source."""
    c = Case(name)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea, A_src_in_sta.to(DEV), c.A_src_src.to(DEV),
                        None, None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    with torch.no_grad():
        y, x = net.forward_fixed_source(c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV),
                                        c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
    assert y.shape == tuple(c.ref("y").shape) and x.shape == tuple(c.ref("x").shape)
    # fp32 max-abs tolerance of BASELINE.json: 1e-5 ABSOLUTE against the reference's fp64 run (the truth), also on the fixture
    # whose outputs are O(1) (`o1_20x500`, max|y| 7.2: 1e-5 is 1.4e-6 relative there, 1.3 x the reference's own fp32 error);
    # against the reference's fp32 run the same 1e-5 plus that run's own distance from the truth
    for got, k in ((y, "y"), (x, "x")):
        e64, e32 = max_abs(got.cpu(), c.ref(k + "64")), max_abs(got.cpu(), c.ref(k))
        print("%s %s: max|ref| %.3g  |hip - ref64| %.3g  |hip - ref32| %.3g" % (name, k, float(c.ref(k).abs().max()), e64, e32))
        assert e64 <= 1e-5, (k, e64)
        assert e32 <= 1e-5 + max_abs(c.ref(k), c.ref(k + "64")), (k, e32)
        # round 4: the G-sized tail runs fp64 MFMA chains, so the grid read-out sits well inside the bound where the reference's
        # own fp32 run uses three quarters of it (o1_20x500: 7.6e-6); observed 3.6e-6 (fp32 chains: 8.7e-6 .. 9.3e-6)
        if k == "y":
            assert e64 <= 5e-6, (k, e64)


@pytest.mark.parametrize("scale", [2.0 ** -12, 512.0])
def test_stage1_piece_range_small_and_large_inputs(scale):
    """The f16x2 stage 1 represents every activation by two fp16 pieces: normal fp16 numbers for |x| in [2^-2, 65504], an absolute
    floor of 2^-25 below. Inputs scaled by 2^-12 (hidden states dominated by the biases, input terms far inside fp16's subnormal
    range) and by 512 (hidden states up to ~1e3): stage 1's intermediates and x_latent against the oracle on the same inputs, to
    the usual 1e-5 of their scale. (Beyond 65504 an activation overflows its first piece and the outputs turn non-finite:
    stage_precision="f32" selects the fp32-MFMA kernels for such models.)"""
    from oracle import genie_oracle as O
    c = Case("odd_33x257")
    Slice = (c.Slice * scale).contiguous()
    w = {k: v.float() for k, v in c.weights.items()}
    A_in_sta, A_in_src, _, _ = c.product_edges()
    ref = O.data_aggregation(w, Slice, c.Mask, A_in_sta, A_in_src, full=True)
    hp = make_engine(c)
    _, _, h0, h1 = hp.da_stage1(Slice.to(DEV), c.Mask.to(DEV), debug=True)
    x_latent, _ = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
    for got, k in ((h0, "h0"), (h1, "h1"), (x_latent, "x_latent")):
        tol = 1e-5 * max(1.0, float(ref[k].abs().max()))
        err = max_abs(got.cpu(), ref[k])
        print("scale %g %s: max|ref| %.3g err %.3g" % (scale, k, float(ref[k].abs().max()), err))
        assert torch.isfinite(got).all() and err <= tol, (k, err, tol)


def test_inputs_beyond_the_verified_range_fail_loudly_and_fall_back_to_fp32():
    """the static fp16 range guard bounds the hidden states for inputs in [-1, 1]; the bound is linear
    in the input magnitude, so the f16x2 kernels are valid up to |Slice| <= 60000 / bound (genie_input_range). Inputs beyond that --
    not producible by the reference's embedding, but fine for its fp32 arithmetic -- used to overflow to inf silently. Now the split
    pass flags them through host-mapped memory: the NEXT entry into the library raises, the context has switched to the fp32
    kernels, and the repeated call is finite and equal to the oracle. Inputs inside the limit (scale 512: the test above) raise nothing."""
    from genie_amd import _lib
    from oracle import genie_oracle as O
    c = Case("odd_33x257")
    hp = make_engine(c)
    Mask = c.Mask.to(DEV)
    hp.da_stage1(c.Slice.to(DEV), Mask)                      # commits the weights: the limit is known afterwards
    lim = hp.input_limit()
    assert 100.0 < lim <= 60000.0
    hp.check_input_range()                                   # nothing flagged so far
    big = (c.Slice * (4.0 * lim)).contiguous()
    hp.da_stage1(big.to(DEV), Mask)                          # f16x2 kernels on out-of-range inputs: flagged by the split pass
    torch.cuda.synchronize()
    with pytest.raises(_lib.GenieHipError, match="magnitude"):
        hp.da_stage1(big.to(DEV), Mask)
    assert hp.stage_precision()["mode"] == "f32"
    hp.check_input_range()                                   # the flag was cleared with the error
    w = {k: v.float() for k, v in c.weights.items()}
    A_in_sta, A_in_src, _, _ = c.product_edges()
    ref = O.data_aggregation(w, big, c.Mask, A_in_sta, A_in_src, full=True)
    hp.da_stage1(big.to(DEV), Mask)                          # the repeated call: fp32 kernels
    x_latent, _ = hp.da_stage2_bipartite(Mask, c.edge_attr.to(DEV), want_x_latent=True)
    assert torch.isfinite(x_latent).all()
    assert max_abs(x_latent.cpu(), ref["x_latent"]) <= 1e-5 * max(1.0, float(ref["x_latent"].abs().max()))
    # the drop-in class: the error surfaces at the next forward, the one after that succeeds
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    A1, A2, A3, A4 = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A3.to(DEV))
    net.set_adjacencies(A1.to(DEV), A2.to(DEV), ea, ea, A4.to(DEV), c.A_src_src.to(DEV), None, None, None, None, c.locs.float().to(DEV),
                        c.x_grid.float().to(DEV))
    args = (None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
    with torch.no_grad():
        net.forward_fixed_source(big.to(DEV), Mask, *args)
        torch.cuda.synchronize()
        with pytest.raises(_lib.GenieHipError):
            net.forward_fixed_source(big.to(DEV), Mask, *args)
        y, x = net.forward_fixed_source(big.to(DEV), Mask, *args)
    assert torch.isfinite(y).all() and torch.isfinite(x).all()


@pytest.mark.parametrize("gains", [(1.0, 1.0, 1.0, 1.0), (64.0, 64.0, 64.0, 64.0), (4096.0, 4096.0, 1.0, 1.0)])
def test_fp16_range_guard_selects_the_fp32_kernels_without_any_switch(gains):
    """The f16x2 kernels split every hidden state of DataAggregation into fp16 pieces: beyond 65504 they would return inf / NaN
    where the reference's fp32 arithmetic is fine. The library bounds those hidden states from the committed weights and runs the
    fp32-MFMA kernels when the bound leaves the fp16 range -- automatically (no environment variable, no argument). Weights of
    init_trns / layer 1 / l2_t?_1 / l2_t?_2 scaled by `gains`: hidden states reach ~3e5 (x64 everywhere) and ~4e8 (x4096 on the
    first two layers). Every output finite and within 1e-5 of its scale from the fp64 oracle; the unscaled model keeps f16x2."""
    from oracle import genie_oracle as O
    c = Case("cfg1_20x500")             # uniform 8 / 15-degree graphs: the shape admits the f16x2 kernels
    layer = {"init_trns": 0, "l1_t1_2": 1, "l1_t2_2": 1, "l2_t1_1": 2, "l2_t2_1": 2, "l2_t1_2": 3, "l2_t2_2": 3}
    w = {}
    for k, v in c.weights.items():
        parts = k.split(".")
        g = gains[layer[parts[1]]] if (parts[0] == "DataAggregation" and parts[1] in layer) else 1.0
        w[k] = (v * g).contiguous()
    sta_nbr, src_nbr = c.tables()
    hp = engine.HipPath(c.S, c.G, engine.csr_from_table(sta_nbr), engine.csr_from_table(src_nbr),
                        grid_order=engine.morton_order(c.x_grid.numpy()), device=DEV, sta_order=engine.sfc_order(c.locs.numpy()))
    hp.set_weights({k: v.to(DEV) for k, v in w.items()})
    info = hp.stage_precision()
    print("gains %s: %s" % (gains, info))
    assert info["mode"] == "auto"
    assert info["f16x2_active"] == (gains[0] == 1.0), info
    _, _, h0, h1 = hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV), debug=True)
    x_latent, bip = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
    w64 = {k: v.double() for k, v in w.items()}
    A_in_sta, A_in_src, A_src_in_prod, _ = c.product_edges()
    ref = O.data_aggregation(w64, c.Slice.double(), c.Mask.double(), A_in_sta, A_in_src, full=True)
    ref["bip"] = O.bipartite_read_in(w64, ref["x_latent"], c.edge_attr.double(), A_src_in_prod, c.Mask.double())
    if gains[0] > 1.0:
        assert max(float(ref[k].abs().max()) for k in ("h0", "h1", "u", "v", "x_latent")) > 65504.0     # the case the guard exists for
    for got, k in ((h0, "h0"), (h1, "h1"), (x_latent, "x_latent"), (bip, "bip")):
        scale = float(ref[k].abs().max())
        err = max_abs(got.cpu(), ref[k])
        print("  %s: max|ref| %.4g err %.3g (%.2g of scale)" % (k, scale, err, err / scale))
        assert torch.isfinite(got).all(), k
        assert err <= 1e-5 * max(1.0, scale), (k, err, scale)
    if gains[0] > 1.0:
        # forced f16x2 on the same weights (A/B mode): the overflow the guard prevents is real
        hp.set_stage_precision("f16x2")
        hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV))
        _, bip_f16 = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV))
        # (inf pieces turn into NaN in the matrix pipe and the NaN-dropping min / max of the PReLUs make them finite again: the
        # forced mode is silently WRONG beyond the fp16 range, which is why the choice is not left to the caller)
        bad = bip_f16.cpu().double()
        assert (not torch.isfinite(bad).all()) or max_abs(bad, ref["bip"]) > 1e-3 * float(ref["bip"].abs().max())


def test_fp16_range_guard_sees_the_static_terms_of_long_tables():
    """The range guard also bounds the static per-station / per-source-node terms of the two other model definitions. Their tables
    are as long as the graph ([n_grid, 48] edge-feature terms, [n_grid, 4] positions): above 4096 entries they are reduced by many
    workgroups first (k_tab_absmax) and the guard reads the partial maxima. 40 stations x 300 source nodes: the unscaled model keeps
    f16x2; with the edge-feature columns of l1_t2_2 (source side, the long table) scaled until the term alone leaves the fp16 range the
    library switches to the fp32 kernels by itself and the bound it reports grows accordingly."""
    S, G = 40, 300
    geom = synthetic.Geometry(S, G, L=200e3, n_query=10, seed=11)
    w0 = Case("edges_12x60").weights
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)

    def info_for(gain):
        net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition=True).eval()
        w = {k: v.clone() for k, v in w0.items()}
        w["DataAggregation.l1_t2_2.weight"][:, 60:64] *= gain
        net.load_state_dict(w, strict=True)
        net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), t(geom.locs),
                                 t(geom.x_grid))
        win = synthetic.make_window(geom, 500, seed=12)
        with torch.no_grad():
            y, x = net.forward_fixed_source(t(win["Slice"]), t(win["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query),
                                            t(geom.t_query))
        assert torch.isfinite(y).all() and torch.isfinite(x).all()
        return net._hip.stage_precision()
    a, b = info_for(1.0), info_for(3.0e6)
    print(a, b)
    assert G * 48 > 4096                         # the long-table path
    assert a["f16x2_active"] and not b["f16x2_active"]
    assert b["act_bound"] > 60000.0 > a["act_bound"]


@pytest.mark.parametrize("name", EDGES_CASES)
@pytest.mark.parametrize("stage1", ["default", "f32"])
def test_updated_model_definition_forward_fixed_source(name, stage1, monkeypatch):
    """a-9: the `use_updated_model_definition` class (DataAggregationEdges, module.py:102-174, :1163-1185) against fixtures
    generated from the reference imported with that flag. edges_12x60 has uniform 8 / 15 degrees (f16x2 stage 1, or the
    pipelined fp32 kernel with stage_precision="f32"), edges_7x13 has 6 / 12 (generic CSR kernels)."""
    if stage1 == "f32":
        monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    c = Case(name)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition=True)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea, A_src_in_sta.to(DEV), c.A_src_src.to(DEV),
                        None, None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    with torch.no_grad():
        y, x = net.forward_fixed_source(c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV),
                                        c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
        hp = net._hip
        _, _, h0, h1 = hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV), debug=True)
        x_latent, bip = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
    for k, v in (("h0", h0), ("h1", h1), ("x_latent", x_latent), ("bip", bip)):
        ref = c.ref(k)
        assert max_abs(v.cpu(), ref) <= rel_tol(ref), (k, max_abs(v.cpu(), ref))
    assert max_abs(y.cpu(), c.ref("y")) <= 1e-5 and max_abs(x.cpu(), c.ref("x")) <= 1e-5
    assert max_abs(y.cpu(), c.ref("y64")) <= 1e-5 and max_abs(x.cpu(), c.ref("x64")) <= 1e-5


@pytest.mark.parametrize("name", ABSPOS_CASES)
def test_use_absolute_pos_forward_fixed_source(name):
    """`use_absolute_pos: True` (config.yaml:92, module.py:1007): every product node's input carries its station and source
    position / (3 scale_rel); the neighbour recompute uses each NEIGHBOUR's positions (genie_set_absolute_pos)."""
    c = Case(name)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_absolute_pos=True)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea, A_src_in_sta.to(DEV), c.A_src_src.to(DEV),
                        None, None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    with torch.no_grad():
        y, x = net.forward_fixed_source(c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV),
                                        c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
        hp = net._hip
        _, _, h0, h1 = hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV), debug=True)
        x_latent, bip = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
    for k, v in (("h0", h0), ("h1", h1), ("x_latent", x_latent), ("bip", bip)):
        ref = c.ref(k)
        assert max_abs(v.cpu(), ref) <= rel_tol(ref), (k, max_abs(v.cpu(), ref))
    assert max_abs(y.cpu(), c.ref("y")) <= 1e-5 and max_abs(x.cpu(), c.ref("x")) <= 1e-5
    assert max_abs(y.cpu(), c.ref("y64")) <= 1e-5 and max_abs(x.cpu(), c.ref("x64")) <= 1e-5


@pytest.mark.parametrize("name", ["subgraph_edges_14x50", "subgraph_abspos_14x50"])
@pytest.mark.parametrize("stage1", ["default", "f32"])
def test_updated_model_definition_on_an_irregular_product_graph(name, stage1, monkeypatch):
    """`use_updated_model_definition: True` with `use_subgraph: True`: the mean edge feature of a product node runs over its PRESENT
    neighbours, so the static terms are per product node (genie_set_edge_features with positions per product node: `k_edge_feat` on
    the product-level CSRs, `[n_prod, 48]` term tables indexed by product node in k_stage1_h2<EDGES, .., PCSR> / k_stage1_pcsr).
    Fixture: the reference imported with the flag, run on the irregular graph of `subgraph_14x50`."""
    if stage1 == "f32":
        monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    # (subgraph_abspos_14x50: `use_absolute_pos: True` on the same graph -- position tables per product node, the generic fp32-MFMA stage 1)
    c = Case(name)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition=c.edges_variant,
                                                use_absolute_pos=c.abspos_variant)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea, A_src_in_sta.to(DEV), c.A_src_src.to(DEV),
                        None, None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    assert net._hip._n_prod is not None
    with torch.no_grad():
        y, x = net.forward_fixed_source(c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV),
                                        c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
        hp = net._hip
        _, _, h0, h1 = hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV), debug=True)
        x_latent, bip = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
    for k, v in (("h0", h0), ("h1", h1), ("x_latent", x_latent), ("bip", bip)):
        ref = c.ref(k)
        assert max_abs(v.cpu(), ref) <= rel_tol(ref), (k, max_abs(v.cpu(), ref))
    assert max_abs(y.cpu(), c.ref("y")) <= 1e-5 and max_abs(x.cpu(), c.ref("x")) <= 1e-5
    assert max_abs(y.cpu(), c.ref("y64")) <= 1e-5 and max_abs(x.cpu(), c.ref("x64")) <= 1e-5


def test_both_model_options_together_forward_fixed_source():
    """`use_updated_model_definition: True` with `use_absolute_pos: True` (the reference's classes take both, module.py:103-109,
    :1056): the edge-feature terms and the position columns are both static additive terms of the stage-1 pre-activations
    (genie_set_edge_features + genie_set_absolute_pos on one context; the generic fp32-MFMA stage 1 serves the combination). Fixture
    from the reference imported with both flags set."""
    c = Case("edges_abspos_12x60")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_absolute_pos=True, use_updated_model_definition=True)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea, A_src_in_sta.to(DEV), c.A_src_src.to(DEV),
                        None, None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    with torch.no_grad():
        y, x = net.forward_fixed_source(c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV),
                                        c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
        hp = net._hip
        _, _, h0, h1 = hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV), debug=True)
        x_latent, bip = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
    for k, v in (("h0", h0), ("h1", h1), ("x_latent", x_latent), ("bip", bip)):
        ref = c.ref(k)
        assert max_abs(v.cpu(), ref) <= rel_tol(ref), (k, max_abs(v.cpu(), ref))
    assert max_abs(y.cpu(), c.ref("y")) <= 1e-5 and max_abs(x.cpu(), c.ref("x")) <= 1e-5
    assert max_abs(y.cpu(), c.ref("y64")) <= 1e-5 and max_abs(x.cpu(), c.ref("x64")) <= 1e-5


@pytest.mark.parametrize("name", SUBGRAPH_CASES)
@pytest.mark.parametrize("stage1", ["default", "f32"])
def test_use_subgraph_irregular_product_graph(name, stage1, monkeypatch):
    """f-4: `use_subgraph: True` (config.yaml:86, process_utils.py:744-849). set_adjacencies receives irregular product edge
    lists; the module builds product-level CSRs (genie_ctx_create_subgraph); stage 1 runs k_stage1_h2<.., PCSR> (neighbours as
    product-node ids, missing ones with weight 0; stage_precision="f32": the generic fp32-MFMA k_stage1_pcsr), then k_stage2_pcsr /
    k_bip_out_seg. Checked against the reference's own run on that graph and against the oracle."""
    if stage1 == "f32":
        monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    c = Case(name)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea, A_src_in_sta.to(DEV), c.A_src_src.to(DEV),
                        None, None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    hp = net._hip
    assert hp.n_prod == A_src_in_sta.shape[1] < c.S * c.G
    with torch.no_grad():
        y, x = net.forward_fixed_source(c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV),
                                        c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
        _, _, h0, h1 = hp.da_stage1(c.Slice.to(DEV), c.Mask.to(DEV), debug=True)
        x_latent, bip = hp.da_stage2_bipartite(c.Mask.to(DEV), c.edge_attr.to(DEV), want_x_latent=True)
    oracle = c.oracle_forward(torch.float32)
    for k, v in (("h0", h0), ("h1", h1), ("x_latent", x_latent), ("bip", bip)):
        ref = c.ref(k)
        assert max_abs(v.cpu(), ref) <= rel_tol(ref), (k, max_abs(v.cpu(), ref))
        assert max_abs(v.cpu(), oracle[k]) <= rel_tol(ref), ("oracle", k)
    assert max_abs(y.cpu(), c.ref("y")) <= 1e-5 and max_abs(x.cpu(), c.ref("x")) <= 1e-5
    assert max_abs(y.cpu(), c.ref("y64")) <= 1e-5 and max_abs(x.cpu(), c.ref("x64")) <= 1e-5


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_readout_kernels_match_golden(name):
    """HIP read-out heads fed with the reference's own sa3: isolates them from the message-passing kernels."""
    c = Case(name)
    hp = make_engine(c)
    sa3 = c.ref("sa3").to(DEV)
    tq = c.t_query.float().to(DEV)
    y = hp.readout_grid(sa3, tq)
    from genie_amd.module import knn_query_edges
    xg, xq = c.x_grid.float().to(DEV), c.x_query.float().to(DEV)
    table = knn_query_edges(xg, xq, 10)[0].view(xq.shape[0], -1).to(torch.int32).contiguous()
    x = hp.readout_query(sa3, xg, xq, table, tq)
    assert y.shape == tuple(c.ref("y").shape) and x.shape == tuple(c.ref("x").shape)
    for got, k in ((y, "y"), (x, "x")):          # 1e-6 absolute; relative to max|ref| on the O(1) fixture
        assert max_abs(got.cpu(), c.ref(k)) <= 1e-6 * max(1.0, float(c.ref(k).abs().max())), k


def _random_case(S, G, seed, n_picks):
    geom = synthetic.Geometry(S, G, L=200e3, n_query=50, seed=seed)
    win = synthetic.make_window(geom, n_picks, seed=seed + 1)
    return geom, win


@pytest.mark.parametrize("S,G", [(1 + 2, 9), (16, 64), (17, 33), (50, 700), (200, 300), (2000, 40)])
def test_random_shapes_vs_structured_oracle(S, G):
    """Ragged tile edges (S not a multiple of 16), S < 16, S = 16 exactly, the config-2 station count (200) and the
    config-4 station count (2000: 125 tiles per source node, station sum over 2000 terms)."""
    from oracle import genie_oracle as O
    geom, win = _random_case(S, G, seed=100 + S, n_picks=10 * S)
    c = Case("tiny_6x40")  # weights only
    w = c.weights
    sta_nbr = graph.neighbour_table(geom.A_sta_sta, S)
    src_nbr = graph.neighbour_table(geom.A_src_src, G)
    hp = engine.HipPath(S, G, engine.csr_from_table(sta_nbr), engine.csr_from_table(src_nbr),
                        grid_order=engine.morton_order(geom.x_grid), device=DEV)
    hp.set_weights({k: v.to(DEV) for k, v in w.items()})
    Slice, Mask = torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"])
    ea = torch.from_numpy(geom.edge_attr())
    pos = torch.from_numpy(geom.x_grid).float()
    out, x_latent, bip = hp.path_fwd(Slice.to(DEV), Mask.to(DEV), ea.to(DEV), pos.to(DEV), want_x_latent=True, want_bip=True)
    da = O.data_aggregation_structured(w, Slice, Mask, sta_nbr, src_nbr, S, G, full=True)
    o_bip = O.bipartite_read_in_structured(w, da["x_latent"], ea, Mask, S, G)
    A_src = torch.from_numpy(geom.A_src_src)
    o = o_bip
    for l in (1, 2, 3):
        o = O.spatial_aggregation(w, o, A_src, pos, "SpatialAggregation%d" % l)
    assert max_abs(x_latent.cpu(), da["x_latent"]) <= rel_tol(da["x_latent"])
    assert max_abs(bip.cpu(), o_bip) <= rel_tol(o_bip)
    assert max_abs(out.cpu(), o) <= rel_tol(o)


def test_non_uniform_degree_and_empty_neighbourhoods():
    """CSR graphs with ragged degrees, including nodes with NO in-edges (mean of an empty set = 0)."""
    from oracle import genie_oracle as O
    S, G = 21, 40
    rng = np.random.default_rng(7)
    geom = synthetic.Geometry(S, G, L=100e3, n_query=5, seed=8)
    keep_sta = rng.random(geom.A_sta_sta.shape[1]) < 0.6
    keep_sta[geom.A_sta_sta[1] == 3] = False                       # station 3 has no neighbours
    keep_src = rng.random(geom.A_src_src.shape[1]) < 0.6
    keep_src[geom.A_src_src[1] == 5] = False                       # source node 5 has no neighbours
    A_sta = torch.from_numpy(geom.A_sta_sta[:, keep_sta])
    A_src = torch.from_numpy(geom.A_src_src[:, keep_src])
    c = Case("odd_33x257")
    w = c.weights
    hp = engine.HipPath(S, G, engine.csr_from_edges(A_sta, S), engine.csr_from_edges(A_src, G), device=DEV)
    hp.set_weights({k: v.to(DEV) for k, v in w.items()})
    P = S * G
    Slice = torch.from_numpy(rng.random((P, 4)).astype(np.float32))
    Mask = torch.from_numpy((rng.random((P, 4)) < 0.5).astype(np.float32))
    ea = torch.from_numpy(geom.edge_attr())
    pos = torch.from_numpy(geom.x_grid).float()
    out, x_latent, bip = hp.path_fwd(Slice.to(DEV), Mask.to(DEV), ea.to(DEV), pos.to(DEV), True, True)
    A_in_sta, A_in_src, A_src_in_prod, _ = graph.cartesian_product_edges(A_sta, A_src, S, G)
    da = O.data_aggregation(w, Slice, Mask, A_in_sta, A_in_src, full=True)
    o_bip = O.bipartite_read_in(w, da["x_latent"], ea, A_src_in_prod, Mask)
    o = o_bip
    for l in (1, 2, 3):
        o = O.spatial_aggregation(w, o, A_src, pos, "SpatialAggregation%d" % l)
    assert max_abs(x_latent.cpu(), da["x_latent"]) <= rel_tol(da["x_latent"])
    assert max_abs(bip.cpu(), o_bip) <= rel_tol(o_bip)
    assert max_abs(out.cpu(), o) <= rel_tol(o)


def test_bitwise_deterministic_and_order_independent():
    """Run-to-run bitwise equality (no atomics), and independence from the processing order of source nodes."""
    c = Case("cfg1_20x500")
    outs = []
    for order in ("morton", "morton", None):
        hp = make_engine(c, order)
        o, xl, bip = hp.path_fwd(c.Slice.to(DEV), c.Mask.to(DEV), c.edge_attr.to(DEV), c.x_grid.float().to(DEV), True, True)
        outs.append((o.cpu(), xl.cpu(), bip.cpu()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", ["cfg1_20x500", "random_50x700"])
def test_kernel_variants_agree(case, monkeypatch):
    """The stage kernels exist in two forms: generic CSR fp32-MFMA (any graph; stage_precision="f32" selects them on the reference's kNN
    graphs too: the A/B reference) and the production pair k_stage1_h2 (two-piece fp16 operands on the matrix pipe) + k_stage2_ord (pipelined,
    row-layout loads): another summation order, fp32 tolerance."""
    if case == "cfg1_20x500":
        c = Case(case)
        S, G, w = c.S, c.G, c.weights
        sta_nbr, src_nbr = c.tables()
        Slice, Mask, ea, pos, xg = c.Slice, c.Mask, c.edge_attr, c.x_grid.float(), c.x_grid.numpy()
    else:
        S, G = 50, 700
        geom, win = _random_case(S, G, seed=77, n_picks=600)
        w = Case("tiny_6x40").weights
        sta_nbr, src_nbr = graph.neighbour_table(geom.A_sta_sta, S), graph.neighbour_table(geom.A_src_src, G)
        Slice, Mask = torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"])
        ea, pos, xg = torch.from_numpy(geom.edge_attr()), torch.from_numpy(geom.x_grid).float(), geom.x_grid
    res = {}
    for name, prec in (("generic", "f32"), ("h2", "auto")):
        monkeypatch.setattr(engine, "STAGE_PRECISION", prec)
        hp = engine.HipPath(S, G, engine.csr_from_table(sta_nbr), engine.csr_from_table(src_nbr),
                            grid_order=engine.morton_order(xg), device=DEV)
        hp.set_weights({k: v.to(DEV) for k, v in w.items()})
        _, _, h0, h1 = hp.da_stage1(Slice.to(DEV), Mask.to(DEV), debug=True)
        xl, bip = hp.da_stage2_bipartite(Mask.to(DEV), ea.to(DEV), want_x_latent=True)
        res[name] = [t.cpu() for t in (h0, h1, xl, bip)]
    for a_, b_ in zip(res["generic"], res["h2"]):
        assert max_abs(a_, b_) <= 0.2 * rel_tol(a_), max_abs(a_, b_)     # 2e-6 x max(1, max|ref|)


def test_rows_beyond_4gib_offsets(monkeypatch):
    """Config-4 scale on one GPU: P x 48 B (split rows) and P x 64 B (wu / wv rows) both exceed 4 GiB, so the kernels that
    address rows with 32-bit byte offsets must switch to their 64-bit forms (k_stage1_h2<.., BIG>, wave-uniform 64-bit row
    bases in k_stage2_ord). Property: same result as the generic CSR kernels (64-bit arithmetic throughout)."""
    S, G = 2000, 46000
    assert S * G * 48 > 2 ** 32
    geom = synthetic.Geometry(S, G, L=2000e3, n_query=5, seed=301)
    sta = engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S)
    src = engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G)
    w = {k: v.to(DEV) for k, v in Case("tiny_6x40").weights.items()}
    g = torch.Generator(device=DEV).manual_seed(5)
    P = S * G
    Slice = torch.rand((P, 4), device=DEV, generator=g)
    Mask = (torch.rand((P, 4), device=DEV, generator=g) < 0.3).float()
    ea = torch.rand((P, 3), device=DEV, generator=g) - 0.5
    res = {}
    for name, prec in (("generic", "f32"), ("default", "auto")):
        monkeypatch.setattr(engine, "STAGE_PRECISION", prec)
        hp = engine.HipPath(S, G, sta, src, grid_order=engine.morton_order(geom.x_grid), device=DEV)
        hp.set_weights(w)
        hp.da_stage1(Slice, Mask)
        _, bip = hp.da_stage2_bipartite(Mask, ea)
        res[name] = bip.cpu()
        del hp
        torch.cuda.empty_cache()
    assert max_abs(res["generic"], res["default"]) <= 0.2 * rel_tol(res["generic"]), max_abs(res["generic"], res["default"])


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,M", [(1, 8, 30), (37, 64, 30), (4099, 94, 15), (70001, 33, 30), (50000, 60, 30), (20011, 33, 75), (3001, 38, 5)])
def test_linear_bwd_wb_matches_fp64(N, K, M):
    """genie_linear_bwd_wb (weight / bias gradients of the per-node Linears, training path) against the fp64 products, at the
    widths the path uses (K = 8, 33, 38, 60, 64, 94; M = 5, 15, 30, 75) and at ragged row counts."""
    hp = engine.HipPath(3, 4, engine.csr_from_edges(torch.zeros((2, 0), dtype=torch.long), 3),
                        engine.csr_from_edges(torch.zeros((2, 0), dtype=torch.long), 4), device=DEV)
    g = torch.Generator(device=DEV).manual_seed(N + K)
    x = torch.randn((N, K), device=DEV, generator=g)
    dy = torch.randn((N, M), device=DEV, generator=g)
    dW, db = hp.linear_bwd_wb(x, dy)
    refW, refb = dy.double().t() @ x.double(), dy.double().sum(0)
    tol = 2e-6 * max(1.0, N ** 0.5) * 4
    assert max_abs(dW.double(), refW) <= tol and max_abs(db.double(), refb) <= tol
    dW2, none = hp.linear_bwd_wb(x, dy, bias=False)
    assert none is None and torch.equal(dW, dW2)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 3, 30 * 1201, 4 * 100003 + 1])
def test_prelu_bwd_matches_autograd(n):
    """genie_prelu_bwd (one-pass PReLU backward of the training path) against PyTorch's: dx bit-exact, the slope gradient to
    fp32 summation error; sizes cover the empty tensor, the sub-float4 tail and an odd length."""
    hp = engine.HipPath(3, 4, engine.csr_from_edges(torch.zeros((2, 0), dtype=torch.long), 3),
                        engine.csr_from_edges(torch.zeros((2, 0), dtype=torch.long), 4), device=DEV)
    g = torch.Generator(device=DEV).manual_seed(n)
    x = torch.randn(n, device=DEV, generator=g, requires_grad=True)
    a = torch.tensor([0.25], device=DEV, requires_grad=True)
    dy = torch.randn(n, device=DEV, generator=g)
    torch.nn.functional.prelu(x, a).backward(dy)
    dx, da = hp.prelu_bwd(x.detach(), dy, a.detach())
    assert torch.equal(dx, x.grad)
    ref = (dy.double() * x.detach().double() * (x.detach() < 0)).sum()
    assert abs(float(da[0]) - float(ref)) <= 1e-5 * max(1.0, float((dy * x.detach()).abs().sum()) ** 0.5)
    assert abs(float(a.grad[0]) - float(ref)) <= 1e-3 * max(1.0, abs(float(ref)))



@pytest.mark.parametrize("S,G,C", [(12, 60, 30), (21, 40, 15), (50, 300, 32)])
def test_nbr_mean_matches_torch_gathers(S, G, C):
    """genie_nbr_mean (neighbour means of arbitrary [P, C] rows on the product graph, used by the association heads) against the
    index-gather formulation; (21, 40) uses ragged graphs with an empty neighbourhood."""
    from tests.restatements import _mean_over_src, _mean_over_sta
    geom = synthetic.Geometry(S, G, L=100e3, n_query=5, seed=S + G)
    A_sta, A_src = geom.A_sta_sta, geom.A_src_src
    ragged = (S, G) == (21, 40)
    if ragged:
        rng = np.random.default_rng(3)
        A_sta = A_sta[:, (rng.random(A_sta.shape[1]) < 0.7) & (A_sta[1] != 4)]
        A_src = A_src[:, (rng.random(A_src.shape[1]) < 0.7) & (A_src[1] != 9)]
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(A_sta), S), engine.csr_from_edges(torch.from_numpy(A_src), G),
                        device=DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    x1 = torch.randn((S * G, C), device=DEV, generator=g)
    x2 = torch.randn((S * G, C), device=DEV, generator=g)
    o1, o2 = hp.nbr_mean(x1, x2)
    x3a, x3b = x1.view(G, S, C).cpu(), x2.view(G, S, C).cpu()
    ref1, ref2 = torch.zeros(G, S, C), torch.zeros(G, S, C)
    deg1, deg2 = torch.zeros(S), torch.zeros(G)
    for j, i in torch.from_numpy(A_sta).t().tolist():
        ref1[:, i] += x3a[:, j]; deg1[i] += 1
    for j, i in torch.from_numpy(A_src).t().tolist():
        ref2[i] += x3b[j]; deg2[i] += 1
    ref1 = ref1 / deg1.clamp(min=1).view(1, S, 1)
    ref2 = ref2 / deg2.clamp(min=1).view(G, 1, 1)
    assert max_abs(o1.cpu().view(G, S, C), ref1) <= 1e-6 and max_abs(o2.cpu().view(G, S, C), ref2) <= 1e-6
    if not ragged:
        t1 = _mean_over_sta(x1, graph.neighbour_table(A_sta, S).long().to(DEV), S, G)
        t2 = _mean_over_src(x2, graph.neighbour_table(A_src, G).long().to(DEV), S, G)
        assert max_abs(o1, t1) <= 1e-6 and max_abs(o2, t2) <= 1e-6


@pytest.mark.parametrize("name", ["tiny_6x40", "cfg1_20x500"])
def test_training_mode_forward_and_gradients_match_oracle_autograd(name):
    """a-8 (first pass): in train() mode with gradients enabled forward_fixed_source takes the differentiable formulation
    (neighbour means through genie_nbr_mean / genie_nbr_mean_bwd, dense algebra under autograd). Its outputs equal the fused
    HIP path and the gradient of a random linear functional of (y, x) w.r.t. every parameter of the path equals the oracle's
    autograd gradient (fp32 CPU) to 1e-5 of the gradient scale."""
    from oracle import genie_oracle as O
    c = Case(name)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.set_adjacencies_base(c.A_sta_sta, c.A_src_src, c.edge_attr.to(DEV), c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    args = (c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV),
            c.x_query.float().to(DEV), c.t_query.float().to(DEV))
    net.eval()
    with torch.no_grad():
        y_hip, x_hip = net.forward_fixed_source(*args)
    net.train()
    y, x = net.forward_fixed_source(*args)
    assert y.requires_grad and x.requires_grad
    assert max_abs(y.detach(), y_hip) <= 1e-6 and max_abs(x.detach(), x_hip) <= 1e-6
    g = torch.Generator().manual_seed(7)
    ay, ax = torch.randn(y.shape, generator=g), torch.randn(x.shape, generator=g)
    (y * ay.to(DEV)).sum().add((x * ax.to(DEV)).sum()).backward()
    # oracle autograd on the CPU
    w = {k: v.clone().requires_grad_(True) for k, v in c.weights.items()}
    A_in_sta, A_in_src, A_src_in_prod, _ = c.product_edges()
    yo, xo = O.forward_fixed_source(w, c.Slice, c.Mask, A_in_sta, A_in_src, c.edge_attr, A_src_in_prod, c.A_src_src,
                                    c.x_grid.float(), c.x_query.float(), c.t_query.float())
    ((yo * ay).sum() + (xo * ax).sum()).backward()
    checked = 0
    gmax = max(float(v.grad.abs().max()) for v in w.values() if v.grad is not None)
    for k, p in net.named_parameters():
        if w[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        ref = w[k].grad
        assert p.grad is not None, k
        # relative to the gradient's OWN scale (1e-5 of max(1, scale) hid a missing save once); scalars that are sums of cancelling
        # terms get a floor of 1e-3 of the largest gradient of the model
        tol = 2e-4 * max(float(ref.abs().max()), 1e-3 * gmax) + 1e-12
        assert max_abs(p.grad.cpu(), ref) <= tol, (k, max_abs(p.grad.cpu(), ref), tol)
        checked += 1
    assert checked >= 80          # every tensor of DataAggregation, Bipartite_ReadIn, SpatialAggregation1..3 and the read-outs


@pytest.mark.parametrize("name", ["edges_12x60", "edges_7x13", "abspos_12x60", "abspos_7x13", "edges_abspos_12x60", "subgraph_edges_14x50",
                                  "subgraph_abspos_14x50"])
@pytest.mark.parametrize("stage1", ["default", "f32"])
def test_training_step_of_the_other_model_definitions_matches_oracle_autograd(name, stage1, monkeypatch):
    """a-8 / a-9: the training step of `forward_fixed_source` under `use_updated_model_definition` (DataAggregationEdges,
    module.py:102-174) and `use_absolute_pos` (module.py:56-57, :1007). The forward is the inference kernels of those variants with
    the pre-activations kept; their static terms (mean edge features per station / source node; scaled positions) add only WEIGHT
    gradients, taken from per-station / per-source-node sums of the gradient rows the backward passes keep (k_gr_sum_*,
    k_static_dw). Every parameter gradient -- in the parameter's own [30, 68] / [15, 98] / [30, 14] shape -- equals the oracle's
    autograd (fp32 CPU) to 1e-5 of the gradient scale; an inference call after the step is unchanged (the station-order state of
    the position tables is per call)."""
    from oracle import genie_oracle as O
    if stage1 == "f32":
        monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    c = Case(name)
    kw = dict(use_updated_model_definition=c.edges_variant, use_absolute_pos=c.abspos_variant)      # (edges_abspos_12x60: both)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, **kw)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea, A_src_in_sta.to(DEV), c.A_src_src.to(DEV),
                        None, None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    args = (c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV),
            c.x_query.float().to(DEV), c.t_query.float().to(DEV))
    net.eval()
    with torch.no_grad():
        y_hip, x_hip = net.forward_fixed_source(*args)
    net.train()
    y, x = net.forward_fixed_source(*args)
    assert y.requires_grad and x.requires_grad
    assert max_abs(y.detach(), y_hip) <= 1e-5 and max_abs(x.detach(), x_hip) <= 1e-5
    assert max_abs(y.detach().cpu(), c.ref("y")) <= 1e-5 and max_abs(x.detach().cpu(), c.ref("x")) <= 1e-5
    g = torch.Generator().manual_seed(7)
    ay, ax = torch.randn(y.shape, generator=g), torch.randn(x.shape, generator=g)
    (y * ay.to(DEV)).sum().add((x * ax.to(DEV)).sum()).backward()
    net.eval()
    with torch.no_grad():
        y2, x2 = net.forward_fixed_source(*args)
    assert torch.equal(y2, y_hip) and torch.equal(x2, x_hip)
    # oracle autograd on the CPU
    w = {k: v.clone().requires_grad_(True) for k, v in c.weights.items()}
    Slice, okw = c.Slice, {}
    if c.abspos_variant:
        Slice = O.absolute_pos_inputs(Slice, c.locs.float(), c.x_grid.float(), A_src_in_sta)
    if c.edges_variant:
        okw["pos_rel"] = (O.edge_pos_features(c.locs.float(), A_in_sta, A_src_in_sta[0]),
                          O.edge_pos_features(c.x_grid.float(), A_in_src, A_src_in_sta[1]))
    yo, xo = O.forward_fixed_source(w, Slice, c.Mask, A_in_sta, A_in_src, c.edge_attr, A_src_in_prod, c.A_src_src,
                                    c.x_grid.float(), c.x_query.float(), c.t_query.float(), **okw)
    ((yo * ay).sum() + (xo * ax).sum()).backward()
    checked = 0
    gmax = max(float(v.grad.abs().max()) for v in w.values() if v.grad is not None)
    for k, p in net.named_parameters():
        if w[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        ref = w[k].grad
        assert p.grad is not None and tuple(p.grad.shape) == tuple(ref.shape), k
        tol = 2e-4 * max(float(ref.abs().max()), 1e-3 * gmax) + 1e-12      # own scale; floor for scalars that are sums of cancelling terms
        assert max_abs(p.grad.cpu(), ref) <= tol, (k, max_abs(p.grad.cpu(), ref), tol)
        checked += 1
    assert checked >= 80
    # the static-term columns themselves carry gradient (not a vacuous comparison of zeros)
    if c.edges_variant:
        assert float(w["DataAggregation.l1_t1_2.weight"].grad[:, 60:64].abs().max()) > 0
        assert float(w["DataAggregation.l2_t2_2.weight"].grad[:, 90:94].abs().max()) > 0
    if c.abspos_variant:
        assert float(w["DataAggregation.init_trns.weight"].grad[:, 4:10].abs().max()) > 0


@pytest.mark.parametrize("stage1", ["default", "f32"])
def test_training_step_on_an_irregular_product_graph_matches_oracle_autograd(stage1, monkeypatch):
    """`use_subgraph: True` (config.yaml:86): the training step of `forward_fixed_source` on an irregular product graph. Forward = the
    PCSR stage kernels with the pre-activations kept (k_stage1_h2<.., PCSR> / k_stage1_pcsr, k_stage2_pcsr, per-source-node segment
    sums of the messages); backward = k_train_b2<PCSR>, k_train_b1p, k_train_b0<PCSR>: tiles of 16 consecutive product nodes, the
    transposed means over the reversed PRODUCT-level graphs. Outputs equal the eval path and the reference's fixture; every parameter
    gradient equals the oracle's autograd (literal edge-list formulation on the same irregular graph) to 1e-5 of the gradient scale."""
    from oracle import genie_oracle as O
    if stage1 == "f32":
        monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    c = Case("subgraph_14x50")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = c.product_edges()
    ea = graph.GraphEdges(x=c.edge_attr.to(DEV), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea, A_src_in_sta.to(DEV), c.A_src_src.to(DEV),
                        None, None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    assert net._hip._n_prod is not None and net._hip.n_prod < c.S * c.G
    args = (c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV),
            c.x_query.float().to(DEV), c.t_query.float().to(DEV))
    net.eval()
    with torch.no_grad():
        y_hip, x_hip = net.forward_fixed_source(*args)
    net.train()
    y, x = net.forward_fixed_source(*args)
    assert y.requires_grad and x.requires_grad
    assert max_abs(y.detach(), y_hip) <= 1e-5 and max_abs(x.detach(), x_hip) <= 1e-5
    assert max_abs(y.detach().cpu(), c.ref("y")) <= 1e-5 and max_abs(x.detach().cpu(), c.ref("x")) <= 1e-5
    g = torch.Generator().manual_seed(7)
    ay, ax = torch.randn(y.shape, generator=g), torch.randn(x.shape, generator=g)
    (y * ay.to(DEV)).sum().add((x * ax.to(DEV)).sum()).backward()
    w = {k: v.clone().requires_grad_(True) for k, v in c.weights.items()}
    yo, xo = O.forward_fixed_source(w, c.Slice, c.Mask, A_in_sta, A_in_src, c.edge_attr, A_src_in_prod, c.A_src_src,
                                    c.x_grid.float(), c.x_query.float(), c.t_query.float())
    ((yo * ay).sum() + (xo * ax).sum()).backward()
    checked = 0
    gmax = max(float(v.grad.abs().max()) for v in w.values() if v.grad is not None)
    for k, p in net.named_parameters():
        if w[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        ref = w[k].grad
        assert p.grad is not None, k
        tol = 2e-4 * max(float(ref.abs().max()), 1e-3 * gmax) + 1e-12      # own scale; floor for scalars that are sums of cancelling terms
        assert max_abs(p.grad.cpu(), ref) <= tol, (k, max_abs(p.grad.cpu(), ref), tol)
        checked += 1
    assert checked >= 80
    net.eval()
    with torch.no_grad():
        y2, x2 = net.forward_fixed_source(*args)
    assert torch.equal(y2, y_hip) and torch.equal(x2, x_hip)


def test_all_zero_mask_gates_bipartite_sum():
    """m_p = max_c Mask[p,c] gates every message (module.py:229): Mask = 0 -> r_g = 0 -> out_g = PReLU(fc2.bias)."""
    c = Case("tiny_6x40")
    hp = make_engine(c)
    Mask = torch.zeros_like(c.Mask)
    _, _, bip = hp.path_fwd(c.Slice.to(DEV), Mask.to(DEV), c.edge_attr.to(DEV), c.x_grid.float().to(DEV), False, True)
    b = c.weights["Bipartite_ReadIn.fc2.bias"]
    a = c.weights["Bipartite_ReadIn.activate2.weight"]
    want = torch.where(b >= 0, b, a * b).view(1, -1).expand(c.G, -1)
    assert max_abs(bip.cpu(), want) <= 1e-7


def test_weight_updates_are_picked_up():
    """In-place parameter updates (optimizer steps, load_state_dict) must reach the HIP mirror."""
    c = Case("tiny_6x40")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV).eval()
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.set_adjacencies_base(c.A_sta_sta, c.A_src_src, c.edge_attr.to(DEV), c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    args = (c.Slice.to(DEV), c.Mask.to(DEV), None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV),
            c.x_query.float().to(DEV), c.t_query.float().to(DEV))
    with torch.no_grad():
        y0, _ = net.forward_fixed_source(*args)
        net.DataAggregation.init_trns.weight.mul_(1.5)
        net.SpatialAggregation3.fc2.bias.add_(0.25)
        y1, _ = net.forward_fixed_source(*args)
        net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
        y2, _ = net.forward_fixed_source(*args)
    assert max_abs(y0.cpu(), c.ref("y")) <= 1e-5
    assert max_abs(y1.cpu(), y0.cpu()) > 5e-6
    assert torch.equal(y2, y0)


def test_argument_validation():
    c = Case("tiny_6x40")
    hp = make_engine(c)
    with pytest.raises(ValueError):
        hp.path_fwd(c.Slice[:-1].to(DEV), c.Mask.to(DEV), c.edge_attr.to(DEV), c.x_grid.float().to(DEV))
    with pytest.raises(ValueError):
        hp.path_fwd(c.Slice, c.Mask.to(DEV), c.edge_attr.to(DEV), c.x_grid.float().to(DEV))  # CPU tensor
    with pytest.raises(Exception):
        hp.spatial_agg(4, torch.zeros(c.G, 30, device=DEV), c.x_grid.float().to(DEV))


def _weights_for_station_count(w, n_sta, n_sta_fixture=20):
    """The scaled `o1_20x500` weights give outputs of O(1) on THEIR 20 stations; the Bipartite read-in sums over the stations
    (module.py:224-229), so on 200 / 2000 stations the same weights give outputs of O(100+). PReLU is positively homogeneous: scaling
    `Bipartite_ReadIn.fc1` (weight and bias) by 20 / n_sta scales every term of the station sum by exactly that factor, the sum keeps the
    magnitude the fixture has, and everything after it sees the values it was scaled for."""
    w = dict(w)
    f = float(n_sta_fixture) / float(n_sta)
    for k in ("Bipartite_ReadIn.fc1.weight", "Bipartite_ReadIn.fc1.bias"):
        w[k] = w[k] * f
    return w


@pytest.mark.parametrize("n_picks_window,weights", [(50000, "cfg1_20x500"), (400, "cfg1_20x500"), (50000, "o1_20x500")])
def test_config2_full_size_properties(n_picks_window, weights):
    """BASELINE config 2 (200 stations / 10k grid / 50k picks) at full size through the drop-in class: bitwise determinism,
    finite outputs, and the Bipartite output, the path output `x_spatial` AND the outputs (y, x) against the structured oracle
    on the CPU (~25 s): intermediates 1e-5 x max(1, max|ref|), outputs 1e-5 absolute. Two windows: the headline one of 50 000
    picks, whose masks are saturated (Mask.mean() = 0.999997: `m_p = 1` everywhere), and a sparse one of 400 picks, where a
    third of the product nodes has an all-zero Mask row, so that the `mask.max(1)` gate of Bipartite_ReadIn (module.py:226-229)
    and the Mask inputs of DataAggregation really select at full size (VERDICT round 4 asked for ~5 000 picks; with the 3-s kernel
    25 picks per station in a 140-s window still leave 0.04 % all-zero rows, see the printed statistics).
    Two weight sets: the default-initialised ones of `cfg1_20x500` (max|y| 0.035: 1e-5 absolute is 3e-4 relative there) and the scaled
    `o1_20x500` ones (outputs of O(1): the station sum over 200 terms of the two-piece fp16 arithmetic meets a BINDING 1e-5; the test
    checks that the f16x2 kernels are the ones that ran). Printed next to the fp32 oracle: the fp64 oracle (the truth both deviate from)."""
    from oracle import genie_oracle as O
    S, G, n_picks, L, nq = synthetic.CONFIGS["cfg2_200x10k"]
    geom = synthetic.Geometry(S, G, L=L, n_query=2000, seed=1)
    win = synthetic.make_window(geom, n_picks_window, seed=2 if n_picks_window == n_picks else 9, window=0 if n_picks_window == n_picks else 7)
    zero_rows = float((win["Mask"].max(1) == 0).mean())
    print("config 2 window of %d picks: Mask.mean() %.6f, all-zero Mask rows %.4f" % (n_picks_window, float(win["Mask"].mean()), zero_rows))
    assert (zero_rows > 0.2) == (n_picks_window < 1000)
    c = Case(weights)
    w = _weights_for_station_count(c.weights, S) if weights == "o1_20x500" else c.weights
    sta_nbr = graph.neighbour_table(geom.A_sta_sta, S)
    src_nbr = graph.neighbour_table(geom.A_src_src, G)
    Slice, Mask = torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"])
    ea = torch.from_numpy(geom.edge_attr())
    pos = torch.from_numpy(geom.x_grid).float()
    xq, tq = torch.from_numpy(geom.x_query).float(), torch.from_numpy(geom.t_query).float()
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in w.items()})
    net.eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), ea.to(DEV),
                             torch.from_numpy(geom.locs).float().to(DEV), pos.to(DEV))
    hp = net._hip
    dSlice, dMask, dea, dpos = Slice.to(DEV), Mask.to(DEV), ea.to(DEV), pos.to(DEV)
    net._hip.sync_weights(net._path_params)
    out1, _, bip1 = hp.path_fwd(dSlice, dMask, dea, dpos, False, True)
    out2, _, bip2 = hp.path_fwd(dSlice, dMask, dea, dpos, False, True)
    assert torch.equal(out1, out2) and torch.equal(bip1, bip2)
    assert torch.isfinite(out1).all()
    with torch.no_grad():
        y, x = net.forward_fixed_source(dSlice, dMask, None, None, None, None, dpos, xq.to(DEV), tq.to(DEV))
        y2, x2 = net.forward_fixed_source(dSlice, dMask, None, None, None, None, dpos, xq.to(DEV), tq.to(DEV))
        o = O.forward_fixed_source_structured(w, Slice, Mask, sta_nbr, src_nbr, ea, torch.from_numpy(geom.A_src_src), pos, xq, tq,
                                              S, G, full=True)
    assert torch.equal(y, y2) and torch.equal(x, x2)
    assert max_abs(bip1.cpu(), o["bip"]) <= rel_tol(o["bip"])
    assert max_abs(out1.cpu(), o["sa3"]) <= rel_tol(o["sa3"])
    ey, ex = max_abs(y.cpu(), o["y"]), max_abs(x.cpu(), o["x"])
    info = hp.stage_precision()
    print("config 2 full size (%s, %d picks): max|y - oracle| %.3g  max|x - oracle| %.3g  (max|y| %.3g, max|x| %.3g, max|bip| %.3g; f16x2 %s)"
          % (weights, n_picks_window, ey, ex, float(o["y"].abs().max()), float(o["x"].abs().max()), float(o["bip"].abs().max()), info["f16x2_active"]))
    assert info["f16x2_active"], "the two-piece fp16 stage kernels are the arithmetic this test is about"
    assert y.shape == (G, 9, 1) and x.shape == (2000, 9, 1)
    if weights == "o1_20x500":
        assert float(o["y"].abs().max()) > 0.3, "the scaled weights must give outputs of O(1)"
        if n_picks_window == n_picks:       # the fp64 truth (one more oracle pass, ~40 s): where HIP and the fp32 CPU forward stand against it
            with torch.no_grad():
                o64 = O.forward_fixed_source_structured({k: v.double() for k, v in w.items()}, Slice.double(), Mask.double(), sta_nbr, src_nbr,
                                                        ea.double(), torch.from_numpy(geom.A_src_src), pos.double(), xq.double(), tq.double(),
                                                        S, G, full=True)
            print("config 2 full size (%s) vs the fp64 oracle: HIP y %.3g x %.3g bip %.3g | fp32 oracle y %.3g x %.3g bip %.3g"
                  % (weights, max_abs(y.cpu(), o64["y"]), max_abs(x.cpu(), o64["x"]), max_abs(bip1.cpu(), o64["bip"]),
                     max_abs(o["y"], o64["y"]), max_abs(o["x"], o64["x"]), max_abs(o["bip"], o64["bip"])))
            # BINDING: outputs of magnitude ~8, 1e-5 absolute = 1.3e-6 relative, against the truth (measured: y 3.8e-6, x 1.3e-6)
            assert max_abs(y.cpu(), o64["y"]) <= 1e-5 and max_abs(x.cpu(), o64["x"]) <= 1e-5
            # ... and against the fp32 CPU forward, which at this magnitude is itself 1.25e-5 from the truth (measured): two fp32
            # evaluations of the same function may differ by the tolerance plus the CPU's own deviation from the truth
            assert ey <= 1e-5 + max_abs(o["y"], o64["y"]) and ex <= 1e-5 + max_abs(o["x"], o64["x"])
            return
    assert ey <= 1e-5 and ex <= 1e-5                  # fp32 max-abs tolerance of BASELINE.json


@pytest.mark.parametrize("S,G,W,variant", [(40, 600, 2, None), (200, 300, 3, None), (17, 95, 4, None), (64, 1000, 8, None), (33, 257, 5, None),
                                           (40, 600, 3, "edges"), (40, 600, 3, "abspos"), (200, 300, 2, "edges")])
def test_sharded_kernels_virtual_ranks_match_unsharded(S, G, W, variant):
    """Source-node sharding on ONE GPU: W virtual ranks (halo rows, local CSR numbering, n_grid_ext > n_grid), the
    halo all-to-all replaced by direct copies. The result must equal the unsharded HIP path bit for bit (same kernels,
    same per-node arithmetic) and the oracle to tolerance. The RCCL collective itself is covered by tests/test_dist_cpu.py
    (gloo) and runs for real only on a multi-GPU node."""
    from genie_amd import dist as gdist
    from oracle import genie_oracle as O
    geom = synthetic.Geometry(S, G, L=200e3, n_query=20, seed=41 + W)
    win = synthetic.make_window(geom, 400, seed=42)
    # the two other model definitions on a shard: their static terms are per station / per source node of the EXTENDED list
    w = Case({None: "odd_33x257", "edges": "edges_12x60", "abspos": "abspos_12x60"}[variant]).weights
    if variant == "edges":
        w = module._split_edge_columns(w)
    if variant == "abspos":
        w = module._split_abs_columns(w)
    wd = {k: v.to(DEV) for k, v in w.items()}
    Slice, Mask = torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"])
    ea = torch.from_numpy(geom.edge_attr())
    pos = torch.from_numpy(geom.x_grid).float()
    locs = torch.from_numpy(geom.locs).float()
    sta_csr = engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S)
    # unsharded reference run
    hp = engine.HipPath(S, G, sta_csr, engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G),
                        grid_order=engine.sfc_order(geom.x_grid), device=DEV, sta_order=engine.sfc_order(geom.locs))
    if variant == "edges":
        hp.set_edge_features(locs.to(DEV), pos.to(DEV))
    if variant == "abspos":
        hp.set_absolute_pos(locs.to(DEV), pos.to(DEV))
    hp.set_weights(wd)
    out_ref, xl_ref, bip_ref = hp.path_fwd(Slice.to(DEV), Mask.to(DEV), ea.to(DEV), pos.to(DEV), True, True)
    # virtual ranks (same station processing order: the per-tile station sums then add in the same order)
    ranks = [gdist.ShardedPath(S, G, sta_csr, geom.A_src_src, geom.x_grid, W, r, DEV, pos_sta=geom.locs) for r in range(W)]
    rows = []
    for sp in ranks:
        if variant == "edges":
            sp.set_edge_features(locs, pos)
        if variant == "abspos":
            sp.set_absolute_pos(locs, pos)
        sp.set_weights(wd)
        ext = torch.from_numpy(sp.plan.ext_global)
        r = (ext.view(-1, 1) * S + torch.arange(S).view(1, -1)).reshape(-1)
        rows.append(r)
        sp._S, sp._M = sp.local.da_stage1(Slice[r].to(DEV), Mask[r].to(DEV))
    for sp in ranks:                                  # halo exchange by direct copies
        p, wv = sp.plan, sp.wv_view()
        off = p.n_own
        for q in range(W):
            need = p.need[p.rank][q]
            if need.size == 0:
                continue
            src = ranks[q]
            loc = torch.from_numpy(src.plan.global_to_local[need]).to(DEV)
            blocks = src.wv_view()[: src.plan.n_own * S].view(src.plan.n_own, S * 16).index_select(0, loc)
            wv[off * S:(off + need.size) * S] = blocks.view(-1, 16)
            off += need.size
        assert off == p.n_ext
    bip = torch.empty((G, 15), device=DEV)
    xl = torch.empty((G * S, 30), device=DEV)
    for sp, r in zip(ranks, rows):
        p = sp.plan
        own_rows = r[: p.n_own * S]
        x_latent, bip_own = sp.local.da_stage2_bipartite(sp._M[: p.n_own * S], ea[own_rows].to(DEV), want_x_latent=True)
        bip[torch.from_numpy(p.own_global).to(DEV)] = bip_own
        xl[own_rows.to(DEV)] = x_latent
    assert torch.equal(xl, xl_ref)
    assert torch.equal(bip, bip_ref)
    o = bip
    for layer in (1, 2, 3):
        o = ranks[0].full.spatial_agg(layer, o, pos.to(DEV))
    assert torch.equal(o, out_ref)
    assert min(sp.plan.n_halo for sp in ranks) > 0


@pytest.mark.parametrize("n_picks_window,weights", [(500000, "cfg1_20x500"), (4000, "cfg1_20x500"), (500000, "o1_20x500")])
def test_config4_shape_two_virtual_ranks_vs_unsharded_generic_kernels_and_oracle(monkeypatch, n_picks_window, weights):
    """BASELINE config 4 at its full shape (2000 stations x 50 000 source nodes = 10^8 product nodes, 500 000 picks) on one
    GPU: (1) the unsharded fast path; (2) the same window through the generic CSR kernels (64-bit row addressing, no f16x2, no
    pipelining): Bipartite output equal to fp32 summation-order error; (3) two virtual ranks of the source-node sharding with
    the sub-range launch schedule of genie_amd.dist.ShardedPath.front (halo rows of `wv` copied between the ranks'
    workspaces instead of the RCCL all-to-all): Bipartite output and x_spatial BITWISE equal to the unsharded run; (4) the
    oracle's arithmetic on a sample of source nodes (their two-hop neighbourhood), 1e-5 x max|ref| (the station sum over 2000
    terms drifts 1.7e-4 at max|bip| 313 between two correct fp32 evaluations, tests/golden/s2000_2000x24.npz).
    `weights`: the default-initialised `cfg1_20x500` set, and the scaled `o1_20x500` set, with which the read-outs (y, x) of the two-piece
    fp16 path and of the fp32-MFMA generic path are O(1) and are compared at a BINDING 1e-5 absolute (the CPU oracle cannot evaluate
    the full 50 000-node tail: the fp32 kernels, themselves checked against the oracle on the sampled nodes, stand in for it)."""
    import gc
    from genie_amd import dist as gdist
    from tests.util import oracle_bipartite_for_nodes
    S, G, n_picks, L, nq = synthetic.CONFIGS["cfg4_2000x50k"]
    geom = synthetic.Geometry(S, G, L=L, n_query=512, seed=1)
    # second window: 4 000 picks on 2 000 stations, masks that gate (most product nodes have an all-zero Mask row)
    P = synthetic.make_picks(geom, n_picks_window, seed=2 if n_picks_window == n_picks else 9)
    w = Case(weights).weights
    if weights == "o1_20x500":
        w = _weights_for_station_count(w, S)
    wd = {k: v.to(DEV) for k, v in w.items()}
    CH = 2048
    dS = torch.empty((S * G, 4), dtype=torch.float32, device=DEV)
    dM = torch.empty((S * G, 4), dtype=torch.float32, device=DEV)
    dea = torch.empty((S * G, 3), dtype=torch.float32, device=DEV)
    for g0 in range(0, G, CH):
        sl, mk = synthetic.make_slice_mask(geom, P, 0.0, g_slice=slice(g0, min(G, g0 + CH)))
        dS[g0 * S:g0 * S + sl.shape[0]] = torch.from_numpy(sl).to(DEV)
        dM[g0 * S:g0 * S + mk.shape[0]] = torch.from_numpy(mk).to(DEV)
        dea[g0 * S:g0 * S + sl.shape[0]] = torch.from_numpy(geom.edge_attr(slice(g0, min(G, g0 + CH)))).to(DEV)
    zero_rows = float((dM.max(1)[0] == 0).float().mean())
    print("config 4 window of %d picks: Mask.mean() %.6f, all-zero Mask rows %.4f" % (n_picks_window, float(dM.mean()), zero_rows))
    assert (zero_rows > 0.2) == (n_picks_window < 100000)
    pos = torch.from_numpy(geom.x_grid).float().to(DEV)
    sta_csr = engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S)
    src_csr = engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G)

    xq = torch.from_numpy(geom.x_query).float().to(DEV)
    tq = torch.from_numpy(geom.t_query).float().to(DEV)
    knn = engine.knn_device(pos, xq, 10)
    f16x2 = []

    def unsharded():
        hp = engine.HipPath(S, G, sta_csr, src_csr, grid_order=engine.sfc_order(geom.x_grid), device=DEV,
                            sta_order=engine.sfc_order(geom.locs))
        hp.set_weights(wd)
        hp.set_scale_t(9.0)
        out, _, bip = hp.path_fwd(dS, dM, dea, pos, False, True)
        y, x = hp.readout_grid(out, tq), hp.readout_query(out, pos, xq, knn, tq)
        f16x2.append(hp.stage_precision()["f16x2_active"])
        torch.cuda.synchronize()
        del hp
        gc.collect()
        torch.cuda.empty_cache()
        return out, bip, y, x

    out_ref, bip_ref, y_ref, x_ref = unsharded()
    assert torch.isfinite(out_ref).all() and torch.isfinite(bip_ref).all()
    # (2) generic kernels
    monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    out_gen, bip_gen, y_gen, x_gen = unsharded()
    monkeypatch.setattr(engine, "STAGE_PRECISION", "auto")
    assert f16x2 == [True, False], "the first run must be the two-piece fp16 kernels, the second the fp32-MFMA ones"
    scale = float(bip_ref.abs().max())
    print("config 4 shape (%s): max|bip| %.4g, fast vs generic kernels %.3g; read-outs max|y| %.3g max|x| %.3g, fast vs generic y %.3g x %.3g"
          % (weights, scale, max_abs(bip_ref, bip_gen), float(y_gen.abs().max()), float(x_gen.abs().max()), max_abs(y_ref, y_gen),
             max_abs(x_ref, x_gen)))
    assert max_abs(bip_ref, bip_gen) <= 1e-5 * max(1.0, scale)
    assert max_abs(out_ref, out_gen) <= 1e-5 * max(1.0, float(out_ref.abs().max()))
    assert max_abs(y_ref, y_gen) <= 1e-5 and max_abs(x_ref, x_gen) <= 1e-5          # BASELINE.json's tolerance on the outputs
    if weights == "o1_20x500":
        assert float(y_gen.abs().max()) > 0.3, "the scaled weights must give outputs of O(1)"
    del out_gen, bip_gen, y_gen, x_gen
    # (4) oracle on a sample of source nodes
    sample = np.array([0, 777, 25000, 49999])
    o_bip, _ = oracle_bipartite_for_nodes(w, geom, P, sample)
    err = max_abs(bip_ref[torch.from_numpy(sample).to(DEV)].cpu(), o_bip)
    print("config 4 shape: |bip - oracle| on 4 source nodes %.3g (max|ref| %.4g)" % (err, float(o_bip.abs().max())))
    assert err <= 1e-5 * max(1.0, float(o_bip.abs().max()))
    # (3) two virtual ranks, the schedule of ShardedPath.front with direct copies for the exchange
    W = 2
    ranks = [gdist.ShardedPath(S, G, sta_csr, geom.A_src_src, geom.x_grid, W, r, DEV, pos_sta=geom.locs) for r in range(W)]
    ins = []
    for sp in ranks:
        sp.set_weights(wd)
        p = sp.plan
        ext = torch.from_numpy(p.ext_global).to(DEV)
        rows = (ext.view(-1, 1) * S + torch.arange(S, device=DEV).view(1, -1)).reshape(-1)
        Se, Me, ea_own = dS[rows], dM[rows], dea[rows[: p.n_own * S]]
        ins.append((Se, Me, ea_own))
        (s0, s1) = p.r_send
        sp.local.da_stage1_range(Se, Me, s0, s1, True)
        sp.local.da_stage1_range(Se, Me, s1, p.n_own, False)
        assert p.n_halo > 0 and 0 < s1 < p.n_own and p.r_need[1] < p.n_own
    for sp in ranks:
        p, wv = sp.plan, sp.wv_view()
        off = p.n_own
        for q in range(W):
            need = p.need[p.rank][q]
            if need.size == 0:
                continue
            src = ranks[q]
            loc = torch.from_numpy(src.plan.global_to_local[need]).to(DEV)
            wv[off * S:(off + need.size) * S] = src.wv_view()[: src.plan.n_own * S].view(src.plan.n_own, S * 16).index_select(0, loc).view(-1, 16)
            off += need.size
        assert off == p.n_ext
    bip = torch.empty((G, 15), device=DEV)
    for sp, (Se, Me, ea_own) in zip(ranks, ins):
        p = sp.plan
        (n0, n1), n = p.r_need, p.n_own
        Mo = Me[: n * S]
        sp.local.da_stage2_partials_range(Mo, ea_own, 0, n0)
        sp.local.da_stage2_partials_range(Mo, ea_own, n1, n)
        sp.local.da_stage2_partials_range(Mo, ea_own, n0, n1)
        bip[torch.from_numpy(p.own_global).to(DEV)] = sp.local.bipartite_readout()
    assert torch.equal(bip, bip_ref)
    assert torch.equal(ranks[0].full.spatial_agg3(bip, pos), out_ref)


def test_apply_loop_matches_oracle_windows():
    """The sliding-window caller (process_continuous_days.py:761-810): Out_2 stacked on the GPU over the kept windows
    equals the same stacking of the oracle's per-window outputs."""
    from genie_amd import apply
    from oracle import genie_oracle as O
    S, G = 12, 80
    geom = synthetic.Geometry(S, G, L=60e3, n_query=15, seed=51)
    P = synthetic.make_picks(geom, 150, seed=52)
    P[:, 0] = P[:, 0] * 0.2 + 300.0                 # squeeze into a short interval -> a handful of windows
    c = Case("tiny_6x40")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                             torch.from_numpy(geom.edge_attr()).to(DEV), torch.from_numpy(geom.locs).float().to(DEV),
                             torch.from_numpy(geom.x_grid).float().to(DEV))
    Out_2, times = apply.apply_windows(net, geom, P, step_size="half", min_required_picks=5)
    assert 2 <= len(times) <= 40
    tsteps, offsets, step, n_overlap, dt_win = apply.window_schedule(P[:, 0], geom.max_t, t_win=6.0, step_size="half")
    tsteps_abs = np.arange(tsteps.min() - 3.0, tsteps.max() + 3.0 + dt_win, dt_win)
    want = torch.zeros(Out_2.shape)
    sta_nbr = graph.neighbour_table(geom.A_sta_sta, S)
    src_nbr = graph.neighbour_table(geom.A_src_src, G)
    for t0 in times:
        sel = (P[:, 0] > t0 - 6.0) & (P[:, 0] < t0 + geom.max_t + 6.0)
        Slice, Mask = synthetic.make_slice_mask(geom, P[sel], t0)
        _, x = O.forward_fixed_source_structured(c.weights, torch.from_numpy(Slice), torch.from_numpy(Mask), sta_nbr, src_nbr,
                                                 torch.from_numpy(geom.edge_attr()), torch.from_numpy(geom.A_src_src),
                                                 torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(),
                                                 torch.from_numpy(offsets.reshape(-1, 1)).float(), S, G)
        ip = np.abs(tsteps_abs.reshape(-1, 1) - (t0 + offsets).reshape(1, -1)).argmin(0)
        want[:, ip[:-1]] += x[:, :-1, 0] / 2.0
    assert max_abs(Out_2.cpu(), want) <= 1e-5


@pytest.mark.parametrize("name", ["embed_sign_14x60_a", "embed_sign_14x60_b"])
def test_device_embedding_with_sign_input_matches_reference(name):
    """`use_sign_input: True` (config.yaml:93; process_utils.py:610-614): genie_set_sign_input(1) makes the embedding kernel multiply each
    feature by the sign of the negative forward difference of the series it reads (any-phase series for columns 0, 1; the P / S
    series for columns 2, 3). Fixtures from the reference's extract_input_from_data with the flag set; the split rows written for
    stage 1 carry the same signed values (path output equal to the one computed from the returned Slice / Mask)."""
    import os
    from tests.util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    S, G = int(z["n_sta"]), int(z["n_grid"])
    t0, max_t, sig, dt = float(z["t0"]), float(z["max_t"]), float(z["kernel_sig_t"]), float(z["dt"])
    geom = synthetic.Geometry(S, G, L=90e3, n_query=5, seed=61)
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G), device=DEV)
    hp.set_sign_input(True)
    P = z["P"]
    sel = (P[:, 0] > t0 - 2.0 * sig) & (P[:, 0] < t0 + max_t + 2.0 * sig)
    Ps = P[sel]
    args = (torch.from_numpy(Ps[:, 0].copy()).to(DEV), torch.from_numpy(Ps[:, 1].astype(np.int32)).to(DEV),
            torch.from_numpy(Ps[:, 4].astype(np.int32)).to(DEV), t0, max_t, sig, dt, torch.from_numpy(z["trv_times"].reshape(-1, 2)).to(DEV))
    Slice, Mask = hp.embed_window(*args)
    assert float((Slice.cpu() - torch.from_numpy(z["Slice"])).abs().max()) <= 1e-6
    assert torch.equal(Mask.cpu(), torch.from_numpy(z["Mask"].astype(np.float32)))
    assert int((Slice < -0.5).sum()) > 20 and int((Slice > 0.5).sum()) > 20
    hp.set_sign_input(False)
    S0, _ = hp.embed_window(*args)
    assert float((S0.abs() - Slice.abs()).abs().max()) == 0.0 and float(S0.min()) >= 0.0


@pytest.mark.parametrize("builder", ["cartesian", "subgraph_lists", "subgraph_positions"])
def test_sign_input_flag_reaches_every_context_builder(builder):
    """a model built with `use_sign_input=True` must hand the flag to the HIP context whichever `set_adjacencies*`
    form built it (the Cartesian one, the `use_subgraph` one from product edge lists, the device builder from positions): the
    embedding of the MODULE's context equals embed_oracle.extract_input_from_data(..., use_sign_input=True) on that graph's product
    nodes (process_utils.py:610-614), and carries negative features."""
    import os
    from oracle import embed_oracle as E
    from tests.util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "embed_sign_14x60_a.npz"))
    S, G = int(z["n_sta"]), int(z["n_grid"])
    t0, max_t, sig, dt = float(z["t0"]), float(z["max_t"]), float(z["kernel_sig_t"]), float(z["dt"])
    geom = synthetic.Geometry(S, G, L=90e3, n_query=5, seed=61)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_sign_input=True)
    net.eval()
    if builder == "cartesian":
        A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
        pairs = A_src_in_sta.numpy()
        ea = geom.edge_attr()
    elif builder == "subgraph_lists":
        d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
        keep = np.zeros(d.shape, dtype=bool)
        keep[np.arange(G)[:, None], np.argsort(d, axis=1)[:, :5]] = True
        src_i, sta_i = np.nonzero(keep)
        pairs = np.stack((sta_i, src_i))
        A_in_sta, A_in_src, A_src_in_prod = graph.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
        A_src_in_sta = torch.from_numpy(pairs).long()
        ea = geom.edge_attr().reshape(G, S, 3)[src_i, sta_i]
    if builder == "subgraph_positions":
        _, _, prs = net.set_adjacencies_subgraph_from_positions(t(geom.locs), t(geom.x_grid), max_deg_offset=0.25, k_nearest_pairs=4)
        pairs = prs.cpu().numpy()
        assert pairs.shape[1] < S * G
    else:
        gea = graph.GraphEdges(x=t(ea), edge_index=A_src_in_prod.to(DEV))
        net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), gea, gea, A_src_in_sta.to(DEV), torch.from_numpy(geom.A_src_src).to(DEV),
                            None, None, None, None, t(geom.locs), t(geom.x_grid))
    P = z["P"]
    sel = (P[:, 0] > t0 - 2.0 * sig) & (P[:, 0] < t0 + max_t + 2.0 * sig)
    Ps = P[sel]
    trv = z["trv_times"]
    Slice, Mask = net._hip.embed_window(torch.from_numpy(Ps[:, 0].copy()).to(DEV), torch.from_numpy(Ps[:, 1].astype(np.int32)).to(DEV),
                                        torch.from_numpy(Ps[:, 4].astype(np.int32)).to(DEV), t0, max_t, sig, dt,
                                        t(trv[pairs[1], pairs[0]]))
    want_S, want_M = E.extract_input_from_data(P, t0, np.arange(S), S, trv, pairs, max_t, sig, dt, use_sign_input=True)
    assert float((Slice.cpu() - torch.from_numpy(want_S)).abs().max()) <= 1e-6
    assert torch.equal(Mask.cpu(), torch.from_numpy(want_M))
    assert int((Slice < -0.5).sum()) > 5


@pytest.mark.parametrize("name", ["embed_14x60_a", "embed_14x60_b"])
def test_device_embedding_matches_reference_and_oracle(name):
    """Pick -> Slice/Mask embedding kernels (f-1) against the reference's extract_input_from_data golden vectors
    (process_utils.py:460-642). fp32 tolerance 1e-6 on Slice (float cast of a float64 exp), Mask exact."""
    import os
    from tests.util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    S, G = int(z["n_sta"]), int(z["n_grid"])
    t0, max_t, sig, dt = float(z["t0"]), float(z["max_t"]), float(z["kernel_sig_t"]), float(z["dt"])
    geom = synthetic.Geometry(S, G, L=90e3, n_query=5, seed=61)
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G), device=DEV)
    P = z["P"]
    sel = (P[:, 0] > t0 - 2.0 * sig) & (P[:, 0] < t0 + max_t + 2.0 * sig)           # process_utils.py:476
    Ps = P[sel]
    Slice, Mask = hp.embed_window(torch.from_numpy(Ps[:, 0].copy()).to(DEV), torch.from_numpy(Ps[:, 1].astype(np.int32)).to(DEV),
                                  torch.from_numpy(Ps[:, 4].astype(np.int32)).to(DEV), t0, max_t, sig, dt,
                                  torch.from_numpy(z["trv_times"].reshape(-1, 2)).to(DEV))
    assert float((Slice.cpu() - torch.from_numpy(z["Slice"])).abs().max()) <= 1e-6
    assert torch.equal(Mask.cpu(), torch.from_numpy(z["Mask"].astype(np.float32)))
    # no picks at all -> all zero
    e = torch.zeros(0, device=DEV)
    S0, M0 = hp.embed_window(e.double(), e.int(), e.int(), t0, max_t, sig, dt, torch.from_numpy(z["trv_times"].reshape(-1, 2)).to(DEV))
    assert float(S0.abs().max()) == 0.0 and float(M0.abs().max()) == 0.0
    # use_phase_types: False (config.yaml:91): every pick enters with phase 0 (process_continuous_days.py:562-563) and the
    # phase-informed columns 2, 3 are zero (:783-786); columns 0, 1 (any-phase series) are those of the reference's embedding
    hp.set_phase_types(False)
    S1, M1 = hp.embed_window(torch.from_numpy(Ps[:, 0].copy()).to(DEV), torch.from_numpy(Ps[:, 1].astype(np.int32)).to(DEV),
                             torch.zeros(Ps.shape[0], dtype=torch.int32, device=DEV), t0, max_t, sig, dt,
                             torch.from_numpy(z["trv_times"].reshape(-1, 2)).to(DEV))
    assert float((S1[:, :2].cpu() - torch.from_numpy(z["Slice"][:, :2])).abs().max()) <= 1e-6
    assert torch.equal(M1[:, :2].cpu(), torch.from_numpy(z["Mask"][:, :2].astype(np.float32)))
    assert float(S1[:, 2:].abs().max()) == 0.0 and float(M1[:, 2:].abs().max()) == 0.0


def _day_like_picks(S, t_lo, t_hi, seed, burst_at=None, geom=None):
    """Picks of a continuous day around [t_lo, t_hi]: the background rate of BSSA NC data (~250 picks / station / day),
    so that MOST stations have no pick in a window, plus one synthetic event (P and S arrivals on 80 % of the stations) for dense rows."""
    rng = np.random.default_rng(seed)
    n = max(8, int(250.0 * S * (t_hi - t_lo) / 86400.0))
    rows = [np.stack([rng.uniform(t_lo, t_hi, n), rng.integers(0, S, n).astype(np.float64), np.ones(n), np.ones(n),
                      rng.integers(0, 2, n).astype(np.float64)], axis=1)]
    if burst_at is not None:
        g = int(rng.integers(0, geom.n_grid))
        tt = geom.travel_times(slice(g, g + 1))[0]
        for ph in (0, 1):
            keep = rng.random(S) < 0.8
            t = burst_at + tt[keep, ph] + rng.normal(0.0, 0.1, int(keep.sum()))
            rows.append(np.stack([t, np.nonzero(keep)[0].astype(np.float64), np.ones_like(t), np.ones_like(t), np.full_like(t, ph)], axis=1))
    P = np.concatenate(rows, axis=0)
    return P[np.argsort(P[:, 0], kind="stable")]


def _embed_vs_oracle(Slice, Mask, ref_S, ref_M, what):
    """Slice to 1e-6 (the float cast of a float64 exp), Mask exact except where the reference's value sits within 2e-6 of the 0.01
    threshold (a last-bit difference of the fp32 exponential may fall on the other side there); returns the number of such rows."""
    err = float((Slice - ref_S).abs().max())
    diff = Mask != ref_M
    edge = (ref_S.abs() - 0.01).abs() <= 2e-6
    n_edge = int((diff & edge).sum())
    print("%s: max|Slice - oracle| %.3g, Mask mismatches %d (all within 2e-6 of the threshold: %s), nonzero Slice rows %.4f"
          % (what, err, int(diff.sum()), bool((diff & ~edge).sum() == 0), float((ref_S.abs().max(1)[0] > 0).float().mean())))
    assert err <= 1e-6, what
    assert int((diff & ~edge).sum()) == 0, what
    return n_edge


def test_device_embedding_full_size_config5_stream_200x10000_vs_oracle():
    """`extract_input_from_data` (process_utils.py:460-642, gather at :599-608) on the device at BASELINE config 2 / 5's full size (200
    stations x 10 000 source nodes = 2 000 000 product nodes), inside a 3-window stream at 1 s stride: (1) every window's (Slice, Mask)
    from `genie_embed_window` AND from the stream's own `genie_embed_window_split` (presplit) against oracle/embed_oracle.py on all
    rows; (2) the stream itself (`apply_windows_device`: device embedding -> push_window -> batched tail -> Out_2) against the oracle
    chain embed -> forward -> stacking on the same three windows."""
    from genie_amd import apply
    from oracle import embed_oracle as E
    from oracle import genie_oracle as O
    S, G = 200, 10000
    geom = synthetic.Geometry(S, G, L=300e3, n_query=300, seed=1)
    trv = geom.travel_times().astype(np.float32)
    max_t = float(np.ceil(trv.max() + 1.0))
    sig, dt = 3.0, 0.3
    times = 40000.3 + 1.0 * np.arange(3)
    P = _day_like_picks(S, times[0] - 30.0, times[-1] + max_t + 30.0, seed=95, burst_at=times[0] + 20.0, geom=geom)
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    ea = torch.from_numpy(geom.edge_attr())
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), ea.to(DEV),
                             torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV))
    A = np.stack([np.tile(np.arange(S), G), np.repeat(np.arange(G), S)], axis=0)
    d_t = torch.from_numpy(P[:, 0].copy()).to(DEV)
    d_sta = torch.from_numpy(P[:, 1].astype(np.int32)).to(DEV)
    d_ph = torch.from_numpy(P[:, 4].astype(np.int32)).to(DEV)
    d_trv = net.node_rows(trv, 2)
    refs = []
    for t0 in times:
        rS, rM = E.extract_input_from_data(P, float(t0), np.arange(S), S, trv, A, max_t, sig, dt)
        refs.append((torch.from_numpy(rS), torch.from_numpy(rM)))
        lo, hi = apply.picks_in_embed_range(P[:, 0], float(t0), max_t, sig)
        for presplit in (False, True):
            Sl, Mk = net.embed_window(d_t[lo:hi], d_sta[lo:hi], d_ph[lo:hi], float(t0), max_t, sig, dt, d_trv, presplit=presplit)
            _embed_vs_oracle(Sl.cpu(), Mk.cpu(), refs[-1][0], refs[-1][1], "config 2/5 full size, t0 %.1f, presplit %s" % (t0, presplit))
    assert 0.0 < float((refs[0][1].max(1)[0] == 0).float().mean()) < 1.0         # rows with and without a pick in reach
    # (2) the stream
    tsteps, offsets, step, n_overlap, dt_win = apply.window_schedule(P[:, 0], max_t, t_win=6.0, step_size="half")
    tsteps_abs = np.arange(times.min() - 6.0, times.max() + 6.0 + dt_win, dt_win)
    Out_2, used = apply.apply_windows_device(net, geom, P, trv, tsteps_abs=tsteps_abs, step_size="half", max_t=max_t, kernel_sig_t=sig,
                                             dt_embed=dt, times=times, tail_batch=3)
    assert len(used) == 3
    sta_nbr, src_nbr = graph.neighbour_table(geom.A_sta_sta, S), graph.neighbour_table(geom.A_src_src, G)
    want = np.zeros(tuple(Out_2.shape))
    tq = torch.from_numpy(offsets.reshape(-1, 1)).float()
    with torch.no_grad():
        for t0, (rS, rM) in zip(used, refs):
            _, x = O.forward_fixed_source_structured(c.weights, rS, rM, sta_nbr, src_nbr, ea, torch.from_numpy(geom.A_src_src),
                                                     torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(), tq, S, G)
            cols, keep = apply.window_columns(tsteps_abs, float(t0), offsets, True)
            want[:, cols] += x[:, keep, 0].numpy() / n_overlap
    err = max_abs(Out_2.cpu(), torch.from_numpy(want))
    print("config 5 stream at full size (3 windows): max|Out_2 - oracle chain| %.3g (max|Out_2| %.3g)" % (err, float(np.abs(want).max())))
    assert float(np.abs(want).max()) > 1e-3 and err <= 1e-5


def test_device_embedding_config4_shape_sampled_source_nodes_vs_oracle():
    """The same embedding at BASELINE config 4's shape (2000 stations x 50 000 source nodes = 10^8 product nodes; the 800 MB travel-time
    table and the 3.2 GB of Slice + Mask live on the device only): the rows of 64 sampled source nodes (first, last, random) against
    oracle/embed_oracle.py, which takes any list of product nodes (`A_src_in_sta`). 64-bit row offsets, the per-station series of 2000
    stations, most of them without a pick."""
    from genie_amd import apply
    from oracle import embed_oracle as E
    S, G = 2000, 50000
    geom = synthetic.Geometry(S, G, L=1000e3, n_query=8, seed=1)
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G), device=DEV)
    d_trv = torch.empty((G * S, 2), dtype=torch.float32, device=DEV)
    CH = 2048
    for g0 in range(0, G, CH):
        tt = geom.travel_times(slice(g0, min(G, g0 + CH))).astype(np.float32)
        d_trv[g0 * S:g0 * S + tt.shape[0] * S] = torch.from_numpy(tt.reshape(-1, 2)).to(DEV)
    max_t = float(np.ceil(geom.max_t + 1.0))
    sig, dt, t0 = 3.0, 0.3, 51234.7
    P = _day_like_picks(S, t0 - 30.0, t0 + max_t + 30.0, seed=96, burst_at=t0 + 40.0, geom=geom)
    lo, hi = apply.picks_in_embed_range(P[:, 0], t0, max_t, sig)
    Sl, Mk = hp.embed_window(torch.from_numpy(P[lo:hi, 0].copy()).to(DEV), torch.from_numpy(P[lo:hi, 1].astype(np.int32)).to(DEV),
                             torch.from_numpy(P[lo:hi, 4].astype(np.int32)).to(DEV), t0, max_t, sig, dt, d_trv)
    rng = np.random.default_rng(97)
    sample = np.unique(np.concatenate(([0, 1, G - 1, 26843], rng.integers(0, G, 60))))           # (26843 * 2000 * 16 B > 2^32 / 5: past 32-bit offsets)
    trv_s = geom.travel_times(sample).astype(np.float32)                                           # [n, S, 2]
    A = np.stack([np.tile(np.arange(S), sample.size), np.repeat(np.arange(sample.size), S)], axis=0)
    rS, rM = E.extract_input_from_data(P, t0, np.arange(S), S, trv_s, A, max_t, sig, dt)
    rows = torch.from_numpy((sample.reshape(-1, 1) * S + np.arange(S).reshape(1, -1)).reshape(-1)).to(DEV)
    _embed_vs_oracle(Sl[rows].cpu(), Mk[rows].cpu(), torch.from_numpy(rS), torch.from_numpy(rM), "config 4 shape, %d sampled source nodes" % sample.size)
    assert float(np.abs(rS).max()) > 0.5
    # rows that no sample covers: finite, inside [0, 1], Mask = (|Slice| > 0.01) everywhere (one pass over all 10^8 rows on the device)
    assert bool(torch.isfinite(Sl).all()) and float(Sl.min()) >= 0.0 and float(Sl.max()) <= 1.0
    assert torch.equal(Mk, (Sl.abs() > 0.01).float())


@pytest.mark.parametrize("batch,S,G,n_picks,step_size", [(1, 10, 70, 120, "half"), (4, 10, 70, 120, "half"), (4, 40, 150, 700, "full"),
                                                         (3, 17, 64, 300, "partial")])
def test_device_apply_loop_matches_oracle(batch, S, G, n_picks, step_size):
    """GPU-only apply loop (device embedding + forward + Out_2 stacking; one tail per window, or tails batched 4 windows at a
    time) vs the oracle chain embed_oracle.extract_input_from_data -> genie_oracle.forward_fixed_source_structured -> same
    stacking."""
    from genie_amd import apply
    from oracle import embed_oracle as E
    from oracle import genie_oracle as O
    geom = synthetic.Geometry(S, G, L=60e3, n_query=12, seed=71)
    P = synthetic.make_picks(geom, n_picks, seed=72)
    P[:, 0] = P[:, 0] * 0.25 + 5000.0
    P = P[np.argsort(P[:, 0], kind="stable")]
    trv = geom.travel_times().astype(np.float32)
    c = Case("tiny_6x40")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                             torch.from_numpy(geom.edge_attr()).to(DEV), torch.from_numpy(geom.locs).float().to(DEV),
                             torch.from_numpy(geom.x_grid).float().to(DEV))
    max_t = float(np.ceil(trv.max() + 1.0))
    Out_2, times = apply.apply_windows_device(net, geom, P, trv, step_size=step_size, min_required_picks=5, max_t=max_t,
                                              tail_batch=batch)
    assert 2 <= len(times) <= 60
    tsteps, offsets, step, n_overlap, dt_win = apply.window_schedule(P[:, 0], max_t, t_win=6.0, step_size=step_size)
    tsteps_abs = np.arange(tsteps.min() - 3.0, tsteps.max() + 3.0 + dt_win, dt_win)
    A = np.stack([np.tile(np.arange(S), G), np.repeat(np.arange(G), S)], axis=0)
    sta_nbr = graph.neighbour_table(geom.A_sta_sta, S)
    src_nbr = graph.neighbour_table(geom.A_src_src, G)
    want = torch.zeros(Out_2.shape)
    for t0 in times:
        Slice, Mask = E.extract_input_from_data(P, float(t0), np.arange(S), S, trv, A, max_t, 3.0, 0.3)
        _, x = O.forward_fixed_source_structured(c.weights, torch.from_numpy(Slice), torch.from_numpy(Mask), sta_nbr, src_nbr,
                                                 torch.from_numpy(geom.edge_attr()), torch.from_numpy(geom.A_src_src),
                                                 torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(),
                                                 torch.from_numpy(offsets.reshape(-1, 1)).float(), S, G)
        cols, keep = apply.window_columns(tsteps_abs, float(t0), offsets, step_size == "half")      # (process_continuous_days.py:766,797-805)
        want[:, cols] += x[:, keep, 0] / n_overlap
    assert float(want.abs().max()) > 0
    assert max_abs(Out_2.cpu(), want) <= 1e-5


@pytest.mark.parametrize("batch", [8, 3, 1])
def test_presplit_embedding_with_poisoned_workspace(batch):
    """genie_embed_window_split leaves the split rows AND the message-mask row `mm` in the workspace; `mm` exists once per
    P-sized slot copy, and the window that consumes it runs under a slot chosen later (push_window / the pipelined forward).
    Sparse picks (most product nodes have an all-zero Mask row, a different set in every window), a workspace filled with
    NaN bit patterns beforehand, batches of 8 / 3 / 1 windows: a stage 2 that reads an `mm` copy this window never wrote
    yields NaN or a wrong Bipartite sum."""
    from genie_amd import apply
    from oracle import embed_oracle as E
    from oracle import genie_oracle as O
    S, G = 18, 90
    geom = synthetic.Geometry(S, G, L=400e3, n_query=12, seed=81)
    rng = np.random.default_rng(82)
    n = 260
    P = np.stack([np.sort(rng.uniform(5000.0, 5090.0, n)), rng.integers(0, S, n).astype(np.float64), np.ones(n), np.ones(n),
                  rng.integers(0, 2, n).astype(np.float64)], axis=1)
    trv = geom.travel_times().astype(np.float32)
    c = Case("tiny_6x40")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                             torch.from_numpy(geom.edge_attr()).to(DEV), torch.from_numpy(geom.locs).float().to(DEV),
                             torch.from_numpy(geom.x_grid).float().to(DEV))
    net._hip.ws.fill_(255)                       # every float of the workspace = NaN
    max_t = float(np.ceil(trv.max() + 1.0))
    sig = 0.5                                    # narrow kernel: |residual| > 1.5 s -> Mask 0
    tsteps = apply.window_schedule(P[:, 0], max_t, t_win=6.0, step_size="half")[0]
    times = tsteps[(tsteps > 4990.0) & (tsteps < 5080.0)][:19]
    Out_2, used = apply.apply_windows_device(net, geom, P, trv, step_size="half", max_t=max_t, kernel_sig_t=sig, dt_embed=0.1,
                                             times=times, tail_batch=batch)
    assert len(used) >= 12
    tsteps, offsets, step, n_overlap, dt_win = apply.window_schedule(P[:, 0], max_t, t_win=6.0, step_size="half")
    tsteps_abs = np.arange(tsteps.min() - 3.0, tsteps.max() + 3.0 + dt_win, dt_win)
    A = np.stack([np.tile(np.arange(S), G), np.repeat(np.arange(G), S)], axis=0)
    sta_nbr = graph.neighbour_table(geom.A_sta_sta, S)
    src_nbr = graph.neighbour_table(geom.A_src_src, G)
    want = torch.zeros(Out_2.shape)
    zero_rows = []
    for t0 in used:
        Slice, Mask = E.extract_input_from_data(P, float(t0), np.arange(S), S, trv, A, max_t, sig, 0.1)
        zero_rows.append(float((Mask.max(1) == 0).mean()))
        _, x = O.forward_fixed_source_structured(c.weights, torch.from_numpy(Slice), torch.from_numpy(Mask), sta_nbr, src_nbr,
                                                 torch.from_numpy(geom.edge_attr()), torch.from_numpy(geom.A_src_src),
                                                 torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(),
                                                 torch.from_numpy(offsets.reshape(-1, 1)).float(), S, G)
        cols, keep = apply.window_columns(tsteps_abs, t0, offsets, True)
        want[:, cols] += x[:, keep, 0] / 2.0
    assert 0.2 < np.mean(zero_rows) < 0.98 and np.std(zero_rows) > 0       # the message mask matters and differs by window
    assert torch.isfinite(Out_2).all()
    assert max_abs(Out_2.cpu(), want) <= 1e-5


def test_config5_stream_200_stations_one_second_stride_matches_oracle_chain():
    """BASELINE config 5 in miniature: 200 stations, 72 consecutive windows at 1 s stride (window starts that do NOT sit on
    the 0.75 s output axis: every one is snapped as process_continuous_days.py:766 does), device embedding + forward + `Out_2`
    stacking with tails batched 8 windows at a time, against the oracle chain embed_oracle.extract_input_from_data ->
    genie_oracle.forward_fixed_source_structured -> the reference's stacking statement (numpy fancy `+=`, :797-805)."""
    from genie_amd import apply
    from oracle import embed_oracle as E
    from oracle import genie_oracle as O
    S, G = 200, 150
    geom = synthetic.Geometry(S, G, L=300e3, n_query=40, seed=91)
    rng = np.random.default_rng(92)
    n = 6000
    P = np.stack([np.sort(rng.uniform(20000.0, 20200.0, n)), rng.integers(0, S, n).astype(np.float64), np.ones(n), np.ones(n),
                  rng.integers(0, 2, n).astype(np.float64)], axis=1)
    trv = geom.travel_times().astype(np.float32)
    max_t = float(np.ceil(trv.max() + 1.0))
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src),
                             torch.from_numpy(geom.edge_attr()).to(DEV), torch.from_numpy(geom.locs).float().to(DEV),
                             torch.from_numpy(geom.x_grid).float().to(DEV))
    times = 20010.3 + 1.0 * np.arange(72)                                     # 1 s stride
    sig = 1.0
    tsteps, offsets, step, n_overlap, dt_win = apply.window_schedule(P[:, 0], max_t, t_win=6.0, step_size="half")
    tsteps_abs = np.arange(tsteps.min() - 3.0, tsteps.max() + 3.0 + dt_win, dt_win)
    Out_2, used = apply.apply_windows_device(net, geom, P, trv, tsteps_abs=tsteps_abs, step_size="half", max_t=max_t,
                                             kernel_sig_t=sig, dt_embed=0.1, times=times, tail_batch=8)
    assert len(used) == 72
    A = np.stack([np.tile(np.arange(S), G), np.repeat(np.arange(G), S)], axis=0)
    sta_nbr = graph.neighbour_table(geom.A_sta_sta, S)
    src_nbr = graph.neighbour_table(geom.A_src_src, G)
    want = np.zeros(tuple(Out_2.shape))
    tq = torch.from_numpy(offsets.reshape(-1, 1)).float()
    for t0 in used:
        Slice, Mask = E.extract_input_from_data(P, float(t0), np.arange(S), S, trv, A, max_t, sig, 0.1)
        _, x = O.forward_fixed_source_structured(c.weights, torch.from_numpy(Slice), torch.from_numpy(Mask), sta_nbr, src_nbr,
                                                 torch.from_numpy(geom.edge_attr()), torch.from_numpy(geom.A_src_src),
                                                 torch.from_numpy(geom.x_grid).float(), torch.from_numpy(geom.x_query).float(), tq, S, G)
        i0 = int(np.abs(tsteps_abs - t0).argmin())                                                       # :766
        ip = np.abs(tsteps_abs.reshape(-1, 1) - (tsteps_abs[i0] + offsets).reshape(1, -1)).argmin(0)     # :797
        want[:, ip[0:-1]] += x[:, 0:-1, 0].numpy() / n_overlap                                           # :802-803
    assert float(np.abs(want).max()) > 0.02
    assert max_abs(Out_2.cpu(), torch.from_numpy(want)) <= 1e-5


def test_pipelined_forward_is_bitwise_equal_to_plain_forward():
    """Two-stream window pipeline (G-sized tail of window i overlaps stage 1/2 of window i+1, double-buffered scratch):
    every window's (y, x) must be bit-identical to the single-stream forward_fixed_source."""
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    net.set_adjacencies_base(c.A_sta_sta, c.A_src_src, c.edge_attr.to(DEV), c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    rng = np.random.default_rng(3)
    wins = []
    for k in range(6):
        S_ = torch.from_numpy(rng.random((c.S * c.G, 4)).astype(np.float32)) * (torch.rand(c.S * c.G, 1) < 0.3)
        wins.append((S_.to(DEV), (S_ > 0.01).float().to(DEV)))
    fixed = (None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
    with torch.no_grad():
        plain = [net.forward_fixed_source(s_, m_, *fixed) for s_, m_ in wins]
        torch.cuda.synchronize()
        piped = [net.forward_fixed_source_pipelined(s_, m_, *fixed) for s_, m_ in wins]
        torch.cuda.synchronize()
    for (y0, x0), (y1, x1, ev) in zip(plain, piped):
        assert torch.equal(y0, y1) and torch.equal(x0, x1)


@pytest.mark.parametrize("batch", [16, 8, 1, 3])
def test_batched_windows_are_bitwise_equal_to_plain_forward(batch):
    """push_window / flush_windows (stage 1 / 2 per window, ONE G-sized tail per batch of windows through genie_tail_batched):
    every window's (y, x) bit-identical to the single-stream forward_fixed_source; 19 windows = full batches in flight on
    rotating slot groups and alternating side streams plus a partial batch; a plain forward in between (after wait_tails)."""
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    net.set_adjacencies_base(c.A_sta_sta, c.A_src_src, c.edge_attr.to(DEV), c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    rng = np.random.default_rng(5)
    wins = []
    for k in range(19):
        S_ = torch.from_numpy(rng.random((c.S * c.G, 4)).astype(np.float32)) * (torch.rand(c.S * c.G, 1) < 0.3)
        wins.append((S_.to(DEV), (S_ > 0.01).float().to(DEV)))
    xg, xq, tq = c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV)
    fixed = (None, None, None, c.locs.float().to(DEV), xg, xq, tq)
    with torch.no_grad():
        plain = [net.forward_fixed_source(s_, m_, *fixed) for s_, m_ in wins]
        torch.cuda.synchronize()
        net.window_batch = batch
        assert net.window_batch == batch
        got = []
        for k, (s_, m_) in enumerate(wins):
            if net.push_window(s_, m_) == net.window_batch or k == len(wins) - 1:
                y, x, ev = net.flush_windows(xg, xq, tq)
                got.append((y, x, ev))
            if k == 9:
                net._hip.wait_tails()                                 # a plain forward uses slot 0's scratch: join the tails first
                mid = net.forward_fixed_source(*wins[3], *fixed)
        torch.cuda.synchronize()
    assert [g[0].shape[0] for g in got] == [batch] * (19 // batch) + ([19 % batch] if 19 % batch else [])
    ys, xs = torch.cat([g[0] for g in got]), torch.cat([g[1] for g in got])
    for k, (y0, x0) in enumerate(plain):
        assert torch.equal(y0, ys[k]) and torch.equal(x0, xs[k]), k
    assert torch.equal(mid[0], plain[3][0]) and torch.equal(mid[1], plain[3][1])
    with pytest.raises(RuntimeError):
        net.flush_windows(xg, xq, tq)


@pytest.mark.parametrize("G", [10000, 50000])
def test_window_pipelines_are_bitwise_equal_at_the_baseline_source_counts(G):
    """The grid-wide mean of SpatialAggregation's global term is summed over `vg` virtual blocks that depend on the source-node
    count only (SaArgs.vg), not on the launch grid: at 10 000 / 50 000 source nodes (configs 2 / 4) the tail of a plain call runs
    157 / 512 workgroups, a batch of 16 windows 32 per window, and every output bit must still be the same (round 4: it was not --
    3.7e-9 on y, 1.5e-8 on x -- while the fixtures' 500 source nodes fit one partition in every form). Also: switching from batched
    tails to `forward_fixed_source_pipelined` on one object."""
    S = 16
    geom = synthetic.Geometry(S, G, L=300e3, n_query=500, seed=5)
    win = synthetic.make_window(geom, 4000, seed=6)
    torch.manual_seed(0)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV).eval()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), t(geom.edge_attr()), t(geom.locs), t(geom.x_grid))
    dS, dM, xg, xq, tq = t(win["Slice"]), t(win["Mask"]), t(geom.x_grid), t(geom.x_query), t(geom.t_query)
    fixed = (None, None, None, t(geom.locs), xg, xq, tq)
    with torch.no_grad():
        y, x = net.forward_fixed_source(dS, dM, *fixed)
        for nb in (1, 3, 16):
            net.window_batch = nb
            for _ in range(nb):
                net.push_window(dS, dM)
            yb, xb, ev = net.flush_windows(xg, xq, tq)
            net._hip.wait_tails()
            torch.cuda.synchronize()
            for k in range(nb):
                assert torch.equal(yb[k], y) and torch.equal(xb[k], x), (nb, k)
        for _ in range(nb):
            net.push_window(dS, dM)
        net.flush_windows(xg, xq, tq)
        yp, xp, ev = net.forward_fixed_source_pipelined(dS, dM, *fixed)       # batched tails still in flight
        net._hip.wait_tails()
        torch.cuda.synchronize()
        assert torch.equal(yp, y) and torch.equal(xp, x)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["assoc_7x45", "assoc_20x60", "assoc_20x60_nonull", "assoc_edges_18x50", "assoc_abspos_18x50",
                                  "assoc_subgraph_14x50", "assoc_nophase_18x50", "assoc_edges_abspos_18x50", "assoc_subgraph_edges_14x50",
                                  "assoc_subgraph_abspos_14x50", "assoc_subgraph_edges_abspos_14x50"])
def test_forward_fixed_and_forward_four_outputs_match_reference(name):
    """module.py:963-997 / :908-939: (y, x, arv_p, arv_s) in HIP end to end (front, read-outs with their latents, association
    stages, LocalSliceLgCollapse, Arrivals) against the reference's own forward_fixed golden vectors: 7 stations (generic CSR
    kernels), 20 stations with 270 picks on one station and none on another (pipelined kernels, two LDS chunks of the arrival
    softmax), the same with no candidate source inside 2 eps (`edge_index[0].max()` is then a real pick, module.py:762-765), and
    the two other model definitions (fixtures from the reference imported with the flag set): `use_updated_model_definition`
    (DataAggregationAssociationPhaseEdges, module.py:407-480, :1128-1161) and `use_absolute_pos` (module.py:969-970, :987-988);
    `assoc_subgraph_14x50`: an irregular product graph (`use_subgraph: True`: product-level CSR forms of the association kernels,
    time-pointer tables from the reference's compute_time_embedding_vectors); `assoc_nophase_18x50`: `use_phase_types: False`
    (config.yaml:91; reference imported with the flag flipped). No PyTorch restatement may run in eval mode."""
    import os
    from tests.util import GOLDEN_DIR
    from oracle import genie_oracle as O
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    w = O.weights_from_npz(z)
    S, G = int(z["n_sta"]), int(z["n_grid"])
    t = lambda k, dt=torch.float32: torch.from_numpy(np.asarray(z[k])).to(dt).to(DEV)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition="edges" in name,
                                                use_absolute_pos="abspos" in name, use_phase_types="nophase" not in name)
    net.load_state_dict({k: v.clone() for k, v in w.items()}, strict=True)
    net.eval()
    if "pairs" in z.files:
        A_src_in_sta = torch.from_numpy(z["pairs"]).long()
        A_in_sta, A_in_src, A_src_in_prod = graph.subgraph_product_edges(z["A_sta_sta"], z["A_src_src"], z["pairs"])
    else:
        A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(z["A_sta_sta"], z["A_src_src"], S, G)
    ea = graph.GraphEdges(x=t("edge_attr"), edge_index=A_src_in_prod.to(DEV))
    ea_flip = graph.GraphEdges(x=t("edge_attr"), edge_index=A_src_in_prod.flip(0).contiguous().to(DEV))
    graphs = (A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea_flip, A_src_in_sta.to(DEV), t("A_src_src", torch.long),
              t("A_edges_p", torch.long), t("A_edges_s", torch.long), t("dt_partition"), t("tlatent"))
    tail = (t("tpick"), t("ipick", torch.long), t("phase_label"), t("locs"), t("x_grid"), t("x_query"), t("x_query_src"),
            t("t_query"), t("tq_sample"), t("trv_out_q"))
    def boom(*a, **k):
        raise AssertionError("a PyTorch restatement ran in eval mode")
    for m in (net.SpatialDirect, net.SpatialAttention, net.TemporalAttention, net.BipartiteGraphReadOutOperator,
              net.DataAggregationAssociationPhase, net.LocalSliceLgCollapseP, net.LocalSliceLgCollapseS, net.Arrivals):
        m.forward = boom
    with torch.no_grad():
        net.set_adjacencies(*graphs, t("locs"), t("x_grid"))
        out_fixed = net.forward_fixed(t("Slice"), t("Mask"), *tail)
        out_full = net(t("Slice"), t("Mask"), *graphs, *tail)
    for out in (out_fixed, out_full):
        assert max_abs(out[0].cpu(), torch.from_numpy(z["y"])) <= 1e-5
        assert max_abs(out[1].cpu(), torch.from_numpy(z["x"])) <= 1e-5
        assert out[2].shape == tuple(z["arv_p"].shape) and out[3].shape == tuple(z["arv_s"].shape)
        assert max_abs(out[2].cpu(), torch.from_numpy(z["arv_p"])) <= 1e-5
        assert max_abs(out[3].cpu(), torch.from_numpy(z["arv_s"])) <= 1e-5


@pytest.mark.parametrize("name", ["assoc_7x45", "assoc_20x60", "assoc_20x60_nonull", "assoc_edges_18x50", "assoc_abspos_18x50",
                                  "assoc_edges_abspos_18x50", "assoc_subgraph_14x50", "assoc_subgraph_edges_14x50",
                                  "assoc_subgraph_abspos_14x50", "assoc_subgraph_edges_abspos_14x50"])
def test_training_mode_four_output_forward_gradients_match_oracle_autograd(name):
    """a-8 / f-2: the training call convention `net(Slice, Mask, graphs..., picks...)` (train_GENIE_model.py:1786) in train()
    mode: all four outputs carry gradients and the gradients of every parameter equal the oracle's autograd ones. Every module
    runs in HIP in both directions -- the shared path with the source queries riding along (`_PathTrain`), the P-sized association
    heads (`_AssocTrain`), LocalSliceLgCollapse P / S (`_LslcTrain`) and the arrival head (`_ArrivalsTrain`; assoc_20x60 has a
    station with 266 picks = two softmax chunks, _nonull a case where no source keeps the null pick) -- and no PyTorch restatement
    may be called. The last two cases are the 4-output step of the two other model definitions (`use_updated_model_definition`,
    `use_absolute_pos`; fixtures from the reference imported with the flag set): their static per-station / per-source-node
    terms add weight gradients to DataAggregation AND to DataAggregationAssociationPhase (k_gr_sum_*, k_static_dw)."""
    import os
    from tests.util import GOLDEN_DIR
    from oracle import genie_oracle as O
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    w0 = O.weights_from_npz(z)
    S, G = int(z["n_sta"]), int(z["n_grid"])
    t = lambda k, dt=torch.float32, dev=DEV: torch.from_numpy(np.asarray(z[k])).to(dt).to(dev)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition="edges" in name,
                                                use_absolute_pos="abspos" in name)
    net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
    net.train()
    if "pairs" in z.files:      # `use_subgraph`: irregular product graph (PCSR forms of every P-sized pass, both directions)
        A_src_in_sta = torch.from_numpy(z["pairs"]).long()
        A_in_sta, A_in_src, A_src_in_prod = graph.subgraph_product_edges(z["A_sta_sta"], z["A_src_src"], z["pairs"])
    else:
        A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(z["A_sta_sta"], z["A_src_src"], S, G)
    ea = graph.GraphEdges(x=t("edge_attr"), edge_index=A_src_in_prod.to(DEV))
    ea_flip = graph.GraphEdges(x=t("edge_attr"), edge_index=A_src_in_prod.flip(0).contiguous().to(DEV))
    graphs = (A_in_sta.to(DEV), A_in_src.to(DEV), ea, ea_flip, A_src_in_sta.to(DEV), t("A_src_src", torch.long),
              t("A_edges_p", torch.long), t("A_edges_s", torch.long), t("dt_partition"), t("tlatent"))
    tail = (t("tpick"), t("ipick", torch.long), t("phase_label"), t("locs"), t("x_grid"), t("x_query"), t("x_query_src"),
            t("t_query"), t("tq_sample"), t("trv_out_q"))
    def boom(*a, **k):
        raise AssertionError("a PyTorch restatement of a HIP-trained module ran")
    for m in (net.SpatialDirect, net.SpatialAttention, net.TemporalAttention, net.BipartiteGraphReadOutOperator,
              net.DataAggregationAssociationPhase, net.LocalSliceLgCollapseP, net.LocalSliceLgCollapseS, net.Arrivals):
        m.forward = boom        # every module of the 4-output step: HIP in both directions, no PyTorch restatement
    outs = net(t("Slice"), t("Mask"), *graphs, *tail)
    assert all(o.requires_grad for o in outs)
    for o, k in zip(outs, ("y", "x", "arv_p", "arv_s")):
        assert max_abs(o.detach().cpu(), torch.from_numpy(z[k])) <= 1e-5, k
    g = torch.Generator().manual_seed(11)
    coef = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * c_.to(DEV)).sum() for o, c_ in zip(outs, coef)).backward()
    c = lambda k, dt=torch.float32: t(k, dt, "cpu")
    w = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
    okw = {}
    if "edges" in name:
        okw["pos_rel"] = (O.edge_pos_features(c("locs"), A_in_sta, A_src_in_sta[0]), O.edge_pos_features(c("x_grid"), A_in_src, A_src_in_sta[1]))
    if "abspos" in name:
        okw["abs_pos"] = (c("locs"), A_src_in_sta)
    ref = O.forward_fixed(w, c("Slice"), c("Mask"), A_in_sta, A_in_src, c("edge_attr"), A_src_in_prod, c("A_src_src", torch.long),
                          c("A_edges_p", torch.long), c("A_edges_s", torch.long), c("dt_partition"), c("tlatent"), c("tpick"),
                          c("ipick", torch.long), c("phase_label"), c("x_grid"), c("x_query"), c("x_query_src"), c("t_query"),
                          c("tq_sample"), c("trv_out_q"), S, **okw)
    sum((o * c_).sum() for o, c_ in zip(ref, coef)).backward()
    if "edges" in name:      # the static-term columns carry gradient in both P-sized modules
        for mod in ("DataAggregation", "DataAggregationAssociationPhase"):
            assert float(w[mod + ".l1_t1_2.weight"].grad[:, 60:64].abs().max()) > 0 and float(w[mod + ".l2_t2_2.weight"].grad[:, 90:94].abs().max()) > 0
    if "abspos" in name:
        assert float(w["DataAggregation.init_trns.weight"].grad[:, 4:10].abs().max()) > 0
        assert float(w["DataAggregationAssociationPhase.init_trns.weight"].grad[:, 15:21].abs().max()) > 0
    checked = 0
    gmax4 = max(float(v.grad.abs().max()) for v in w.values() if v.grad is not None)
    for k, p in net.named_parameters():
        if w[k].grad is None:
            continue
        assert p.grad is not None and tuple(p.grad.shape) == tuple(w[k].grad.shape), k
        # relative to the gradient's own scale (2e-4: f_arrival_query_2.bias of the _nonull case is a sum of cancelling terms, 1.3e-4)
        tol = 2e-4 * max(float(w[k].grad.abs().max()), 1e-3 * gmax4) + 1e-12
        assert max_abs(p.grad.cpu(), w[k].grad) <= tol, (k, max_abs(p.grad.cpu(), w[k].grad), tol)
        checked += 1
    assert checked >= 130


@pytest.mark.parametrize("S,G", [(37, 90), (200, 300)])
def test_station_processing_order_is_internal_only(S, G):
    """genie_set_station_order: with the stations processed in a different (space-filling-curve, or random) order every input and
    output keeps the caller's station order — h0 / h1 / c / wu / wv exports, x_latent, the Bipartite output and the path output
    agree with the run in the caller's order to fp32 summation error; a registered static edge_attr gives the same result as an
    unregistered one; bad permutations are rejected."""
    geom = synthetic.Geometry(S, G, L=150e3, n_query=10, seed=S)
    win = synthetic.make_window(geom, 30 * S, seed=S + 1)
    wd = {k: v.to(DEV) for k, v in Case("cfg1_20x500").weights.items()}
    Slice, Mask = torch.from_numpy(win["Slice"]).to(DEV), torch.from_numpy(win["Mask"]).to(DEV)
    ea, pos = torch.from_numpy(geom.edge_attr()).to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV)
    sta = engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S)
    src = engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G)

    def run(order, register):
        hp = engine.HipPath(S, G, sta, src, grid_order=engine.morton_order(geom.x_grid), device=DEV, sta_order=order)
        hp.set_weights(wd)
        if register:
            hp.set_static_edge_attr(ea)
        dbg = hp.da_stage1(Slice, Mask, debug=True)[2:]
        exports = [hp.export(k) for k in (0, 1, 2)]
        out, xl, bip = hp.path_fwd(Slice, Mask, ea, pos, True, True)
        return list(dbg) + exports + [out, xl, bip]

    base = run(None, False)
    rng = np.random.default_rng(S)
    for order, register in ((engine.morton_order(geom.locs), True), (engine.morton_order(geom.locs), False), (rng.permutation(S), True)):
        got = run(order, register)
        for a, b in zip(base, got):
            assert a.shape == b.shape and max_abs(a, b) <= 2e-6 * max(1.0, float(a.abs().max()))
    with pytest.raises(Exception):
        engine.HipPath(S, G, sta, src, device=DEV, sta_order=np.zeros(S, dtype=np.int64))


@pytest.mark.parametrize("nc,nq,k", [(500, 300, 10), (10000, 2500, 10), (37, 50, 10), (5, 7, 10), (3000, 3000, 15), (200, 200, 8),
                                     (5000, 1, 10), (10000, 8, 10), (4096, 64, 8), (4097, 3, 10)])      # (the last four: k_knn_b)
def test_device_knn_matches_exact_search(nc, nq, k):
    """genie_knn (module.py:282 / process_utils.py:718-719) against an exact fp64 search on the host: same neighbour sets in
    the same (nearest-first) order; with `exclude_self` the table equals genie_amd.graph.knn_graph (cKDTree, self removed)."""
    rng = np.random.default_rng(nc + nq)
    xc = np.stack([rng.uniform(0, 300e3, nc), rng.uniform(0, 300e3, nc), rng.uniform(-40e3, 2e3, nc)], axis=1).astype(np.float32)
    xq = np.stack([rng.uniform(0, 300e3, nq), rng.uniform(0, 300e3, nq), rng.uniform(-40e3, 2e3, nq)], axis=1).astype(np.float32)
    if nc >= 4096:                                                              # ties: duplicated context points, a query on one of them
        xc[nc // 2: nc // 2 + 12] = xc[100:112]
        xq[0] = xc[105]
    got = engine.knn_device(torch.from_numpy(xc).to(DEV), torch.from_numpy(xq).to(DEV), k).cpu().numpy()
    kk = min(k, nc)
    d = ((xq.astype(np.float64)[:, None, :] - xc.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    want = np.argsort(d, axis=1, kind="stable")[:, :kk]
    assert got.shape == (nq, kk) and np.array_equal(got, want)
    if nc == nq:       # base graph of a point set with itself
        tab, edges = engine.knn_graph_device(torch.from_numpy(xc).to(DEV), k)
        ref = graph.knn_graph(xc.astype(np.float64) / 1000.0, k)
        assert np.array_equal(edges.cpu().numpy(), ref)
        assert np.array_equal(tab.cpu().numpy().reshape(-1), ref[0])


@pytest.mark.parametrize("nc,nq,k,self_", [(3000, 20000, 10, False), (5, 16400, 10, False), (2049, 17000, 8, False), (16500, 16500, 10, True)])
def test_device_knn_lane_per_query_kernel(nc, nq, k, self_):
    """Query sets of >= 16 384 points take k_knn_t (one lane per query, fp32 guard band in front of the exact fp64 decision; the refine
    pass's 112 000-point clouds, process_continuous_days.py:929): same table as the exact host search, nearest first, ties by index --
    the context holds DUPLICATED points (equal distances) and a cluster the queries sit in (the guard band's job), fewer context
    points than k, a context one point past an LDS tile, and the base graph of a set with itself (`exclude_self`)."""
    rng = np.random.default_rng(nc + nq)
    xc = np.stack([rng.uniform(0, 300e3, nc), rng.uniform(0, 300e3, nc), rng.uniform(-40e3, 2e3, nc)], axis=1).astype(np.float32)
    if nc >= 100:
        xc[nc // 2: nc // 2 + 20] = xc[10:30]                                  # duplicates: ties in distance, decided by index
        xc[60:90] = xc[59] + rng.uniform(-40.0, 40.0, (30, 3)).astype(np.float32)  # a tight cluster
    if self_:
        xq = xc
    else:
        xq = np.stack([rng.uniform(0, 300e3, nq), rng.uniform(0, 300e3, nq), rng.uniform(-40e3, 2e3, nq)], axis=1).astype(np.float32)
        if nc >= 100:
            xq[:4000] = xc[59] + rng.uniform(-15e3, 15e3, (4000, 3)).astype(np.float32)   # a cloud around the cluster
            xq[4000:4040] = xc[10:50]                                                     # queries ON context points (distance 0, tied)
    got = engine.knn_device(torch.from_numpy(xc).to(DEV), torch.from_numpy(xq).to(DEV), k, exclude_self=self_).cpu().numpy()
    kk = min(k, nc - (1 if self_ else 0))
    assert got.shape == (xq.shape[0], kk)
    c64 = xc.astype(np.float64)
    for a in range(0, xq.shape[0], 2000):
        q = xq[a:a + 2000].astype(np.float64)
        d = ((q[:, None, :] - c64[None, :, :]) ** 2).sum(-1)
        if self_:
            d[np.arange(q.shape[0]), np.arange(a, a + q.shape[0])] = np.inf
        want = np.argsort(d, axis=1, kind="stable")[:, :kk]
        assert np.array_equal(got[a:a + 2000], want), a


@pytest.mark.parametrize("nc,nq,k", [(4096, 64, 10), (4096, 65, 10), (4095, 64, 10), (3000, 16383, 10), (3000, 16384, 8), (5000, 33, 8)])
def test_device_knn_kernel_dispatch_boundaries(nc, nq, k):
    """genie_knn picks one of three kernels by the sizes (a workgroup per query up to 64 queries against >= 4096 points, a wave per query,
    a lane per query from 16 384 queries): the same exact table on either side of every boundary."""
    rng = np.random.default_rng(nc * 7 + nq)
    xc = np.stack([rng.uniform(0, 100e3, nc), rng.uniform(0, 100e3, nc), rng.uniform(-40e3, 2e3, nc)], axis=1).astype(np.float32)
    xq = np.stack([rng.uniform(0, 100e3, nq), rng.uniform(0, 100e3, nq), rng.uniform(-40e3, 2e3, nq)], axis=1).astype(np.float32)
    xc[7] = xc[3]                                                              # a tie
    xq[0] = xc[3]
    got = engine.knn_device(torch.from_numpy(xc).to(DEV), torch.from_numpy(xq).to(DEV), k).cpu().numpy()
    c64 = xc.astype(np.float64)
    for a in range(0, nq, 4000):
        q = xq[a:a + 4000].astype(np.float64)
        d = ((q[:, None, :] - c64[None, :, :]) ** 2).sum(-1)
        assert np.array_equal(got[a:a + 4000], np.argsort(d, axis=1, kind="stable")[:, :k]), a


def test_set_adjacencies_from_positions_equals_host_built_graphs():
    """Graph setup on the device (genie_knn -> device CSR) gives the same forward as the host-built base graphs."""
    c = Case("cfg1_20x500")
    fixed = (None, None, None, c.locs.float().to(DEV), c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV))
    outs = []
    for mode in ("host", "device"):
        net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
        net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
        net.eval()
        if mode == "host":
            net.set_adjacencies_base(c.A_sta_sta, c.A_src_src, c.edge_attr.to(DEV), c.locs.float().to(DEV), c.x_grid.float().to(DEV))
        else:
            A_sta, A_src = net.set_adjacencies_from_positions(c.locs.float().to(DEV), c.x_grid.float().to(DEV), c.edge_attr.to(DEV))
            assert torch.equal(A_sta.cpu(), c.A_sta_sta) and torch.equal(A_src.cpu(), c.A_src_src)
        with torch.no_grad():
            outs.append(net.forward_fixed_source(c.Slice.to(DEV), c.Mask.to(DEV), *fixed))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("S,G", [(7, 45), (17, 33), (40, 300), (200, 120)])
def test_association_heads_hip_match_oracle(S, G):
    """genie_assoc_fwd (BipartiteGraphReadOutOperator + DataAggregationAssociationPhase, module.py:333-403) against the oracle's
    restatement (pinned to the reference's forward_fixed by tests/golden/assoc_7x45.npz, tests/test_assoc_cpu.py): ragged tiles
    and small graphs (generic stage-2 kernel), 40 / 200 stations (station processing order, k_stage2_fast without its Bipartite
    half); mask1 both 0 and 1; 1e-5 x max(1, max|ref|)."""
    from oracle import genie_oracle as O
    geom = synthetic.Geometry(S, G, L=150e3, n_query=10, seed=S + G)
    win = synthetic.make_window(geom, 25 * S, seed=S)
    z = np.load(__import__("os").path.join(__import__("tests.util", fromlist=["GOLDEN_DIR"]).GOLDEN_DIR, "assoc_7x45.npz"))
    w = O.weights_from_npz(z)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in w.items()}, strict=True)
    net.eval()
    ea = torch.from_numpy(geom.edge_attr())
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), ea.to(DEV),
                             torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV))
    hp = net._hip
    hp.sync_weights(net._path_params)
    assert hp.assoc_ready
    Slice, Mask = torch.from_numpy(win["Slice"]), torch.from_numpy(win["Mask"])
    rng = np.random.default_rng(S)
    y_latent = torch.from_numpy(rng.normal(0, 1, (G, 30)).astype(np.float32))
    mask_src = torch.from_numpy((rng.random((G, 1)) < 0.6).astype(np.float32))
    with torch.no_grad():
        _, x_latent, _ = hp.path_fwd(Slice.to(DEV), Mask.to(DEV), ea.to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV), True, False)
        got = hp.assoc_fwd(y_latent.to(DEV), mask_src.to(DEV), x_latent, Mask.to(DEV), ea.to(DEV))
        got2 = hp.assoc_fwd(y_latent.to(DEV), mask_src.to(DEV), x_latent, Mask.to(DEV), ea.to(DEV))
        A_in_sta, A_in_src, _, _ = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
        s, m1 = O.bipartite_read_out(w, y_latent, ea, mask_src, S)
        want = O.data_aggregation_association(w, s, x_latent.cpu(), m1, Mask, A_in_sta, A_in_src)
    assert torch.equal(got, got2)
    assert got.shape == want.shape and float(want.abs().max()) > 0.05
    assert max_abs(got.cpu(), want) <= rel_tol(want), max_abs(got.cpu(), want)


@pytest.mark.parametrize("S,G", [(200, 300), (40, 90), (100, 64)])
def test_stage2_work_maps_are_bitwise_equal(S, G):
    """k_stage2_ord with its large-station-count work map (blocks of 4 source nodes per workgroup, one node per wave; the default
    from 1024 stations up) against the interleaved map: x_latent, Bipartite output and the association pass (stage 2 without its
    Bipartite half) bit for bit -- per-tile results do not depend on which wave computes them."""
    geom = synthetic.Geometry(S, G, L=200e3, n_query=10, seed=S)
    win = synthetic.make_window(geom, 20 * S, seed=S + 1)
    wd = {k: v.to(DEV) for k, v in Case("cfg1_20x500").weights.items()}
    Slice, Mask = torch.from_numpy(win["Slice"]).to(DEV), torch.from_numpy(win["Mask"]).to(DEV)
    ea, pos = torch.from_numpy(geom.edge_attr()).to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV)
    rng = np.random.default_rng(1)
    yl = torch.from_numpy(rng.normal(0, 1, (G, 30)).astype(np.float32)).to(DEV)
    ms = torch.from_numpy((rng.random(G) < 0.5).astype(np.float32)).to(DEV)

    def run(blocks_of_four):
        hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                            engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G), grid_order=engine.sfc_order(geom.x_grid),
                            device=DEV, sta_order=engine.sfc_order(geom.locs))
        hp.set_stage2_workmap(blocks_of_four)          # (genie_set_stage2_workmap: the library reads no environment variable)
        hp.set_weights(wd)
        out, xl, bip = hp.path_fwd(Slice, Mask, ea, pos, True, True)
        return out, xl, bip, hp.assoc_fwd(yl, ms, xl, Mask, ea)

    base = run(False)
    got = run(True)
    for a, b in zip(base, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name,T", [("cfg1_20x500", 9), ("odd_33x257", 1), ("o1_20x500", 10), ("tiny_6x40", 4)])
def test_tail_kernels_other_time_query_counts_vs_oracle(name, T):
    """The G- / Q-sized tail (Bipartite read-out, SpatialAggregation x3, both read-out heads) with 1 / 4 / 9 / 10 time queries, G and
    Q that are not multiples of 16: y / x against the oracle's read-out heads on the HIP x_spatial (1e-5 absolute, 2e-6 of the
    scale), every output bitwise equal when run twice."""
    from oracle import genie_oracle as O
    c = Case(name)
    hp = make_engine(c)
    Slice, Mask, ea = c.Slice.to(DEV), c.Mask.to(DEV), c.edge_attr.to(DEV)
    xg, xq = c.x_grid.float().to(DEV), c.x_query.float().to(DEV)
    tq = (torch.arange(T, dtype=torch.float32) * 1.7 - 3.0).to(DEV)
    from genie_amd.module import knn_query_edges
    edges = knn_query_edges(xg, xq, 10)
    table = edges[0].view(xq.shape[0], -1).to(torch.int32).contiguous()

    def run():
        out, _, bip = hp.path_fwd(Slice, Mask, ea, xg, want_x_latent=True, want_bip=True)
        y = hp.readout_grid(out, tq)
        x = hp.readout_query(out, xg, xq, table, tq)
        torch.cuda.synchronize()
        return [t.clone() for t in (bip, out, y, x)]

    got = run()
    again = run()
    for a, b, k in zip(got, again, ("bip", "sa3", "y", "x")):
        assert torch.equal(a, b), k
        assert torch.isfinite(a).all(), k
    w = c.weights
    sa3 = got[1].cpu()
    tqc = tq.cpu().view(-1, 1)
    y_o = O.temporal_attention(w, O.spatial_direct(w, sa3), tqc)
    x_o = O.temporal_attention(w, O.spatial_attention(w, sa3, c.x_query.float(), c.x_grid.float(), edge_index=edges.cpu()), tqc)
    for a, r, k in ((got[2].cpu(), y_o, "y"), (got[3].cpu(), x_o, "x")):
        assert a.shape == r.shape, k
        assert max_abs(a, r) <= 2e-6 * max(5.0, float(r.abs().max())), (k, max_abs(a, r))


def test_device_subgraph_builder_matches_reference_builder():
    """f-4: the irregular product graph of `use_subgraph` built on the device (genie_knn + subgraph_pairs_device +
    genie_subgraph_csr_count / _fill) against the output of the reference's own extract_inputs_adjacencies_subgraph on the same
    geometry (tests/golden/subgraph_builder_14x50.npz, oracle/make_golden.py --subgraph): the same product nodes in the same
    order, the same edge sets; then against the host builder on a larger random geometry (exact CSR equality), and the module
    entry point runs the path on the device-built graph and agrees with the host-built one bit for bit."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subgraph_builder_14x50.npz"))
    geom = synthetic.Geometry(14, 50, L=80e3, n_query=21, seed=71)
    locs, xg = torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV)
    pairs = engine.subgraph_pairs_device(torch.from_numpy(geom.locs).to(DEV), torch.from_numpy(geom.x_grid).to(DEV),
                                         max_deg_offset=0.15, k_nearest_pairs=6, scale_deg=110e3)
    assert np.array_equal(pairs.cpu().numpy(), z["A_src_in_sta"])
    sta_tab, A_sta = engine.knn_graph_device(locs, 8)
    src_tab, A_src = engine.knn_graph_device(xg, 15)
    as_set = lambda a: set(map(tuple, np.asarray(a).T.tolist()))
    assert as_set(A_sta.cpu().numpy()) == as_set(z["A_sta_sta"]) and as_set(A_src.cpu().numpy()) == as_set(z["A_src_src"])
    sub = engine.subgraph_csr_device(pairs, 50, engine.csr_from_table(sta_tab), engine.csr_from_table(src_tab))

    def edges_of(csr):
        rp, col = csr[0].cpu().numpy(), csr[1].cpu().numpy()
        tgt = np.repeat(np.arange(rp.size - 1), np.diff(rp))
        return np.stack((col, tgt))
    assert as_set(edges_of(sub["sta_csr"])) == as_set(z["A_prod_sta_sta"])
    assert as_set(edges_of(sub["src_csr"])) == as_set(z["A_prod_src_src"])
    assert np.array_equal(np.diff(sub["seg_rowptr"].cpu().numpy()), np.bincount(z["A_src_in_sta"][1], minlength=50))

    # larger random geometry: exact equality (edge order included) with the host builder, then the path on both graphs
    S, G = 40, 300
    geom = synthetic.Geometry(S, G, L=150e3, n_query=30, seed=5)
    locs, xg = torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV)
    c = Case("odd_33x257")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    A_sta, A_src, pairs = net.set_adjacencies_subgraph_from_positions(locs, xg, None, k_sta_edges=8, k_spc_edges=15,
                                                                      max_deg_offset=0.3, k_nearest_pairs=12)
    N = pairs.shape[1]
    assert S * 12 <= N < S * G
    A1, A2, Ap = graph.subgraph_product_edges(A_sta.cpu().numpy(), A_src.cpu().numpy(), pairs.cpu().numpy())
    sub = engine.subgraph_csr_device(pairs, G, engine.csr_from_edges(A_sta, S), engine.csr_from_edges(A_src, G))
    for got, host in ((sub["sta_csr"], engine.csr_from_edges(A1, N)), (sub["src_csr"], engine.csr_from_edges(A2, N))):
        assert torch.equal(got[0].cpu(), host[0]) and torch.equal(got[1].cpu(), host[1])
    rng = np.random.default_rng(11)
    Slice = torch.from_numpy(rng.random((N, 4)).astype(np.float32)).to(DEV)
    Mask = torch.from_numpy((rng.random((N, 4)) < 0.4).astype(np.float32)).to(DEV)
    xq, tq = torch.from_numpy(geom.x_query).float().to(DEV), torch.from_numpy(geom.t_query).float().to(DEV)
    with torch.no_grad():
        y1, x1 = net.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
    ea_x = net._edge_attr.clone()
    net2 = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net2.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net2.eval()
    ea = graph.GraphEdges(x=ea_x, edge_index=Ap.to(DEV))
    net2.set_adjacencies(A1.to(DEV), A2.to(DEV), ea, ea, pairs, A_src, None, None, None, None, locs, xg)
    with torch.no_grad():
        y2, x2 = net2.forward_fixed_source(Slice, Mask, None, None, None, locs, xg, xq, tq)
    assert torch.equal(y1, y2) and torch.equal(x1, x2) and torch.isfinite(y1).all()


@pytest.mark.parametrize("S,G,n_picks", [(7, 45, 23), (40, 300, 1000), (21, 64, 1)])
def test_local_slice_collapse_hip_matches_module(S, G, n_picks):
    """f-2: LocalSliceLgCollapse P / S (module.py:610-659) in HIP (genie_lslc_fwd) against the PyTorch restatement in
    genie_amd/module.py (itself pinned to the reference by tests/test_assoc_cpu.py): time-pointer tables of
    graph.time_pointers, picks scattered over the stations, some with no product node inside 2 eps (empty mean = 0)."""
    rng = np.random.default_rng(S * 1000 + n_picks)
    geom = synthetic.Geometry(S, G, L=150e3, n_query=10, seed=S)
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), torch.from_numpy(geom.edge_attr()).to(DEV),
                             torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV))
    d = np.linalg.norm(geom.x_grid[:, None, :] - geom.locs[None, :, :], axis=2)              # [G, S]
    trv = np.stack((d / 6000.0, d / 3500.0), axis=2).astype(np.float32)                        # P / S travel times
    ep, es, dtp = graph.time_pointers(trv, max_t=float(trv.max()), dt=0.6, k=10, win=6.0)
    tlatent = torch.from_numpy(trv.reshape(G * S, 2)).to(DEV)
    ipick = torch.from_numpy(rng.integers(0, S, n_picks)).to(DEV)
    tpick = torch.from_numpy(rng.uniform(-5.0, float(trv.max()) + 5.0, n_picks).astype(np.float32)).to(DEV)
    tpick[0] = float(dtp[0]) + 0.01                                                            # first time bin
    phase = torch.from_numpy(rng.integers(0, 2, (n_picks, 1)).astype(np.float32)).to(DEV)
    s = torch.from_numpy(rng.normal(0, 1, (G * S, 30)).astype(np.float32)).to(DEV)
    dtp_t = torch.from_numpy(dtp.astype(np.float32)).to(DEV)
    net._hip.sync_weights(net._path_params)
    from tests import restatements as R
    R.attach(net)            # the PyTorch restatement of the head (test infrastructure) on this model's parameters
    for head, (mod, tab, col) in enumerate(((net.LocalSliceLgCollapseP, ep, 0), (net.LocalSliceLgCollapseS, es, 1))):
        tab_t = torch.from_numpy(tab).to(DEV)
        with torch.no_grad():
            ref = mod(tab_t, dtp_t, tpick, ipick, phase, s, tlatent[:, col].reshape(-1, 1))
        got = net._hip.lslc_fwd(head, s, tab_t.to(torch.int32), dtp_t, tpick, ipick.to(torch.int32), phase, tlatent, col, mod.eps)
        assert got.shape == ref.shape and torch.isfinite(got).all()
        assert max_abs(got.cpu(), ref.cpu()) <= 2e-6 * max(1.0, float(ref.abs().max())), head


def test_local_slice_collapse_reports_a_pick_outside_the_table_at_the_next_call():
    """module.py:635-640 indexes the time-pointer table with floor((tpick - t0) / dt) and ipick; an index outside the table is a
    device-side assertion there (reported at the next synchronisation). Here the kernel clamps the index and sets a bit in host-mapped
    memory (genie_index_flags); the host raises IndexError at its next call on the context, without a synchronisation per call (the
    bounds used to be read back in every lslc_fwd). Covered: a time beyond the partition, a station beyond the table, and that the
    report is made once."""
    S, G, n = 7, 45, 12
    rng = np.random.default_rng(3)
    geom = synthetic.Geometry(S, G, L=150e3, n_query=10, seed=S)
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), torch.from_numpy(geom.edge_attr()).to(DEV),
                             torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV))
    d = np.linalg.norm(geom.x_grid[:, None, :] - geom.locs[None, :, :], axis=2)
    trv = np.stack((d / 6000.0, d / 3500.0), axis=2).astype(np.float32)
    ep, es, dtp = graph.time_pointers(trv, max_t=float(trv.max()), dt=0.6, k=10, win=6.0)
    tlatent = torch.from_numpy(trv.reshape(G * S, 2)).to(DEV)
    s = torch.from_numpy(rng.normal(0, 1, (G * S, 30)).astype(np.float32)).to(DEV)
    dtp_t = torch.from_numpy(dtp.astype(np.float32)).to(DEV)
    tab = torch.from_numpy(ep).to(DEV).to(torch.int32)
    phase = torch.zeros((n, 1), device=DEV)
    net._hip.sync_weights(net._path_params)
    hp, eps = net._hip, net.LocalSliceLgCollapseP.eps
    ip_ok = torch.from_numpy(rng.integers(0, S, n)).to(DEV).to(torch.int32)
    tp_ok = torch.from_numpy(rng.uniform(0.0, float(trv.max()), n).astype(np.float32)).to(DEV)
    call = lambda tp, ip: hp.lslc_fwd(0, s, tab, dtp_t, tp, ip, phase, tlatent, 0, eps)
    good = call(tp_ok, ip_ok)
    for bad in ("time", "station", "negative time"):
        tp, ip = tp_ok.clone(), ip_ok.clone()
        if bad == "time":
            tp[5] = float(dtp[-1]) + 100.0
        elif bad == "negative time":
            tp[0] = float(dtp[0]) - 50.0
        else:
            ip[n - 1] = S + 3
        out = call(tp, ip)                                   # not refused here: the index is clamped on the device ...
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        with pytest.raises(IndexError, match="outside the time-pointer table"):
            call(tp_ok, ip_ok)                               # ... and reported by the next call
        again = call(tp_ok, ip_ok)                           # once
        torch.cuda.synchronize()
        assert torch.equal(again, good)
    hp.check_index_flags()
    # the same report for the Arrivals head's station indices (bit 1, genie_index_check): clamped for the call, raised by the next one
    n_src = 2
    trv_src = torch.from_numpy(np.stack((d[:n_src] / 6000.0, d[:n_src] / 3500.0), axis=2).astype(np.float32)).to(DEV)
    stime, emb = torch.zeros(n_src, device=DEV), torch.zeros((n_src, 30), device=DEV)
    arr = lambda ip: hp.arrivals_fwd(stime, emb, trv_src, good, good, tp_ok, ip, phase, net.Arrivals.eps)
    ref = arr(ip_ok.long())
    ip_bad = ip_ok.long().clone()
    ip_bad[3] = S
    out = arr(ip_bad)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    with pytest.raises(IndexError, match="station index"):
        arr(ip_ok.long())
    assert torch.equal(arr(ip_ok.long()), ref)


def test_device_side_verdicts_of_a_contexts_last_call_are_not_lost():
    """The flag word is per context and read at the NEXT entry point: for the last call on a context nothing would read it. Covered here:
    `check_index_flags(synchronize=True)` reports a call that is still in flight; replacing the model's context (`set_adjacencies*`, what
    `forward` does per training sample) raises what the old context's last call left behind; `GridLeg.check()` (the per-day loops call it
    after their final copy to the host); dropping a context with unread verdicts warns."""
    import gc
    import warnings
    from genie_amd import apply
    S, G, n = 7, 45, 12
    rng = np.random.default_rng(3)
    geom = synthetic.Geometry(S, G, L=150e3, n_query=10, seed=S)
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    adj = (torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), torch.from_numpy(geom.edge_attr()).to(DEV),
           torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV))
    net.set_adjacencies_base(*adj)
    d = np.linalg.norm(geom.x_grid[:, None, :] - geom.locs[None, :, :], axis=2)
    trv = np.stack((d / 6000.0, d / 3500.0), axis=2).astype(np.float32)
    ep, es, dtp = graph.time_pointers(trv, max_t=float(trv.max()), dt=0.6, k=10, win=6.0)
    tlatent = torch.from_numpy(trv.reshape(G * S, 2)).to(DEV)
    s = torch.from_numpy(rng.normal(0, 1, (G * S, 30)).astype(np.float32)).to(DEV)
    dtp_t = torch.from_numpy(dtp.astype(np.float32)).to(DEV)
    tab = torch.from_numpy(ep).to(DEV).to(torch.int32)
    phase = torch.zeros((n, 1), device=DEV)
    ip = torch.from_numpy(rng.integers(0, S, n)).to(DEV).to(torch.int32)
    tp_bad = torch.from_numpy(rng.uniform(0.0, float(trv.max()), n).astype(np.float32)).to(DEV)
    tp_bad[4] = float(dtp[-1]) + 100.0

    def bad_call():
        net._hip.sync_weights(net._path_params)
        return net._hip.lslc_fwd(0, s, tab, dtp_t, tp_bad, ip, phase, tlatent, 0, net.LocalSliceLgCollapseP.eps)

    bad_call()                                                       # (1) the explicit, synchronising check
    with pytest.raises(IndexError, match="outside the time-pointer table"):
        net._hip.check_index_flags(synchronize=True)
    net._hip.check_index_flags(synchronize=True)                     # reported once
    bad_call()                                                       # (2) the context is replaced right after its last call
    with pytest.raises(IndexError, match="outside the time-pointer table"):
        net.set_adjacencies_base(*adj)
    net.set_adjacencies_base(*adj)                                   # the model is usable again
    leg = apply.GridLeg(net, geom.x_grid, trv)                       # (3) the per-day loops' final check
    bad_call()
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        leg.check()
    leg.check()
    with pytest.raises(ValueError, match="ind_use"):                 # a table over another station set is refused (process_utils.py:599)
        apply.GridLeg(net, geom.x_grid, np.concatenate((trv, trv[:, :2]), axis=1))
    leg2 = apply.GridLeg(net, geom.x_grid, np.concatenate((trv[:, :2], trv), axis=1), ind_use=np.arange(2, S + 2))
    assert torch.equal(leg2.trv, leg.trv)
    bad_call()                                                       # (4) dropped with unread verdicts: a warning, not silence
    torch.cuda.synchronize()
    hp, net._hip = net._hip, None
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        del hp, leg, leg2
        gc.collect()
    assert any("unread device-side verdicts" in str(w.message) for w in rec)


@pytest.mark.parametrize("S,n_src,n_picks", [(7, 4, 23), (40, 9, 1500), (12, 1, 1), (30, 3, 600), (3, 2, 1300)])
def test_arrivals_head_hip_matches_module(S, n_src, n_picks):
    """f-2: StationSourceAttentionMergedPhases (`Arrivals`, module.py:662-775) in HIP (genie_arrivals_fwd) against the PyTorch
    restatement in genie_amd/module.py (pinned to the reference's forward_fixed by tests/test_assoc_cpu.py): picks clustered
    around the theoretical arrivals of the sources (so that the 2-eps windows hold several picks per station, some stations
    none), picks outside every window (targets without edges), one source whose null pick is filtered (|stime| >= 2 eps);
    3 stations x 1300 picks: several LDS chunks and two target blocks per station (streaming softmax)."""
    rng = np.random.default_rng(S * 100 + n_picks)
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()}, strict=True)
    net.eval()
    geom = synthetic.Geometry(S, 40, L=100e3, n_query=5, seed=S)
    net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), torch.from_numpy(geom.edge_attr()).to(DEV),
                             torch.from_numpy(geom.locs).float().to(DEV), torch.from_numpy(geom.x_grid).float().to(DEV))
    net._hip.sync_weights(net._path_params)
    eps = net.Arrivals.eps
    trv = rng.uniform(5.0, 60.0, (n_src, S, 1)).astype(np.float32)
    trv = np.concatenate((trv, trv * 1.7), axis=2)
    stime = rng.uniform(-10.0, 10.0, n_src).astype(np.float32)
    if n_src > 2:
        stime[-1] = 2.5 * eps                                      # this source's null pick is filtered
    ipick = rng.integers(0, S, n_picks)
    src_of = rng.integers(0, n_src, n_picks)
    tpick = (trv[src_of, ipick, rng.integers(0, 2, n_picks)] + stime[src_of] + rng.normal(0, 0.7 * eps, n_picks)).astype(np.float32)
    far = rng.random(n_picks) < 0.1
    tpick[far] += 500.0                                            # picks no source explains
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    arv_p = t(rng.normal(0, 1, (n_picks, 15)).astype(np.float32))
    arv_s = t(rng.normal(0, 1, (n_picks, 15)).astype(np.float32))
    phase = t(rng.integers(0, 2, (n_picks, 1)).astype(np.float32))
    x_src = t(rng.normal(0, 1, (n_src, 30)).astype(np.float32))
    from tests import restatements as R
    R.attach(net)
    with torch.no_grad():
        ref = net.Arrivals(n_src, t(stime), x_src, t(trv), arv_p, arv_s, t(tpick), t(ipick), phase)
    got = net._hip.arrivals_fwd(t(stime), x_src, t(trv), arv_p, arv_s, t(tpick), t(ipick), phase, eps)
    assert got is not None and got.shape == ref.shape and torch.isfinite(got).all()
    assert max_abs(got.cpu(), ref.cpu()) <= 5e-6 * max(1.0, float(ref.abs().max()))
    # no source inside 2 eps of the origin time: `edge_index[0].max()` (module.py:762-763) is then a real pick (k_arr_e0max) and the
    # self / null links follow it, as the reference computes them (reference-pinned by tests/golden/assoc_20x60_nonull.npz)
    if n_picks < 20:              # (`remainder(e1, e0max)` with e0max = 0 is undefined on the device: the reference's own code breaks there)
        return
    st2 = stime.copy()
    st2[:] = np.where(st2 >= 0, 2.0 * eps + 1.0 + st2, -2.0 * eps - 1.0 + st2)
    tp2 = (tpick + st2[src_of] - stime[src_of]).astype(np.float32)
    try:
        with torch.no_grad():
            ref2 = net.Arrivals(n_src, t(st2), x_src, t(trv), arv_p, arv_s, t(tp2), t(ipick), phase)
    except RuntimeError:          # no edge survives, or only pick 0 does (`remainder(e1, 0)`): the reference's own code fails there
        return
    got2 = net._hip.arrivals_fwd(t(st2), x_src, t(trv), arv_p, arv_s, t(tp2), t(ipick), phase, eps)
    assert max_abs(got2.cpu(), ref2.cpu()) <= 5e-6 * max(1.0, float(ref2.abs().max()))


def test_bench_line_contract_on_the_gpu():
    """`python bench.py` (short run, the optional extras off): ONE JSON line with the fields the driver reads, the headline
    workload named, a roofline object priced on algorithmic bytes against the HBM peak and per-kernel HIP-event times."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "16", "--warmup", "8", "--settle", "16", "--no-cpu-baseline",
                        "--no-cfg4-one-gpu", "--no-live-traffic", "--no-train-step", "--no-stream", "--no-day-loops"], cwd=repo,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    # the record echoes the command (`warmup` = the requested warm-up calls); the clock-settle windows are their own key, and the figure
    # of the command as given on an idle GPU sits next to the settled one that `value` is quoted on
    assert d["n_gpus"] == 1 and d["steps"] == 16 and d["warmup"] == 8 and d["settle_windows"] == 16 and d["unit"] == "picks/s"
    assert d["settled_ms_per_step"] == d["ms_per_step"] and 0.1 < d["cold_ms_per_step"] < 10.0
    assert d["settled_long_run"]["steps"] == 1000 and 0.1 < d["settled_long_run"]["ms_per_step"] < 5.0     # 16 calls last under 0.1 s
    assert d["dtype"] == "f32" and "ONE literal call forward_fixed_source" in d["config"]["workload"]
    # the headline IS the literal call; the window pipeline is an extra and not slower than it
    assert d["drop_in_call_ms"] == d["ms_per_step"] and 0.1 < d["pipelined_windows_ms"] <= d["ms_per_step"] * 1.05
    assert 0.0 < d["roofline"]["fused_bytes_frac"] < d["roofline"]["frac"]
    assert d["config"]["workload"].startswith("cfg2_200x10k") and d["config"]["n_picks"] == 50000
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(d["value"] - 50000.0 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]
    assert abs(rf["achieved"] - 3.07216 / d["ms_per_step"] * 1e3) <= 2e-3 * rf["achieved"]          # B_alg = 3.072 GB per window
    assert 0.1 < d["ms_per_step"] < 5.0 and set(rf["kernels"]) == {"k_stage1", "k_stage2"}


def test_drop_in_accepts_the_tensor_forms_pytorch_callers_pass():
    """The reference's methods are plain PyTorch and take whatever tensors a caller has: float64 / non-contiguous Slice, a uint8 or
    bool Mask, positions in float64, a strided view of the query table. The drop-in normalises them (host side, before the C ABI)
    and returns the same bits as for contiguous fp32 inputs."""
    c = Case("cfg1_20x500")
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    net.set_adjacencies_base(c.A_sta_sta, c.A_src_src, c.edge_attr.to(DEV), c.locs.float().to(DEV), c.x_grid.float().to(DEV))
    Sl, Mk = c.Slice.to(DEV), c.Mask.to(DEV)
    locs, xg, xq, tq = c.locs.float().to(DEV), c.x_grid.float().to(DEV), c.x_query.float().to(DEV), c.t_query.float().to(DEV)
    with torch.no_grad():
        y0, x0 = net.forward_fixed_source(Sl, Mk, None, None, None, locs, xg, xq, tq)
        wide = torch.zeros((Sl.shape[0], 8), device=DEV)
        wide[:, ::2] = Sl
        xq_wide = torch.zeros((xq.shape[0], 6), device=DEV)
        xq_wide[:, :3] = xq
        forms = [
            (Sl.double(), Mk, locs, xg, xq, tq),                                   # float64 features
            (wide[:, ::2], Mk, locs, xg, xq, tq),                                  # strided view
            (Sl, Mk.to(torch.uint8), locs, xg, xq, tq),                            # uint8 mask
            (Sl, Mk.bool(), locs, xg, xq, tq),                                     # bool mask
            (Sl, Mk, locs.double(), xg.double(), xq_wide[:, :3], tq.double()),     # float64 positions, strided queries
            (Sl, Mk, locs, xg, xq, tq.reshape(-1)),                                # flat t_query
        ]
        for k, (s_, m_, l_, g_, q_, t_) in enumerate(forms):
            y, x = net.forward_fixed_source(s_, m_, None, None, None, l_, g_, q_, t_)
            assert torch.equal(y, y0) and torch.equal(x, x0), k


def test_product_edge_lists_are_verified_on_the_device():
    """Training call convention (train_GENIE_model.py:1722-1786): new product edge lists per sample, resident on the GPU. genie_product_check
    verifies their Cartesian structure in one pass and the base tables are cut on the device: same tables as the host path, and every kind
    of non-Cartesian list is refused (a single altered entry anywhere, a shifted block, a base block that leaves its node range)."""
    geom = synthetic.Geometry(23, 310, L=100e3, n_query=5, seed=8)
    S, G = 23, 310
    A1, A2, _, _ = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    want = graph.base_tables_from_product(A1, A2, S, G)                       # host path (torch.equal of materialised copies)
    got = graph.base_tables_from_product(A1.to(DEV), A2.to(DEV), S, G)        # device path
    assert got[0].is_cuda and got[0].dtype == torch.int32
    assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1])
    rng = np.random.default_rng(3)
    for which, A in ((0, A1), (1, A2)):
        for row in (0, 1):
            for pos in (0, int(rng.integers(1, A.shape[1] - 1)), A.shape[1] - 1):
                B = A.clone()
                B[row, pos] += 1 if which == 0 else S
                args = (B.to(DEV), A2.to(DEV)) if which == 0 else (A1.to(DEV), B.to(DEV))
                with pytest.raises(ValueError):
                    graph.base_tables_from_product(*args, S, G)
    B = A2.clone()
    B[:, : A2.shape[1] // S] += 1                                             # first block is not station 0
    with pytest.raises(ValueError):
        graph.base_tables_from_product(A1.to(DEV), B.to(DEV), S, G)
    # an irregular (use_subgraph) pair of lists of a compatible length is refused too, and set_adjacencies then takes the CSR path
    with pytest.raises(ValueError):
        graph.base_tables_from_product(A1.flip(1).contiguous().to(DEV), A2.to(DEV), S, G)


def test_forward_defers_the_structure_checks_and_recovers_when_they_fail():
    """`forward` (graphs per call, train_GENIE_model.py:1786) builds its context from the first blocks of the GPU product edge lists at
    once and reads the verdicts of the structure checks after issuing its kernels (one read-back where set_adjacencies made two before any
    kernel could be queued). Covered: the verdicts are consumed; the same graph with A_src written in another edge ORDER fails the
    literal comparison, is rebuilt with the checks up front (CSR comparison) and gives bit-identical outputs; a product list with one
    altered entry is not Cartesian: the call falls back to the general builder with the checks up front, as it always did."""
    import os
    from tests.util import GOLDEN_DIR
    from oracle import genie_oracle as O
    z = np.load(os.path.join(GOLDEN_DIR, "assoc_7x45.npz"))
    S, G = int(z["n_sta"]), int(z["n_grid"])
    t = lambda k, dt=torch.float32: torch.from_numpy(np.asarray(z[k])).to(dt).to(DEV)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in O.weights_from_npz(z).items()}, strict=True)
    net.eval()
    A1, A2, A3, A4 = graph.cartesian_product_edges(z["A_sta_sta"], z["A_src_src"], S, G)
    ea = graph.GraphEdges(x=t("edge_attr"), edge_index=A3.to(DEV))
    tabs = (t("A_edges_p", torch.long), t("A_edges_s", torch.long), t("dt_partition"), t("tlatent"))
    tail = (t("tpick"), t("ipick", torch.long), t("phase_label"), t("locs"), t("x_grid"), t("x_query"), t("x_query_src"), t("t_query"),
            t("tq_sample"), t("trv_out_q"))
    A_src = t("A_src_src", torch.long)
    calls = []
    plain = net.set_adjacencies
    net.set_adjacencies = lambda *a, **k: (calls.append(bool(k.get("_defer_checks"))), plain(*a, **k))[1]
    with torch.no_grad():
        ref = net(t("Slice"), t("Mask"), A1.to(DEV), A2.to(DEV), ea, ea, A4.to(DEV), A_src, *tabs, *tail)
        assert calls == [True] and net._pending_checks is None
        assert max_abs(ref[0].cpu(), torch.from_numpy(z["y"])) <= 1e-5 and max_abs(ref[2].cpu(), torch.from_numpy(z["arv_p"])) <= 1e-5
        perm = torch.sort(-A_src[1], stable=True)[1]          # in-edges still grouped by centre and in their order, centres descending
        got = net(t("Slice"), t("Mask"), A1.to(DEV), A2.to(DEV), ea, ea, A4.to(DEV), A_src[:, perm].contiguous(), *tabs, *tail)
        assert calls == [True, True, False] and net._pending_checks is None
        assert all(torch.equal(a, b) for a, b in zip(ref, got))
        bad = A1.clone()
        bad[0, bad.shape[1] // 2] = (bad[0, bad.shape[1] // 2] + 1) % (S * G)
        try:           # (the general builder takes the altered list as the irregular graph it describes, or refuses it: as before deferral)
            other = net(t("Slice"), t("Mask"), bad.to(DEV), A2.to(DEV), ea, ea, A4.to(DEV), A_src, *tabs, *tail)
            assert other[0].shape == ref[0].shape and torch.isfinite(other[0]).all()
        except (ValueError, RuntimeError):
            pass
        # a model whose lists failed the literal checks once stops deferring (every later sample would pay a discarded forward and a
        # second build): the altered list goes straight to the checked path
        assert net._defer_failed and calls == [True, True, False, False] and getattr(net, "_pending_checks", None) is None
        again = net(t("Slice"), t("Mask"), A1.to(DEV), A2.to(DEV), ea, ea, A4.to(DEV), A_src, *tabs, *tail)      # and the model still works
        assert all(torch.equal(a, b) for a, b in zip(ref, again)) and calls[-1] is False
    # a fresh model with an altered list as its FIRST sample: deferred build, verdict bad, rebuilt with the checks up front
    net2 = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net2.load_state_dict({k: v.clone() for k, v in O.weights_from_npz(z).items()}, strict=True)
    net2.eval()
    calls2 = []
    plain2 = net2.set_adjacencies
    net2.set_adjacencies = lambda *a, **k: (calls2.append(bool(k.get("_defer_checks"))), plain2(*a, **k))[1]
    with torch.no_grad():
        try:
            net2(t("Slice"), t("Mask"), bad.to(DEV), A2.to(DEV), ea, ea, A4.to(DEV), A_src, *tabs, *tail)
        except (ValueError, RuntimeError):
            pass
        assert calls2 == [True, False] and net2._defer_failed and getattr(net2, "_pending_checks", None) is None


@pytest.mark.parametrize("n", [3, 200, 10000, 50000])
def test_device_space_filling_curve_order_equals_the_host_one(n):
    rng = np.random.default_rng(n)
    x = np.stack([rng.uniform(0, 3e5, n), rng.uniform(0, 3e5, n), rng.uniform(-4e4, 2e3, n)], 1).astype(np.float32)
    x[n // 2] = x[0]                                                          # equal codes: the stable sort keeps index order
    a = engine.sfc_order(x)
    b = engine.sfc_order(torch.from_numpy(x).to(DEV))
    assert b.is_cuda and b.dtype == torch.int32 and np.array_equal(a, b.cpu().numpy())


def test_contexts_rebuilt_per_sample_reuse_pooled_memory_and_stay_correct():
    """The library's device-memory pool and the size-class allocations hand a rebuilt context the blocks of the one it replaces: a
    sequence of graphs of changing size through ONE model object (forward with fresh product edge lists per sample, as the reference's
    training loop calls it) must give, for every sample, exactly what a freshly created model gives on that sample alone."""
    c = Case("tiny_6x40")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)

    def sample(i):
        geom = synthetic.Geometry(18 + (i * 7) % 5, 90 + 10 * (i % 3), L=80e3, n_query=40, seed=300 + i)
        win = synthetic.make_window(geom, 300, seed=400 + i)
        A1, A2, A3, A4 = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, geom.n_sta, geom.n_grid, device=DEV)
        ea = graph.GraphEdges(x=t(geom.edge_attr()), edge_index=A3)
        return geom, win, (A1, A2, ea, ea, A4, torch.from_numpy(geom.A_src_src).to(DEV))

    def run(net, s):
        geom, win, graphs = s
        net.set_adjacencies(*graphs, None, None, None, None, t(geom.locs), t(geom.x_grid))
        with torch.no_grad():
            return net.forward_fixed_source(t(win["Slice"]), t(win["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query),
                                            t(geom.t_query))

    samples = [sample(i) for i in range(7)]
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
    net.eval()
    outs = [tuple(o.clone() for o in run(net, s)) for s in samples + samples[::-1]]
    for k, s in enumerate(samples + samples[::-1]):
        fresh = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
        fresh.load_state_dict({k_: v.clone() for k_, v in c.weights.items()})
        fresh.eval()
        y, x = run(fresh, s)
        assert torch.equal(y, outs[k][0]) and torch.equal(x, outs[k][1]), k
        del fresh


def test_edge_attr_and_node_tables_as_callables_equal_the_tensor_forms():
    """`set_adjacencies_base(..., edge_attr=callable)` and `node_rows(callable)` evaluate the caller's function for the source nodes the
    model holds, block by block (what keeps config 4's 1.2 GB `edge_attr` and 0.8 GB travel-time table off every rank of a sharded job);
    on an unsharded model they must give exactly what the full tensors give."""
    S, G = 12, 70
    geom = synthetic.Geometry(S, G, L=80e3, n_query=9, seed=15)
    win = synthetic.make_window(geom, 150, seed=16)
    c = Case("tiny_6x40")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)

    def make(edge_attr):
        net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
        net.load_state_dict({k: v.clone() for k, v in c.weights.items()})
        net.eval()
        net.set_adjacencies_base(torch.from_numpy(geom.A_sta_sta), torch.from_numpy(geom.A_src_src), edge_attr, t(geom.locs), t(geom.x_grid))
        return net

    a, b = make(t(geom.edge_attr())), make(geom.edge_attr)
    assert torch.equal(a._edge_attr, b._edge_attr)
    with torch.no_grad():
        ya, xa = a.forward_fixed_source(t(win["Slice"]), t(win["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query), t(geom.t_query))
        yb, xb = b.forward_fixed_source(t(win["Slice"]), t(win["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query), t(geom.t_query))
    assert torch.equal(ya, yb) and torch.equal(xa, xb)
    trv = geom.travel_times().astype(np.float32)
    r1, r2, r3 = a.node_rows(trv, 2), a.node_rows(geom.travel_times), a.node_rows(torch.from_numpy(trv).reshape(-1, 2))
    assert tuple(r1.shape) == (S * G, 2) and torch.equal(r1, r2) and torch.equal(r1, r3)
    with pytest.raises(ValueError):
        a.node_rows(trv, 3)
