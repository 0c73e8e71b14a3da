"""GPU: randomized parity sweeps (seven tests: 2-output and 4-output forward + every gradient, window pipelines, device embedding,
apply loop on irregular graphs, one object across changing graphs, Adam steps). Every case draws a geometry (stations, source nodes, queries, picks), a model definition
(`use_updated_model_definition`, `use_absolute_pos`, both, neither), a product graph (Cartesian, or the irregular one of
`use_subgraph: True`, process_utils.py:744-849), a stage precision and perturbed weights, and compares the drop-in class with the
oracle's literal edge-list formulation (oracle/genie_oracle.py, pinned to the reference by tests/golden): `(y, x)` of
`forward_fixed_source` (module.py:999-1020) in eval mode to 1e-5 absolute, train mode equal to eval mode, and every parameter
gradient of a random cotangent to 2e-4 of that gradient's own scale (a mismatch is accepted only where the oracle's own gradient is
discontinuous: a pre-activation within rounding of a PReLU kink). `GENIE_FUZZ_CASES` / `GENIE_FUZZ_SEED` widen the sweep
(e.g. GENIE_FUZZ_SEED=20000 GENIE_FUZZ_CASES=160: 832 cases, 7.5 min on an MI355X; GENIE_FUZZ_BIG=1: 200 stations x 1000 / 2500 source nodes); the default is a handful of fixed seeds."""
import os

import numpy as np
import pytest
import torch

from genie_amd import engine, graph, module, synthetic
from tests.util import Case, max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_CASES = int(os.environ.get("GENIE_FUZZ_CASES", "16"))      # (seed 14: both model options, 31 stations -- the case of the round-4 fix)
SEED0 = int(os.environ.get("GENIE_FUZZ_SEED", "0"))
_WEIGHT_SOURCE = {(False, False): "cfg1_20x500", (True, False): "edges_12x60", (False, True): "abspos_12x60", (True, True): "edges_abspos_12x60"}


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.choice([3, 4, 7, 15, 16, 17, 31, 33, 48, 64, 70]))
    G = int(rng.choice([10, 16, 17, 40, 63, 129, 257, 400]))       # (the read-out takes the 10 nearest source nodes of a query)
    cfg = dict(S=S, G=G, Q=int(rng.integers(1, 80)), n_picks=int(rng.choice([0, 1, 50, 400, 2000])),
               edges=bool(rng.integers(2)), abspos=bool(rng.integers(2)), subgraph=bool(rng.integers(2)),
               stage="f32" if rng.integers(3) == 0 else "default", L=float(rng.choice([40e3, 150e3, 400e3])), seed=seed)
    # (drawn after the fields above so that the cases of earlier sweeps keep their seeds)
    cfg["ragged"] = bool(rng.integers(4) == 0)      # base graphs with edges dropped at random: non-uniform degrees, empty neighbourhoods
    if rng.integers(5) == 0:                        # the production station counts (k_stage2_h2u, work maps), fewer source nodes
        cfg["S"], cfg["G"] = int(rng.choice([100, 200, 333])), int(rng.choice([16, 40, 129]))
    if os.environ.get("GENIE_FUZZ_BIG"):            # production-like sizes (the CPU oracle then takes seconds per case)
        cfg["S"], cfg["G"], cfg["n_picks"] = 200, int(rng.choice([1000, 2500])), int(rng.choice([2000, 20000]))
    return cfg


def _geometry(cfg):
    geom = synthetic.Geometry(cfg["S"], cfg["G"], L=cfg["L"], n_query=cfg["Q"], seed=300 + cfg["seed"])
    if cfg.get("ragged"):
        rng = np.random.default_rng(1300 + cfg["seed"])
        geom.A_sta_sta = np.ascontiguousarray(geom.A_sta_sta[:, rng.random(geom.A_sta_sta.shape[1]) >= 0.25])
        geom.A_src_src = np.ascontiguousarray(geom.A_src_src[:, rng.random(geom.A_src_src.shape[1]) >= 0.25])
    return geom


def _weights(cfg):
    w = {k: v.clone() for k, v in Case(_WEIGHT_SOURCE[(cfg["edges"], cfg["abspos"])]).weights.items()}
    g = torch.Generator().manual_seed(77 + cfg["seed"])
    for k, v in w.items():
        if v.numel() == 1:
            v.fill_(float(0.05 + 0.45 * torch.rand(1, generator=g)))               # PReLU slopes away from the 0.25 default
        else:
            v.mul_(0.8 + 0.4 * torch.rand(v.shape, generator=g))
    return w


@pytest.mark.parametrize("i", range(N_CASES))
def test_random_case_forward_and_gradients_match_the_oracle(i, monkeypatch):
    from oracle import genie_oracle as O
    cfg = _draw(SEED0 + i)
    print(cfg)
    if cfg["stage"] == "f32":
        monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    S, G, Q = cfg["S"], cfg["G"], cfg["Q"]
    geom = _geometry(cfg)
    win = synthetic.make_window(geom, cfg["n_picks"], seed=500 + cfg["seed"])
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    t = lambda a: c(a).to(DEV)
    Slice, Mask, ea = win["Slice"], win["Mask"], geom.edge_attr()
    if cfg["subgraph"]:
        rng = np.random.default_rng(900 + cfg["seed"])
        d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)          # [G, S]
        keep = np.zeros(d.shape, dtype=bool)
        keep[np.arange(G)[:, None], np.argsort(d, axis=1)[:, :int(rng.integers(1, min(6, S) + 1))]] = True
        keep |= rng.random(d.shape) < rng.choice([0.0, 0.1, 0.5])
        src_i, sta_i = np.nonzero(keep)
        pairs = np.stack((sta_i, src_i))
        rows = src_i * S + sta_i
        Slice, Mask, ea = Slice[rows], Mask[rows], ea.reshape(G, S, 3)[src_i, sta_i]
        A_in_sta, A_in_src, A_src_in_prod = graph.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
        A_src_in_sta = torch.from_numpy(pairs).long()
    else:
        A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    w0 = _weights(cfg)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition=cfg["edges"],
                                                use_absolute_pos=cfg["abspos"])
    net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
    gea = graph.GraphEdges(x=t(ea), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), gea, gea, A_src_in_sta.to(DEV), torch.from_numpy(geom.A_src_src).to(DEV),
                        None, None, None, None, t(geom.locs), t(geom.x_grid))
    args = (t(Slice), t(Mask), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query), t(geom.t_query))
    net.eval()
    with torch.no_grad():
        y_e, x_e = net.forward_fixed_source(*args)
    net.train()
    y, x = net.forward_fixed_source(*args)
    gen = torch.Generator().manual_seed(7 + cfg["seed"])
    ay, ax = torch.randn(y.shape, generator=gen), torch.randn(x.shape, generator=gen)
    (y * ay.to(DEV)).sum().add((x * ax.to(DEV)).sum()).backward()
    # ---- the oracle on the same edge lists
    def oracle(scale):
        w = {k: (v.clone() * scale).requires_grad_(True) for k, v in w0.items()}
        So, okw = c(Slice), {}
        if cfg["abspos"]:
            So = O.absolute_pos_inputs(So, c(geom.locs), c(geom.x_grid), A_src_in_sta)
        if cfg["edges"]:
            okw["pos_rel"] = (O.edge_pos_features(c(geom.locs), A_in_sta, A_src_in_sta[0]), O.edge_pos_features(c(geom.x_grid), A_in_src, A_src_in_sta[1]))
        yo, xo = O.forward_fixed_source(w, So, c(Mask), A_in_sta, A_in_src, c(ea), A_src_in_prod, torch.from_numpy(geom.A_src_src),
                                        c(geom.x_grid), c(geom.x_query), c(geom.t_query), **okw)
        ((yo * ay).sum() + (xo * ax).sum()).backward()
        return yo.detach(), xo.detach(), {k: v.grad for k, v in w.items()}

    yo, xo, gref = oracle(1.0)
    ey, ex = max_abs(y_e.cpu(), yo), max_abs(x_e.cpu(), xo)
    ty, tx = max_abs(y.detach().cpu(), yo), max_abs(x.detach().cpu(), xo)
    print("eval - oracle: y %.1e x %.1e; train - oracle: y %.1e x %.1e" % (ey, ex, ty, tx))
    assert ey <= 1e-5 and ex <= 1e-5, (cfg, ey, ex)
    assert ty <= 1e-5 and tx <= 1e-5, (cfg, ty, tx)
    gmax = max(float(v.abs().max()) for v in gref.values() if v is not None)
    checked, worst, nudged, excused = 0, 0.0, None, 0
    for k, p in net.named_parameters():
        ref = gref[k]
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, (cfg, k)
            continue
        assert p.grad is not None, (cfg, k)
        tol = 2e-4 * max(float(ref.abs().max()), 1e-3 * gmax) + 1e-12
        if k == "TemporalAttention.proj_2.bias":
            # this gradient IS the sum of the cotangent (d out / d bias = 1): ~1 400 terms of size 1 that cancel to ~1e-2 (sweep seed
            # 47118: 0.0285), so the two fp32 sums -- the oracle's and the kernel's, in different orders -- differ at the rounding
            # level of the TERMS, not of the result: a floor of 2e-8 of the sum of their magnitudes (a third of one fp32 ulp per term)
            tol = max(tol, 2e-8 * float(ay.abs().sum() + ax.abs().sum()))
        err = max_abs(p.grad.cpu(), ref)
        if err > tol:
            # A pre-activation within rounding of a PReLU kink makes the REFERENCE's gradient discontinuous there (seed 108: |z| = 5e-9
            # in SpatialAggregation3.fc1, one unit of one edge): accepted only if the oracle's own gradient of this parameter moves
            # by as much when every weight is scaled by 1 +- 1e-6.
            if nudged is None:
                nudged = [oracle(1.0 + 1e-6)[2], oracle(1.0 - 1e-6)[2]]
            jump = max(max_abs(g[k], ref) for g in nudged)
            print("kink check %s: error %.2e, oracle's own jump under a 1e-6 weight scaling %.2e" % (k, err, jump))
            assert jump >= 0.5 * err, (cfg, k, err, tol, jump)
            excused += 1
            continue
        worst = max(worst, err / tol)
        checked += 1
    assert checked + excused >= 80
    print("worst gradient error / tolerance %.3f over %d parameters" % (worst, checked))


@pytest.mark.parametrize("i", range(N_CASES))
def test_random_case_window_pipelines_are_bitwise_equal_to_the_plain_forward(i, monkeypatch):
    """The apply loop's two ways of overlapping windows -- `forward_fixed_source_pipelined` (tails on side streams) and
    `push_window` / `flush_windows` (one G-sized tail per batch of windows) -- on the drawn model definition / graph shape /
    precision: every window's (y, x) bit-identical to the single-stream `forward_fixed_source`."""
    cfg = _draw(9000 + SEED0 + i)
    print(cfg)
    if cfg["stage"] == "f32":
        monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    S, G = cfg["S"], cfg["G"]
    geom = _geometry(cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
    ea = geom.edge_attr()
    rows = None
    if cfg["subgraph"]:
        rng = np.random.default_rng(900 + cfg["seed"])
        d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
        keep = np.zeros(d.shape, dtype=bool)
        keep[np.arange(G)[:, None], np.argsort(d, axis=1)[:, :int(rng.integers(1, min(6, S) + 1))]] = True
        keep |= rng.random(d.shape) < rng.choice([0.0, 0.1, 0.5])
        src_i, sta_i = np.nonzero(keep)
        pairs = np.stack((sta_i, src_i))
        rows = src_i * S + sta_i
        ea = ea.reshape(G, S, 3)[src_i, sta_i]
        A_in_sta, A_in_src, A_src_in_prod = graph.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
        A_src_in_sta = torch.from_numpy(pairs).long()
    else:
        A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition=cfg["edges"],
                                                use_absolute_pos=cfg["abspos"])
    net.load_state_dict({k: v.clone() for k, v in _weights(cfg).items()}, strict=True)
    net.eval()
    gea = graph.GraphEdges(x=t(ea), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), gea, gea, A_src_in_sta.to(DEV), torch.from_numpy(geom.A_src_src).to(DEV),
                        None, None, None, None, t(geom.locs), t(geom.x_grid))
    wins = []
    for k in range(5):
        win = synthetic.make_window(geom, max(cfg["n_picks"], 20), seed=700 + 10 * cfg["seed"] + k)
        Sl, Mk = (win["Slice"], win["Mask"]) if rows is None else (win["Slice"][rows], win["Mask"][rows])
        wins.append((t(Sl), t(Mk)))
    xg, xq, tq = t(geom.x_grid), t(geom.x_query), t(geom.t_query)
    fixed = (None, None, None, t(geom.locs), xg, xq, tq)
    with torch.no_grad():
        plain = [net.forward_fixed_source(s_, m_, *fixed) for s_, m_ in wins]
        torch.cuda.synchronize()
        piped = [net.forward_fixed_source_pipelined(s_, m_, *fixed) for s_, m_ in wins]
        torch.cuda.synchronize()
        for k, ((y0, x0), (y1, x1, ev)) in enumerate(zip(plain, piped)):
            assert torch.equal(y0, y1) and torch.equal(x0, x1), (cfg, "pipelined", k)
        net.window_batch = 2           # (no wait_tails in between: the switch joins the pending tails itself)
        got = []
        for k, (s_, m_) in enumerate(wins):
            if net.push_window(s_, m_) == net.window_batch or k == len(wins) - 1:
                got.append(net.flush_windows(xg, xq, tq))
        torch.cuda.synchronize()
        # ... and back to the side-stream form with batched tails still in flight (both forms share the workspace slots)
        again = [net.forward_fixed_source_pipelined(s_, m_, *fixed) for s_, m_ in wins[:3]]
        net._hip.wait_tails()
        torch.cuda.synchronize()
    ys, xs = torch.cat([g[0] for g in got]), torch.cat([g[1] for g in got])
    for k, (y0, x0) in enumerate(plain):
        assert torch.equal(y0, ys[k]) and torch.equal(x0, xs[k]), (cfg, "batched", k)
    for k, (y1, x1, ev) in enumerate(again):
        assert torch.equal(plain[k][0], y1) and torch.equal(plain[k][1], x1), (cfg, "pipelined after batched", k)


def _subgraph_time_pointers(trv, pairs, max_t, dt, k, win):
    """Time-pointer tables of an irregular product graph for the sweep: per station and time step of `dt_partition` the k product
    nodes OF THAT STATION whose travel time is nearest (cycled when a station has fewer than k), as product-node ids -- the layout
    `LocalSliceLgCollapse` indexes (module.py:635-637). Any table of valid ids serves here: both sides consume the same one."""
    dtp = np.arange(-win, win + max_t + dt, dt)
    S = trv.shape[1]
    out = []
    for ph in range(2):
        tab = np.zeros((S, dtp.size, k), dtype=np.int64)
        for i in range(S):
            nodes = np.nonzero(pairs[0] == i)[0]
            tt = trv[pairs[1][nodes], i, ph]
            order = np.argsort(np.abs(tt[None, :] - dtp[:, None]), axis=1, kind="stable")
            tab[i] = nodes[np.take(order, np.arange(k) % nodes.size, axis=1)]
        out.append(tab.reshape(-1))
    return out[0], out[1], dtp.astype(np.float32)


@pytest.mark.parametrize("i", range(N_CASES))
def test_random_case_four_outputs_and_gradients_match_the_oracle(i, monkeypatch):
    """The reference's own step `mz(*input_tensors)` (train_GENIE_model.py:1786; module.py:908-939): the shared path plus
    BipartiteGraphReadOutOperator, DataAggregationAssociationPhase, LocalSliceLgCollapse P / S and StationSourceAttentionMergedPhases,
    eval and train mode, all four outputs and every parameter gradient against `oracle.forward_fixed`."""
    from oracle import genie_oracle as O
    from genie_amd import graph as Gm
    cfg = _draw(5000 + SEED0 + i)
    rng = np.random.default_rng(4000 + cfg["seed"])
    cfg["G"] = max(cfg["G"], 16)
    cfg["n_src"] = int(rng.integers(1, 7))
    cfg["n_picks"] = int(rng.choice([1, 30, 300, 1500]))
    print(cfg)
    if cfg["stage"] == "f32":
        monkeypatch.setattr(engine, "STAGE_PRECISION", "f32")
    S, G, Q = cfg["S"], cfg["G"], cfg["Q"]
    geom = _geometry(cfg)
    smp = synthetic.training_sample(geom, cfg["n_picks"], n_src=min(cfg["n_src"], G), seed=600 + cfg["seed"], window=0)
    Slice, Mask, ea, tlat = smp["Slice"], smp["Mask"], geom.edge_attr(), smp["tlatent"]
    A_edges_p, A_edges_s, dtp = smp["A_edges_p"], smp["A_edges_s"], smp["dt_partition"]
    if cfg["subgraph"]:
        d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)          # [G, S]
        keep = np.zeros(d.shape, dtype=bool)
        keep[np.arange(G)[:, None], np.argsort(d, axis=1)[:, :int(rng.integers(1, min(6, S) + 1))]] = True
        keep[np.argmin(d, axis=0), np.arange(S)] = True                                        # every station keeps a source node
        keep |= rng.random(d.shape) < rng.choice([0.0, 0.1, 0.5])
        src_i, sta_i = np.nonzero(keep)
        pairs = np.stack((sta_i, src_i))
        rows = src_i * S + sta_i
        Slice, Mask, ea, tlat = Slice[rows], Mask[rows], ea[rows], tlat[rows]
        A_in_sta, A_in_src, A_src_in_prod = Gm.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
        A_src_in_sta = torch.from_numpy(pairs).long()
        trv = geom.travel_times().astype(np.float32)
        A_edges_p, A_edges_s, dtp = _subgraph_time_pointers(trv, pairs, float(np.ceil(trv.max())), synthetic.KERNEL_SIG_T / 5.0, 10,
                                                            2.0 * synthetic.KERNEL_SIG_T)
    else:
        A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = Gm.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    c = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    t = lambda a, dt=torch.float32: c(a, dt).to(DEV)
    w0 = _weights(cfg)
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition=cfg["edges"],
                                                use_absolute_pos=cfg["abspos"])
    net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
    gea = graph.GraphEdges(x=t(ea), edge_index=A_src_in_prod.to(DEV))
    gea_flip = graph.GraphEdges(x=t(ea), edge_index=A_src_in_prod.flip(0).contiguous().to(DEV))
    graphs = (A_in_sta.to(DEV), A_in_src.to(DEV), gea, gea_flip, A_src_in_sta.to(DEV), t(geom.A_src_src, torch.long),
              t(A_edges_p, torch.long), t(A_edges_s, torch.long), t(dtp), t(tlat))
    tail = (t(smp["tpick"]), t(smp["ipick"], torch.long), t(smp["phase_label"]), t(geom.locs), t(geom.x_grid), t(geom.x_query),
            t(smp["x_query_src"]), t(geom.t_query), t(smp["tq_sample"]), t(smp["trv_out_q"]))
    net.eval()
    with torch.no_grad():
        out_e = net(t(Slice), t(Mask), *graphs, *tail)
        net.set_adjacencies(*graphs, t(geom.locs), t(geom.x_grid))                  # the cached-graph form (module.py:941-997)
        out_f = net.forward_fixed(t(Slice), t(Mask), *tail)
    assert all(torch.equal(a, b) for a, b in zip(out_e, out_f)), cfg
    net.train()
    outs = net(t(Slice), t(Mask), *graphs, *tail)
    gen = torch.Generator().manual_seed(11 + cfg["seed"])
    coef = [torch.randn(o.shape, generator=gen) for o in outs]
    sum((o * c_.to(DEV)).sum() for o, c_ in zip(outs, coef)).backward()

    def oracle(scale, dt=torch.float32):
        w = {k: (v.clone().to(dt) * scale).requires_grad_(True) for k, v in w0.items()}
        f = lambda a: c(a, dt)
        okw = {}
        if cfg["edges"]:
            okw["pos_rel"] = (O.edge_pos_features(f(geom.locs), A_in_sta, A_src_in_sta[0]), O.edge_pos_features(f(geom.x_grid), A_in_src, A_src_in_sta[1]))
        if cfg["abspos"]:
            okw["abs_pos"] = (f(geom.locs), A_src_in_sta)
        ref = O.forward_fixed(w, f(Slice), f(Mask), A_in_sta, A_in_src, f(ea), A_src_in_prod, c(geom.A_src_src, torch.long),
                              c(A_edges_p, torch.long), c(A_edges_s, torch.long), f(dtp), f(tlat), f(smp["tpick"]),
                              c(smp["ipick"], torch.long), f(smp["phase_label"]), f(geom.x_grid), f(geom.x_query), f(smp["x_query_src"]),
                              f(geom.t_query), f(smp["tq_sample"]), f(smp["trv_out_q"]), S, **okw)
        sum((o * c_.to(dt)).sum() for o, c_ in zip(ref, coef)).backward()
        return [o.detach() for o in ref], {k: v.grad for k, v in w.items()}

    try:
        ref, gref = oracle(1.0)
    except ValueError as e:        # no pick within the window of any source: the reference raises there too (np.hstack of an empty list, module.py:713)
        assert "at least one array" in str(e), e
        pytest.skip("window without a pick near any source: the reference's forward raises")
    errs = [(max_abs(a.cpu(), r), max_abs(b.detach().cpu(), r)) for a, b, r in zip(out_e, outs, ref)]
    print("eval / train - oracle (y, x, arv_p, arv_s):", " ".join("%.1e/%.1e" % e for e in errs))
    assert all(a.shape == r.shape for a, r in zip(out_e, ref))
    assert max(max(e) for e in errs) <= 1e-5, (cfg, errs)
    gmax = max(float(v.abs().max()) for v in gref.values() if v is not None)
    checked, worst, nudged, g64, excused = 0, 0.0, None, None, 0
    for k, p in net.named_parameters():
        ref_g = gref[k]
        if ref_g is None:
            continue
        assert p.grad is not None and tuple(p.grad.shape) == tuple(ref_g.shape), (cfg, k)
        tol = 2e-4 * max(float(ref_g.abs().max()), 1e-3 * gmax) + 1e-12
        err = max_abs(p.grad.cpu(), ref_g)
        if err > tol:
            # (1) the reference's own fp32 rounding? (sums over hundreds of picks per station in the arrival head): against the
            # oracle in fp64, no worse than 4 x the fp32 oracle. (2) a discontinuity of the reference itself (PReLU kinks, the
            # max-pooling of LocalSliceLgCollapse): see the 2-output sweep above.
            if g64 is None:
                g64 = oracle(1.0, torch.float64)[1]
            e_hip, e_o32 = max_abs(p.grad.cpu().double(), g64[k]), max_abs(ref_g.double(), g64[k])
            print("fp64 check %s: error %.2e (tolerance %.2e); against fp64: HIP %.2e, fp32 oracle %.2e" % (k, err, tol, e_hip, e_o32))
            if e_hip <= max(tol, 4.0 * e_o32):
                excused += 1
                continue
            if nudged is None:
                nudged = [oracle(1.0 + 1e-6)[1], oracle(1.0 - 1e-6)[1]]
            jump = max(max_abs(g[k], ref_g) for g in nudged)
            print("kink check %s: error %.2e, oracle's own jump under a 1e-6 weight scaling %.2e" % (k, err, jump))
            assert jump >= 0.5 * err, (cfg, k, err, tol, jump)
            excused += 1
            continue
        worst = max(worst, err / tol)
        checked += 1
    assert checked + excused >= 120
    print("worst gradient error / tolerance %.3f over %d parameters" % (worst, checked))


@pytest.mark.parametrize("i", range(N_CASES))
def test_random_case_device_embedding_matches_the_oracle(i):
    """Pick -> Slice / Mask embedding on the device (genie_embed_window; f-1) against oracle/embed_oracle.py (pinned to the reference's
    `extract_input_from_data`, process_utils.py:460-642, by tests/golden/embed_*): random station / source counts, pick counts
    (none, one, thousands), window start, kernel width, time step, `use_sign_input`. Slice to 1e-6 (float cast of a float64 exp),
    Mask exact."""
    from oracle import embed_oracle as E
    rng = np.random.default_rng(7000 + SEED0 + i)
    S, G = int(rng.choice([3, 7, 16, 33, 64, 200])), int(rng.choice([10, 17, 129, 400]))
    n_picks = int(rng.choice([0, 1, 40, 700, 5000]))
    sig, dt = float(rng.choice([1.5, 3.0, 4.5])), float(rng.choice([0.25, 0.3, 0.6]))
    sign = bool(rng.integers(2))
    cfg = dict(S=S, G=G, n_picks=n_picks, sig=sig, dt=dt, sign=sign, seed=7000 + SEED0 + i)
    print(cfg)
    geom = synthetic.Geometry(S, G, L=float(rng.choice([40e3, 150e3])), n_query=3, seed=int(rng.integers(1 << 30)))
    P = synthetic.make_picks(geom, n_picks, seed=int(rng.integers(1 << 30))) if n_picks else np.zeros((0, 5))
    t0 = float(rng.uniform(-4.0, 4.0)) + float(rng.choice([0.0, 5000.0]))
    P[:, 0] += t0 - float(rng.uniform(-1.0, 1.0))
    trv = geom.travel_times().astype(np.float32)
    max_t = float(np.ceil(trv.max() + 1.0))
    A = np.stack([np.tile(np.arange(S), G), np.repeat(np.arange(G), S)], axis=0)
    want_S, want_M = E.extract_input_from_data(P, t0, np.arange(S), S, trv, A, max_t, sig, dt, use_sign_input=sign)
    hp = engine.HipPath(S, G, engine.csr_from_edges(torch.from_numpy(geom.A_sta_sta), S),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G), device=DEV)
    hp.set_sign_input(sign)
    sel = (P[:, 0] > t0 - 2.0 * sig) & (P[:, 0] < t0 + max_t + 2.0 * sig)           # process_utils.py:476
    Ps = P[sel]
    Slice, Mask = hp.embed_window(torch.from_numpy(Ps[:, 0].copy()).to(DEV), torch.from_numpy(Ps[:, 1].astype(np.int32)).to(DEV),
                                  torch.from_numpy(Ps[:, 4].astype(np.int32)).to(DEV), t0, max_t, sig, dt,
                                  torch.from_numpy(trv.reshape(-1, 2)).to(DEV))
    err = float((Slice.cpu() - torch.from_numpy(want_S)).abs().max())
    assert err <= 1e-6, (cfg, err)
    assert torch.equal(Mask.cpu(), torch.from_numpy(np.asarray(want_M, dtype=np.float32))), (cfg, int((Mask.cpu() != torch.from_numpy(np.asarray(want_M, dtype=np.float32))).sum()))


@pytest.mark.parametrize("i", range(max(2, N_CASES // 4)))
def test_random_case_device_apply_loop_on_an_irregular_product_graph(i):
    """The GPU-only apply loop (device embedding per listed product node, genie_set_subgraph_stations; batched tails) of a
    `use_subgraph` model against the oracle chain embed_oracle.extract_input_from_data (pairs as `A_src_in_sta`) ->
    genie_oracle.forward_fixed_source on the irregular edge lists -> the reference's stacking (process_continuous_days.py:766,797-805)."""
    from genie_amd import apply
    from oracle import embed_oracle as E
    from oracle import genie_oracle as O
    rng = np.random.default_rng(11000 + SEED0 + i)
    S, G = int(rng.choice([7, 16, 33])), int(rng.choice([17, 40, 70]))
    batch = int(rng.choice([1, 3, 16]))
    step_size = str(rng.choice(["half", "full", "partial"]))
    print(dict(S=S, G=G, batch=batch, step_size=step_size, seed=11000 + SEED0 + i))
    geom = synthetic.Geometry(S, G, L=60e3, n_query=12, seed=int(rng.integers(1 << 30)))
    P = synthetic.make_picks(geom, int(rng.choice([100, 400])), seed=int(rng.integers(1 << 30)))
    P[:, 0] = P[:, 0] * 0.25 + 5000.0
    P = P[np.argsort(P[:, 0], kind="stable")]
    trv = geom.travel_times().astype(np.float32)
    d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
    keep = np.zeros(d.shape, dtype=bool)
    keep[np.arange(G)[:, None], np.argsort(d, axis=1)[:, :int(rng.integers(2, min(6, S) + 1))]] = True
    keep |= rng.random(d.shape) < 0.2
    src_i, sta_i = np.nonzero(keep)
    pairs = np.stack((sta_i, src_i))
    A_in_sta, A_in_src, A_src_in_prod = graph.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
    ea = geom.edge_attr().reshape(G, S, 3)[src_i, sta_i]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
    w = Case("tiny_6x40").weights
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
    net.load_state_dict({k: v.clone() for k, v in w.items()})
    net.eval()
    gea = graph.GraphEdges(x=t(ea), edge_index=A_src_in_prod.to(DEV))
    net.set_adjacencies(A_in_sta.to(DEV), A_in_src.to(DEV), gea, gea, torch.from_numpy(pairs).to(DEV), torch.from_numpy(geom.A_src_src).to(DEV),
                        None, None, None, None, t(geom.locs), t(geom.x_grid))
    max_t = float(np.ceil(trv.max() + 1.0))
    Out_2, times = apply.apply_windows_device(net, geom, P, trv, step_size=step_size, min_required_picks=5, max_t=max_t, tail_batch=batch,
                                              pairs=pairs)
    assert 1 <= len(times) <= 60
    tsteps, offsets, step, n_overlap, dt_win = apply.window_schedule(P[:, 0], max_t, t_win=6.0, step_size=step_size)
    tsteps_abs = np.arange(tsteps.min() - 3.0, tsteps.max() + 3.0 + dt_win, dt_win)
    want = torch.zeros(Out_2.shape)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    for t0 in times:
        Slice, Mask = E.extract_input_from_data(P, float(t0), np.arange(S), S, trv, pairs, max_t, 3.0, 0.3)
        with torch.no_grad():
            _, x = O.forward_fixed_source(w, torch.from_numpy(Slice), torch.from_numpy(Mask), A_in_sta, A_in_src, c(ea), A_src_in_prod,
                                          torch.from_numpy(geom.A_src_src), c(geom.x_grid), c(geom.x_query), c(offsets.reshape(-1, 1)))
        cols, kp = apply.window_columns(tsteps_abs, float(t0), offsets, step_size == "half")
        want[:, cols] += x[:, kp, 0] / n_overlap
    assert float(want.abs().max()) > 0
    assert max_abs(Out_2.cpu(), want) <= 1e-5


def _two_output_case(cfg):
    """Graphs and one window of a drawn configuration (numpy / CPU tensors), as the 2-output sweep builds them."""
    S, G = cfg["S"], cfg["G"]
    geom = _geometry(cfg)
    win = synthetic.make_window(geom, cfg["n_picks"], seed=500 + cfg["seed"])
    Slice, Mask, ea = win["Slice"], win["Mask"], geom.edge_attr()
    if cfg["subgraph"]:
        rng = np.random.default_rng(900 + cfg["seed"])
        d = np.linalg.norm(geom.x_grid[:, None, :2] - geom.locs[None, :, :2], axis=2)
        keep = np.zeros(d.shape, dtype=bool)
        keep[np.arange(G)[:, None], np.argsort(d, axis=1)[:, :int(rng.integers(1, min(6, S) + 1))]] = True
        keep |= rng.random(d.shape) < rng.choice([0.0, 0.1, 0.5])
        src_i, sta_i = np.nonzero(keep)
        pairs = np.stack((sta_i, src_i))
        rows = src_i * S + sta_i
        Slice, Mask, ea = Slice[rows], Mask[rows], ea.reshape(G, S, 3)[src_i, sta_i]
        A_in_sta, A_in_src, A_src_in_prod = graph.subgraph_product_edges(geom.A_sta_sta, geom.A_src_src, pairs)
        A_src_in_sta = torch.from_numpy(pairs).long()
    else:
        A_in_sta, A_in_src, A_src_in_prod, A_src_in_sta = graph.cartesian_product_edges(geom.A_sta_sta, geom.A_src_src, S, G)
    return dict(geom=geom, Slice=Slice, Mask=Mask, ea=ea, A_in_sta=A_in_sta, A_in_src=A_in_src, A_src_in_prod=A_src_in_prod,
                A_src_in_sta=A_src_in_sta)


@pytest.mark.parametrize("i", range(max(2, N_CASES // 2)))
def test_random_case_one_model_object_across_changing_graphs(i):
    """One drop-in object, three `set_adjacencies` calls (graph A, graph B of another size and shape, graph A again) with an
    in-place weight update in between, as a job that processes several station sets does: every `(y, x)` equals the oracle on
    that graph with the weights of that moment (no stale context, table, static term or weight mirror)."""
    from oracle import genie_oracle as O
    cfgs = [_draw(13000 + 2 * (SEED0 + i)), _draw(13001 + 2 * (SEED0 + i))]
    for k in ("edges", "abspos", "stage"):
        cfgs[1][k] = cfgs[0][k]                     # one object: one model definition
    print(cfgs)
    w0 = _weights(cfgs[0])
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition=cfgs[0]["edges"],
                                                use_absolute_pos=cfgs[0]["abspos"])
    net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
    net.eval()
    cases = [_two_output_case(c_) for c_ in cfgs]
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    t = lambda a: c(a).to(DEV)
    for step, k in enumerate((0, 1, 0)):
        cfg, cs = cfgs[k], cases[k]
        geom = cs["geom"]
        if step == 2:                                # an optimizer-like in-place update of every parameter
            with torch.no_grad():
                for p in net.parameters():
                    p.mul_(1.01)
        gea = graph.GraphEdges(x=t(cs["ea"]), edge_index=cs["A_src_in_prod"].to(DEV))
        net.set_adjacencies(cs["A_in_sta"].to(DEV), cs["A_in_src"].to(DEV), gea, gea, cs["A_src_in_sta"].to(DEV),
                            torch.from_numpy(geom.A_src_src).to(DEV), None, None, None, None, t(geom.locs), t(geom.x_grid))
        with torch.no_grad():
            y, x = net.forward_fixed_source(t(cs["Slice"]), t(cs["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query),
                                            t(geom.t_query))
            w = {n: v.detach().cpu().clone() for n, v in net.state_dict().items()}
            So, okw = c(cs["Slice"]), {}
            if cfg["abspos"]:
                So = O.absolute_pos_inputs(So, c(geom.locs), c(geom.x_grid), cs["A_src_in_sta"])
            if cfg["edges"]:
                okw["pos_rel"] = (O.edge_pos_features(c(geom.locs), cs["A_in_sta"], cs["A_src_in_sta"][0]),
                                  O.edge_pos_features(c(geom.x_grid), cs["A_in_src"], cs["A_src_in_sta"][1]))
            yo, xo = O.forward_fixed_source(w, So, c(cs["Mask"]), cs["A_in_sta"], cs["A_in_src"], c(cs["ea"]), cs["A_src_in_prod"],
                                            torch.from_numpy(geom.A_src_src), c(geom.x_grid), c(geom.x_query), c(geom.t_query), **okw)
        ey, ex = max_abs(y.cpu(), yo), max_abs(x.cpu(), xo)
        assert ey <= 1e-5 and ex <= 1e-5, (step, cfg, ey, ex)


@pytest.mark.parametrize("i", range(max(2, N_CASES // 2)))
def test_random_case_adam_steps_match_the_oracle(i):
    """Four Adam(1e-3) steps of the `forward_fixed_source` training step (MSE against random labels) on the drawn model definition
    and graph shape: the loss of every step equals the oracle's autograd + torch.optim.Adam on the CPU to 1e-5 relative -- gradients,
    and the refresh of the library's weight mirror (static-term columns under their own names included) after every in-place update."""
    from oracle import genie_oracle as O
    cfg = _draw(15000 + SEED0 + i)
    cfg["n_picks"] = max(cfg["n_picks"], 50)
    print(cfg)
    cs = _two_output_case(cfg)
    geom = cs["geom"]
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    t = lambda a: c(a).to(DEV)
    w0 = _weights(cfg)
    rng = np.random.default_rng(99 + cfg["seed"])
    ly = c(rng.random((cfg["G"], 9, 1)))
    lx = c(rng.random((cfg["Q"], 9, 1)))
    net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV, use_updated_model_definition=cfg["edges"],
                                                use_absolute_pos=cfg["abspos"])
    net.load_state_dict({k: v.clone() for k, v in w0.items()}, strict=True)
    net.train()
    gea = graph.GraphEdges(x=t(cs["ea"]), edge_index=cs["A_src_in_prod"].to(DEV))
    net.set_adjacencies(cs["A_in_sta"].to(DEV), cs["A_in_src"].to(DEV), gea, gea, cs["A_src_in_sta"].to(DEV),
                        torch.from_numpy(geom.A_src_src).to(DEV), None, None, None, None, t(geom.locs), t(geom.x_grid))
    args = (t(cs["Slice"]), t(cs["Mask"]), None, None, None, t(geom.locs), t(geom.x_grid), t(geom.x_query), t(geom.t_query))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    got = []
    for _ in range(4):
        opt.zero_grad()
        y, x = net.forward_fixed_source(*args)
        loss = ((y - ly.to(DEV)) ** 2).mean() + ((x - lx.to(DEV)) ** 2).mean()
        loss.backward()
        opt.step()
        got.append(float(loss.detach()))
    w = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
    opt_o = torch.optim.Adam(list(w.values()), lr=1e-3)
    So, okw = c(cs["Slice"]), {}
    if cfg["abspos"]:
        So = O.absolute_pos_inputs(So, c(geom.locs), c(geom.x_grid), cs["A_src_in_sta"])
    if cfg["edges"]:
        okw["pos_rel"] = (O.edge_pos_features(c(geom.locs), cs["A_in_sta"], cs["A_src_in_sta"][0]),
                          O.edge_pos_features(c(geom.x_grid), cs["A_in_src"], cs["A_src_in_sta"][1]))
    want = []
    for _ in range(4):
        opt_o.zero_grad()
        yo, xo = O.forward_fixed_source(w, So, c(cs["Mask"]), cs["A_in_sta"], cs["A_in_src"], c(cs["ea"]), cs["A_src_in_prod"],
                                        torch.from_numpy(geom.A_src_src), c(geom.x_grid), c(geom.x_query), c(geom.t_query), **okw)
        lo = ((yo - ly) ** 2).mean() + ((xo - lx) ** 2).mean()
        lo.backward()
        opt_o.step()
        want.append(float(lo.detach()))
    print(got, want)
    assert min(want[1:]) < want[0]          # (the oracle's own curve: Adam at 1e-3 may overshoot within four steps -- sweep seed 62070)
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-5 * abs(b), (cfg, got, want)
