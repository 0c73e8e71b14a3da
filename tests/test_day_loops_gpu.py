"""GPU: the per-window pick selection on the device and the two other per-day loops of the caller (refine pass
process_continuous_days.py:926-980, association pass :1020-1065; genie_amd/apply.py) against the reference's fixtures and the oracle
chain embed_oracle -> genie_oracle. Tolerances: index / order work exact; outputs 1e-5 absolute (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from genie_amd import apply, graph, module, synthetic
from tests.util import GOLDEN_DIR, Case, max_abs

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)


def _c(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


class _Setup(object):
    """A station file of `S_all` stations of which the model uses `ind_use`, a source grid, picks with ABSOLUTE station indices in
    file (not time) order around a few events, the model with adjacencies + time-pointer tables set, and the oracle's view of it."""

    def __init__(self, S_all=15, n_use=11, G=70, n_picks=500, seed=91, weights="assoc_7x45"):
        rng = np.random.default_rng(seed)
        self.geom_all = ga = synthetic.Geometry(S_all, G, L=60e3, n_query=8, seed=seed)
        self.ind_use = np.sort(rng.choice(S_all, n_use, replace=False))
        self.S, self.G = n_use, G
        self.locs = ga.locs[self.ind_use]
        self.A_sta_sta = graph.knn_graph(self.locs / 1000.0, graph.k_sta_effective(8, n_use))
        P = synthetic.make_picks(ga, n_picks, seed=seed + 1)
        P[:, 0] = P[:, 0] * 0.5 + 7000.0
        self.P = P[rng.permutation(P.shape[0])]
        self.trv_all = ga.travel_times().astype(np.float32)                       # [G, S_all, 2]
        self.trv_use = np.ascontiguousarray(self.trv_all[:, self.ind_use])        # [G, S, 2] = x_grids_trv of the model's stations
        self.max_t = float(np.ceil(self.trv_all.max() + 1.0))
        self.sig, self.dt = 3.0, 0.3
        ea = ((ga.x_grid[:, None, :] - self.locs[None, :, :]) / ga.scale_x_extend.reshape(1, 1, 3)).reshape(-1, 3).astype(np.float32)
        self.ea = ea
        z = np.load(os.path.join(GOLDEN_DIR, weights + ".npz"))
        from oracle import genie_oracle as O
        self.w = O.weights_from_npz(z)
        self.A_edges_p, self.A_edges_s, self.dt_partition = graph.time_pointers(self.trv_use, max_t=self.max_t, dt=self.sig / 5.0, k=10,
                                                                                 win=2.0 * self.sig)
        self.net = net = module.GCN_Detection_Network_extended(lambda x: x, lambda x: x, device=DEV)
        net.load_state_dict({k: v.clone() for k, v in self.w.items()}, strict=True)
        net.eval()
        net.set_adjacencies_base(torch.from_numpy(self.A_sta_sta), torch.from_numpy(ga.A_src_src), _t(ea), _t(self.locs), _t(ga.x_grid),
                                 torch.from_numpy(self.A_edges_p).to(DEV), torch.from_numpy(self.A_edges_s).to(DEV),
                                 _t(self.dt_partition), _t(self.trv_use.reshape(-1, 2)))
        self.picks = apply.ResidentPicks(self.P, self.ind_use, S_all, DEV)
        self.leg = apply.GridLeg(net, ga.x_grid, self.trv_use)
        self.tq = np.arange(-3.0, 3.75, 0.75).reshape(-1, 1)
        self.A_pairs = np.stack([np.tile(np.arange(n_use), G), np.repeat(np.arange(G), n_use)], axis=0)

    def oracle_window(self, t0):
        from oracle import embed_oracle as E
        return E.extract_input_from_data(self.P, float(t0), self.ind_use, self.geom_all.n_sta, self.trv_all, self.A_pairs, self.max_t,
                                         self.sig, self.dt)


@pytest.mark.parametrize("name", ["picks_14x60_a", "picks_14x60_b", "picks_14x60_c"])
def test_device_pick_selection_and_embedding_match_the_reference_call(name):
    """Everything one `extract_input_from_data` call of the reference returns (process_utils.py:460-642 -> :644-699), produced on the GPU
    from `ResidentPicks`: Slice / Mask by genie_embed_window on the window's contiguous pick range (1e-6 / exact), the pick lists by
    the stable station sort on the device (exact). Station subset of the station file, picks in file order."""
    from genie_amd import engine
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    P, t0, ind = z["P"], float(z["t0"]), z["ind_use"]
    max_t, sig, dt, n_all, G = float(z["max_t"]), float(z["kernel_sig_t"]), float(z["dt"]), int(z["n_sta_all"]), int(z["n_grid"])
    rp = apply.ResidentPicks(P, ind, n_all, DEV)
    tp, ip, ph, idx = rp.pick_inputs(t0, max_t, sig)
    assert tp.is_cuda and np.array_equal(tp.cpu().numpy(), z["lp_times"]) and np.array_equal(ip.cpu().numpy(), z["lp_stations"])
    assert np.array_equal(ph.cpu().numpy(), z["lp_phases"]) and np.array_equal(rp.meta(idx), z["lp_meta"])
    geom = synthetic.Geometry(n_all, G, L=90e3, n_query=5, seed=61)
    locs = geom.locs[ind]
    A_sta = graph.knn_graph(locs / 1000.0, graph.k_sta_effective(8, len(ind)))
    hp = engine.HipPath(len(ind), G, engine.csr_from_edges(torch.from_numpy(A_sta), len(ind)),
                        engine.csr_from_edges(torch.from_numpy(geom.A_src_src), G), device=DEV)
    a = rp.embed_args(t0, max_t, sig)
    Slice, Mask = hp.embed_window(a[0], a[1], a[2], t0, max_t, sig, dt, _t(z["trv_times"][:, ind].reshape(-1, 2)))
    assert float((Slice.cpu() - torch.from_numpy(z["Slice"])).abs().max()) <= 1e-6
    assert torch.equal(Mask.cpu(), torch.from_numpy(z["Mask"].astype(np.float32)))


def test_refine_pass_matches_the_oracle_chain():
    """process_continuous_days.py:926-980: per candidate source a random query cloud, forward_fixed_source on the window starting at its
    origin time read out at the cloud, refined source = argmax. Oracle chain: embed_oracle.extract_input_from_data ->
    genie_oracle.forward_fixed_source_structured at the same cloud (same seeded draws) -> numpy argmax. One candidate sits in a quiet
    stretch of the day (no pick: all-zero read-out, first query and first offset), one at the region's corner (part of its cloud is
    outside and dropped)."""
    from oracle import genie_oracle as O
    s = _Setup()
    ga = s.geom_all
    rng = np.random.default_rng(5)
    nodes = rng.choice(s.G, 4, replace=False)
    srcs = np.concatenate((ga.x_grid[nodes], rng.uniform(6995.0, 7010.0, (4, 1)), np.full((4, 1), 0.5)), axis=1)
    srcs = np.concatenate((srcs, [[20e3, 30e3, -5e3, 30000.0, 0.5]], [[500.0, 59.6e3, 1500.0, 7002.0, 0.5]]), axis=0)
    off_min, off_rng = np.array([[-5e3, -5e3, -3e3]]), np.array([[10e3, 10e3, 6e3]])
    ident = lambda x: x
    ranges = ((0.0, 60e3), (0.0, 60e3), (-40e3, 2e3))
    got, order = apply.refine_sources([s.leg], s.picks, srcs, s.locs, s.tq, s.max_t, off_min, off_rng, 300, ident, ident, *ranges,
                                      kernel_sig_t=s.sig, dt_embed=s.dt, rand=np.random.RandomState(77).rand)
    assert got.shape == (6, 5) and np.all(np.diff(got[:, 3]) >= 0)
    # the same pass with the cloud's arithmetic on the device (`ftrns2_device`): the same refined sources, bit for bit
    got_d, order_d = apply.refine_sources([s.leg], s.picks, srcs, s.locs, s.tq, s.max_t, off_min, off_rng, 300, ident, ident, *ranges,
                                          kernel_sig_t=s.sig, dt_embed=s.dt, rand=np.random.RandomState(77).rand, ftrns2_device=ident)
    assert np.array_equal(order, order_d) and np.array_equal(got, got_d)
    rs = np.random.RandomState(77)
    sta_nbr, src_nbr = graph.neighbour_table(s.A_sta_sta, s.S), graph.neighbour_table(ga.A_src_src, s.G)
    n_out = 0
    for i in range(6):
        Xc = srcs[i, 0:3].reshape(1, -1) + (rs.rand(300, 3) * off_rng + off_min)
        inside = np.all([(Xc[:, a] > ranges[a][0]) & (Xc[:, a] < ranges[a][1]) for a in range(3)], axis=0)
        n_out += int((~inside).sum())
        Xc = Xc[inside]
        Slice, Mask = s.oracle_window(srcs[i, 3])
        row = got[np.nonzero(order == i)[0][0]]
        if i == 4:
            assert float(np.abs(Slice).max()) == 0.0
            assert np.array_equal(row[0:3], Xc[0]) and row[3] == srcs[i, 3] + s.tq[0, 0] and row[4] == 0.0
            continue
        _, x = O.forward_fixed_source_structured(s.w, torch.from_numpy(Slice), torch.from_numpy(Mask), sta_nbr, src_nbr, _c(s.ea),
                                                 torch.from_numpy(ga.A_src_src), _c(ga.x_grid), _c(Xc), _c(s.tq), s.S, s.G)
        x = x[:, :, 0].numpy()
        ip = int(np.argmax(x.max(1)))
        it = int(np.argmax(x[ip]))
        assert abs(row[4] - x.max()) <= 1e-5 and x.max() > 1e-3
        # the refined location / time is the oracle's argmax, or (values within the tolerance of each other) an equally good one
        jp = int(np.nonzero(np.all(Xc == row[0:3], axis=1))[0][0])
        jt = int(np.argmin(np.abs(srcs[i, 3] + s.tq[:, 0] - row[3])))
        assert x[ip, it] - x[jp, jt] <= 2e-5
    assert n_out > 50


def test_association_pass_matches_the_oracle_chain():
    """process_continuous_days.py:1020-1065: per refined source a 4-output forward_fixed on the window starting at its origin time, with
    the window's pick lists, one spatial query, the source as the only candidate: Out_p_save / Out_s_save per pick. Oracle chain:
    embed_oracle.extract_input_from_data + extract_pick_inputs_from_data -> genie_oracle.forward_fixed."""
    from oracle import embed_oracle as E
    from oracle import genie_oracle as O
    s = _Setup()
    ga = s.geom_all
    rng = np.random.default_rng(6)
    nodes = rng.choice(s.G, 3, replace=False)
    srcs = np.concatenate((ga.x_grid[nodes] + rng.normal(0, 300.0, (3, 3)), rng.uniform(6996.0, 7008.0, (3, 1)), np.full((3, 1), 0.5)), 1)
    srcs = np.concatenate((srcs, [[20e3, 30e3, -5e3, 40000.0, 0.5]]), axis=0)            # a window without picks
    srcs = srcs[np.argsort(srcs[:, 3])]
    d = np.linalg.norm(srcs[:, None, 0:3] - s.locs[None, :, :], axis=2)
    trv_out_srcs = np.stack((d / synthetic.VP, d / synthetic.VS), axis=2).astype(np.float32)
    ident = lambda x: x
    x_save = np.array([1000.0, 2000.0, 0.0])
    Out_p, Out_s, Save_picks, lp_meta = apply.associate_sources([s.leg], s.picks, srcs, s.locs, s.tq, s.max_t, trv_out_srcs, ident, x_save,
                                                                kernel_sig_t=s.sig, dt_embed=s.dt)
    A_in_sta, A_in_src, A_src_in_prod, _ = graph.cartesian_product_edges(s.A_sta_sta, ga.A_src_src, s.S, s.G)
    seen = 0
    for i in range(srcs.shape[0]):
        P_slice = E.window_pick_slice(s.P, srcs[i, 3], s.ind_use, s.max_t, s.sig)
        lt, ls, lph, lm = E.extract_pick_inputs_from_data(P_slice, ga.n_sta, s.ind_use, srcs[i, 3], s.max_t)
        assert np.array_equal(Save_picks[i][:, 0], lt) and np.array_equal(Save_picks[i][:, 1], ls.astype(np.float64))
        assert np.array_equal(lp_meta[i], lm)
        assert Out_p[i].shape == (len(lt),) and Out_s[i].shape == (len(lt),) and Out_p[i].is_cuda
        if len(lt) == 0:
            continue
        Slice, Mask = s.oracle_window(srcs[i, 3])
        xs = x_save.copy()
        xs[2] = srcs[i, 2]
        with torch.no_grad():
            out = O.forward_fixed(s.w, torch.from_numpy(Slice), torch.from_numpy(Mask), A_in_sta, A_in_src, _c(s.ea), A_src_in_prod,
                                  torch.from_numpy(ga.A_src_src), torch.from_numpy(s.A_edges_p), torch.from_numpy(s.A_edges_s),
                                  _c(s.dt_partition), _c(s.trv_use.reshape(-1, 2)), _c(lt), torch.from_numpy(ls).long(),
                                  _c(lph.reshape(-1, 1)), _c(ga.x_grid), _c(xs.reshape(1, 3)), _c(srcs[i, 0:3].reshape(1, 3)), _c(s.tq),
                                  torch.zeros(1), _c(trv_out_srcs[i:i + 1]), s.S)
        ep, es = max_abs(Out_p[i].cpu(), out[2][0, :, 0]), max_abs(Out_s[i].cpu(), out[3][0, :, 0])
        assert ep <= 1e-5 and es <= 1e-5, (i, ep, es)
        seen += int(float(out[2].abs().max()) > 1e-4)
    assert seen >= 2 and any(len(m) == 0 for m in lp_meta)


def test_detection_to_association_chain_matches_its_steps():
    """process_continuous_days.py:811-1105 in one call (apply.detect_refine_associate) against the same statements made step by step
    from pieces that each have their own oracle test (detect_sources, refine_sources, associate_sources, local_marching): what the
    composition adds is the glue -- travel times of the refined sources, X_save, the second LocalMarching's match back (cKDTree in
    position + 3500 x time, np.unique), the final sort -- and those are restated here in the reference's own words."""
    from scipy.spatial import cKDTree
    from genie_amd import postproc
    s = _Setup()
    rng = np.random.default_rng(23)
    Q, dt_win, src_t_kernel, thresh = 150, 0.75, 5.0, 0.15
    xq = np.c_[rng.uniform(0, 60e3, (Q, 2)), rng.uniform(-30e3, 0, Q)]
    ts = 6990.0 + np.arange(400) * dt_win
    out = np.zeros((Q, len(ts)), dtype=np.float32)
    far = int(np.argmax(np.linalg.norm(xq - xq[3], axis=1)))
    # the picks of the setup lie in 6997 .. 7017 s: two sources inside that stretch (one of them a double peak), two in quiet stretches
    centres = [(xq[3], 7001.0), (xq[3] + [2e3, 0, 0], 7002.5), (xq[far], 7011.0), (xq[90], 7100.0), (xq[91], 7190.0)]
    for c, t0 in centres:
        d = np.linalg.norm((xq - c) * np.array([1, 1, 0.3]), axis=1)
        out += (0.7 * np.exp(-0.5 * (d / 12e3) ** 2)[:, None] * np.exp(-0.5 * ((ts - t0) / 3.0) ** 2)[None, :]).astype(np.float32)
    Out_2 = torch.from_numpy(out).to(DEV)
    ident = lambda x: x
    ranges = ((0.0, 60e3), (0.0, 60e3), (-40e3, 2e3))
    off_min, off_rng = np.array([[-5e3, -5e3, -3e3]]), np.array([[10e3, 10e3, 6e3]])
    tc_win, sp_win, break_win = src_t_kernel * 1.35, 20e3, 15.0

    def trv(locs, srcs):                                                        # [n, S, 2], the call shape of the reference's `trv`
        d = torch.linalg.norm(locs[None, :, :] - srcs[:, None, :], dim=2)
        return torch.stack((d / 6000.0, d / 3500.0), dim=2)

    kw = dict(kernel_sig_t=s.sig, dt_embed=s.dt)
    got = apply.detect_refine_associate([s.leg], s.picks, Out_2, xq, ts, s.locs, trv, s.tq, s.max_t, ident, ident, *ranges, off_min, off_rng,
                                        200, thresh, src_t_kernel, dt_win, break_win, tc_win, sp_win, rand=np.random.RandomState(3).rand,
                                        ftrns2_device=ident, **kw)
    srcs = postproc.detect_sources(Out_2, xq, ts, ident, thresh, src_t_kernel, dt_win, break_win, tc_win, sp_win)
    assert 3 <= len(srcs) <= len(centres) and np.array_equal(got["srcs"], srcs)
    ref, _ = apply.refine_sources([s.leg], s.picks, srcs, s.locs, s.tq, s.max_t, off_min, off_rng, 200, ident, ident, *ranges,
                                  rand=np.random.RandomState(3).rand, ftrns2_device=ident, **kw)
    trv_out = trv(_t(s.locs), _t(ref[:, 0:3]))                                                                          # :1004
    Op, Os, Sp, Lm = apply.associate_sources([s.leg], s.picks, ref, s.locs, s.tq, s.max_t, trv_out, ident,
                                             np.array([ranges[0][0], ranges[1][0], 0.0]), **kw)
    ref_1 = postproc.local_marching(ref, ident, tc_win=tc_win, sp_win=sp_win, scale_depth=0.2, n_steps_max=2, use_directed=False)  # :1075
    tree = cKDTree(np.concatenate((ref, 3500.0 * ref[:, [3]]), axis=1)[:, [0, 1, 2, 5]])                                # :1083
    ip = np.unique(tree.query(np.concatenate((ref_1[:, 0:3], 3500.0 * ref_1[:, [3]]), axis=1))[1])                      # :1084-1085
    want = ref[ip]
    io = np.argsort(want[:, 3])                                                                                         # :1097
    assert np.array_equal(got["srcs_refined"], want[io]) and len(got["Out_p_save"]) == len(ip)
    assert torch.equal(got["trv_out_srcs"], trv(_t(s.locs), _t(want[io][:, 0:3])))                                      # :1092, :1099
    n_assoc = 0
    for j, i in enumerate(ip[io]):
        assert torch.equal(got["Out_p_save"][j], Op[i]) and torch.equal(got["Out_s_save"][j], Os[i])
        assert np.array_equal(got["Save_picks"][j], Sp[i]) and np.array_equal(got["lp_meta"][j], Lm[i])
        n_assoc += int(Op[i].numel() > 0)
    assert n_assoc >= 2
    # nothing above the threshold: the caller's early exit (:886-888)
    none = apply.detect_refine_associate([s.leg], s.picks, torch.zeros_like(Out_2), xq, ts, s.locs, trv, s.tq, s.max_t, ident, ident, *ranges,
                                         off_min, off_rng, 200, thresh, src_t_kernel, dt_win, break_win, tc_win, sp_win, **kw)
    assert len(none["srcs"]) == 0 and none["Out_p_save"] == []


def test_two_identical_grid_legs_equal_one():
    """The reference averages the loops' outputs over its source grids (`x_grid_ind` loops, `/ n_scale_x_grid_1`, process_continuous_days.py
    :972, :1054-1055). With the same grid listed twice the average is exact in fp32 (x / 2 + x / 2), so both passes must return what one
    leg returns, bit for bit: the accumulation over legs, the scale and the per-leg embedding calls are what this exercises."""
    s = _Setup()
    ga = s.geom_all
    rng = np.random.default_rng(9)
    nodes = rng.choice(s.G, 3, replace=False)
    srcs = np.concatenate((ga.x_grid[nodes], rng.uniform(6998.0, 7008.0, (3, 1)), np.full((3, 1), 0.5)), axis=1)
    off_min, off_rng = np.array([[-5e3, -5e3, -3e3]]), np.array([[10e3, 10e3, 6e3]])
    ident = lambda x: x
    ranges = ((0.0, 60e3), (0.0, 60e3), (-40e3, 2e3))
    kw = dict(kernel_sig_t=s.sig, dt_embed=s.dt)
    outs = []
    for legs in ([s.leg], [s.leg, s.leg]):
        ref, order = apply.refine_sources(legs, s.picks, srcs, s.locs, s.tq, s.max_t, off_min, off_rng, 250, ident, ident, *ranges,
                                          rand=np.random.RandomState(5).rand, ftrns2_device=ident, **kw)
        d = np.linalg.norm(s.locs[None, :, :] - ref[:, None, 0:3], axis=2)
        trv = np.stack((d / 6000.0, d / 3500.0), axis=2)
        Op, Os, Sp, Lm = apply.associate_sources(legs, s.picks, ref, s.locs, s.tq, s.max_t, trv, ident, np.array([0.0, 0.0, 0.0]), **kw)
        outs.append((ref, order, Op, Os, Sp))
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert len(a[2]) == len(b[2]) == 3 and sum(int(o.numel()) for o in a[2]) > 0
    for i in range(3):
        assert torch.equal(a[2][i], b[2][i]) and torch.equal(a[3][i], b[3][i]) and np.array_equal(a[4][i], b[4][i])
