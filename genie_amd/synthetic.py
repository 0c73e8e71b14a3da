"""Seeded synthetic pick windows on fixed (N_stations, N_grid, N_picks) shapes.

Everything is plain numpy, deterministic given the seeds (geometry 1, picks 2), and independent of
the reference: this is *data*, used identically by the golden-vector generator, the parity tests,
`bench.py` and `__graft_entry__.smoke()`.

Semantics of the per-product-node features follow the reference's definition of the model input
(`/root/reference/Code/process_utils.py:262-275`): `Slice[p, 0/1] = exp(-r^2 / (2 sigma_t^2))` with r the
residual between the theoretical P / S arrival at (source g, station s) and the nearest pick of any
phase on station s; `[p, 2/3]` the same restricted to P- / S-labelled picks; `Mask = Slice > 0.01`;
product node id `p = g*S + s`.
"""
import numpy as np

from . import graph as _graph

VP, VS = 6000.0, 3400.0           # straight-ray stand-in for the travel-time model `trv`
KERNEL_SIG_T = 3.0                # train_config.yaml:17
SCALE_REL = 30000.0               # config.yaml:73
K_STA, K_SPC = 8, 15              # config.yaml:79-80
THRESH_MASK = 0.01                # process_utils.py:251,626

CONFIGS = {
    # name: (n_sta, n_grid, n_picks, L_metres, n_query)
    "cfg1_20x500": (20, 500, 2000, 100e3, 300),
    "cfg2_200x10k": (200, 10000, 50000, 300e3, 10000),
    "cfg4_2000x50k": (2000, 50000, 500000, 1000e3, 10000),
}


class Geometry(object):
    """Stations, source grid, base kNN graphs and static per-pair quantities."""

    def __init__(self, n_sta, n_grid, L=300e3, n_query=10000, seed=1, k_sta=K_STA, k_spc=K_SPC):
        rng = np.random.default_rng(seed)
        self.n_sta, self.n_grid, self.L = int(n_sta), int(n_grid), float(L)
        self.locs = np.stack([rng.uniform(0, L, n_sta), rng.uniform(0, L, n_sta),
                              rng.uniform(0.0, 2000.0, n_sta)], axis=1)
        self.x_grid = np.stack([rng.uniform(0, L, n_grid), rng.uniform(0, L, n_grid),
                                rng.uniform(-40000.0, 2000.0, n_grid)], axis=1)
        self.x_query = np.stack([rng.uniform(0, L, n_query), rng.uniform(0, L, n_query),
                                 rng.uniform(-40000.0, 2000.0, n_query)], axis=1)
        self.k_sta = _graph.k_sta_effective(k_sta, n_sta)
        self.k_spc = int(min(k_spc, n_grid - 1))
        # kNN on km coordinates as the reference does (process_utils.py:718-719)
        self.A_sta_sta = _graph.knn_graph(self.locs / 1000.0, self.k_sta)
        self.A_src_src = _graph.knn_graph(self.x_grid / 1000.0, self.k_spc)
        self.scale_x_extend = np.array([L, L, 42000.0]).reshape(1, 3)
        self.t_query = np.arange(-3.0, 3.75, 0.75).reshape(-1, 1)  # 9 origin-time offsets

    @property
    def n_prod(self):
        return self.n_sta * self.n_grid

    def travel_times(self, g_slice=None):
        """[G', S, 2] straight-ray P/S travel times (float64)."""
        xg = self.x_grid if g_slice is None else self.x_grid[g_slice]
        d = np.linalg.norm(xg[:, None, :] - self.locs[None, :, :], axis=2)
        return np.stack([d / VP, d / VS], axis=2)

    def edge_attr(self, g_slice=None):
        """`spatial_vals` = (x_grid[g] - locs[s]) / scale_x_extend, [G'*S, 3] float32
        (process_continuous_days.py:630)."""
        xg = self.x_grid if g_slice is None else self.x_grid[g_slice]
        v = (xg[:, None, :] - self.locs[None, :, :]) / self.scale_x_extend.reshape(1, 1, 3)
        return v.reshape(-1, 3).astype(np.float32)

    @property
    def max_t(self):
        return float(np.sqrt(2 * self.L ** 2 + 42000.0 ** 2) / VS)


def make_picks(geom, n_picks, seed=2, window=0):
    """Picks `[n, 5]` float64 columns (t, station index, amp, prob, phase) like the reference's P
    (utils.py:983). 80 % come from synthetic events, 20 % are uniform noise; sorted by time."""
    rng = np.random.default_rng([seed, window])
    S = geom.n_sta
    n_ev = max(1, int(round(0.8 * n_picks / (1.6 * S))))
    rows = []
    for _ in range(n_ev):
        g = int(rng.integers(0, geom.n_grid))
        t_org = rng.uniform(-3.0, 3.0)
        tt = geom.travel_times(slice(g, g + 1))[0]                      # [S, 2]
        for ph in (0, 1):
            keep = rng.random(S) < 0.8
            t = t_org + tt[keep, ph] + rng.normal(0.0, 0.1, int(keep.sum()))
            rows.append(np.stack([t, np.nonzero(keep)[0].astype(np.float64),
                                  np.ones_like(t), np.ones_like(t), np.full_like(t, ph)], axis=1))
    ev = np.concatenate(rows, axis=0)
    if ev.shape[0] > int(0.8 * n_picks):
        ev = ev[rng.permutation(ev.shape[0])[: int(0.8 * n_picks)]]
    n_noise = n_picks - ev.shape[0]
    t = rng.uniform(-6.0, geom.max_t + 6.0, n_noise)
    noise = np.stack([t, rng.integers(0, S, n_noise).astype(np.float64), np.ones(n_noise),
                      np.ones(n_noise), rng.integers(0, 2, n_noise).astype(np.float64)], axis=1)
    P = np.concatenate([ev, noise], axis=0)
    return P[np.argsort(P[:, 0], kind="stable")]


def _nearest_residual(pick_t, pick_s, S, query_t):
    """|query - nearest pick on the same station|; query_t [G, S]; inf where a station has no pick."""
    out = np.full(query_t.shape, np.inf)
    if pick_t.size == 0:
        return out
    order = np.lexsort((pick_t, pick_s))
    pt, ps = pick_t[order], pick_s[order].astype(np.int64)
    start = np.searchsorted(ps, np.arange(S), side="left")
    stop = np.searchsorted(ps, np.arange(S), side="right")
    for s in range(S):
        a = pt[start[s]:stop[s]]
        if a.size == 0:
            continue
        q = query_t[:, s]
        ip = np.searchsorted(a, q)
        lo = np.clip(ip - 1, 0, a.size - 1)
        hi = np.clip(ip, 0, a.size - 1)
        out[:, s] = np.minimum(np.abs(q - a[lo]), np.abs(q - a[hi]))
    return out


def make_slice_mask(geom, P, t0=0.0, sigma_t=KERNEL_SIG_T, g_slice=None):
    """Slice[P', 4] float32, Mask[P', 4] float32 for the window starting at t0 (process_utils.py:262-275)."""
    tt = geom.travel_times(g_slice)                                     # [G', S, 2]
    S = geom.n_sta
    t, s, ph = P[:, 0] - t0, P[:, 1], P[:, 4]
    feats = []
    for col, (phase_arrival, sel) in enumerate([(0, None), (1, None), (0, 0), (1, 1)]):
        m = np.ones(t.shape, dtype=bool) if sel is None else (ph == sel)
        r = _nearest_residual(t[m], s[m], S, tt[:, :, phase_arrival])
        with np.errstate(over="ignore", invalid="ignore"):
            feats.append(np.where(np.isfinite(r), np.exp(-0.5 * (r ** 2) / (sigma_t ** 2)), 0.0))
    Slice = np.stack(feats, axis=2).reshape(-1, 4).astype(np.float32)
    Mask = (Slice > THRESH_MASK).astype(np.float32)
    return Slice, Mask


def make_window(geom, n_picks, seed=2, window=0, g_slice=None):
    """One synthetic pick window: dict with Slice, Mask and the pick arrays forward* takes."""
    P = make_picks(geom, n_picks, seed=seed, window=window)
    Slice, Mask = make_slice_mask(geom, P, 0.0, g_slice=g_slice)
    order = np.lexsort((P[:, 0], P[:, 1]))                             # process_utils.py:293
    return {"Slice": Slice, "Mask": Mask, "P": P,
            "tpick": P[order, 0].astype(np.float32), "ipick": P[order, 1].astype(np.int64),
            "phase_label": P[order, 4].astype(np.float32).reshape(-1, 1), "n_picks": int(P.shape[0])}


def training_sample(geom, n_picks, n_src=4, seed=3, window=0):
    """Everything one training-mode call `mz(*input_tensors)` consumes (train_GENIE_model.py:1770-1786) for a synthetic window
    on `geom`, plus seeded random labels of the right shapes: dict of numpy arrays. The association-head tables come from
    `graph.time_pointers` (utils.py:602-622 as called at train_GENIE_model.py:1364: k = 10, dt = kernel_sig_t / 5, win = 2 x
    kernel_sig_t)."""
    win = make_window(geom, n_picks, seed=seed, window=window)
    trv = geom.travel_times().astype(np.float32)                                   # [G, S, 2]
    max_t = float(np.ceil(trv.max()))
    A_edges_p, A_edges_s, dt_partition = _graph.time_pointers(trv, max_t=max_t, dt=KERNEL_SIG_T / 5.0, k=10, win=2.0 * KERNEL_SIG_T)
    rng = np.random.default_rng([seed, window, 17])
    nodes = rng.choice(geom.n_grid, n_src, replace=False)
    x_query_src = geom.x_grid[nodes] + rng.normal(0.0, 500.0, (n_src, 3))
    tq_sample = rng.uniform(-2.0, 2.0, n_src).astype(np.float32)
    d = np.linalg.norm(x_query_src[:, None, :] - geom.locs[None, :, :], axis=2)
    trv_out_q = np.stack([d / VP, d / VS], axis=2).astype(np.float32)
    keep = (win["tpick"] > dt_partition[0] + 0.5) & (win["tpick"] < dt_partition[-1] - 0.5)    # inside the time-pointer table
    out = {"Slice": win["Slice"], "Mask": win["Mask"], "tpick": win["tpick"][keep], "ipick": win["ipick"][keep],
           "phase_label": win["phase_label"][keep], "A_edges_p": A_edges_p, "A_edges_s": A_edges_s,
           "dt_partition": dt_partition.astype(np.float32), "tlatent": trv.reshape(-1, 2), "x_query_src": x_query_src.astype(np.float32),
           "tq_sample": tq_sample, "trv_out_q": trv_out_q}
    T = geom.t_query.shape[0]
    out["Lbls"] = rng.random((geom.n_grid, T)).astype(np.float32) * (rng.random((geom.n_grid, 1)) < 0.1)
    out["Lbls_query"] = rng.random((geom.x_query.shape[0], T)).astype(np.float32) * (rng.random((geom.x_query.shape[0], 1)) < 0.1)
    out["pick_lbls"] = (rng.random((n_src, int(keep.sum()), 2)) < 0.05).astype(np.float32)
    return out
