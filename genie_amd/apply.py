"""Sliding-window apply loop around `forward_fixed_source` — the caller of the hot path
(`/root/reference/Code/process_continuous_days.py:761-810`, setup `:357-381, :571, :725-754`).

Semantics kept from the reference:
* prediction window of `n_resolution = 9` origin-time offsets `arange(-t_win/2, t_win/2 + dt_win, dt_win)` (`:357-362`);
* window starts `tsteps = arange(max(0, min(pick_t) - max_t), min(day_len, max(pick_t)), step)`, `step` and `n_overlap`
  from `step_size` in {'full', 'partial', 'half'} (`:367-381, :571`);
* windows with fewer than `min_required_picks` picks in `[t0 - t_win, t0 + max_t + t_win]` are skipped (`:725-741`);
* `Out_2[:, idx(tsteps_abs[idx(t0)] + offsets)] += x[:, :, 0] / n_overlap / n_grids`, dropping the last offset when
  `step_size == 'half'` (`:766, :797-805`): the window start is snapped to `tsteps_abs` first, and a column listed twice is
  written once (numpy fancy `+=`), see `window_columns`;
* windows with no pick in the embedding range are skipped (`:792-793`).
Differences by design: `Out_2` stays on the GPU and is accumulated with `index_add_` (the reference copies every window's
output to the host, `:803-805`); nothing in the loop synchronises with the host.

`apply_windows` takes any host embedding callable (default: `genie_amd.synthetic.make_slice_mask`, the exact
nearest-pick semantics of `process_utils.py:262-275`). `apply_windows_device` is the GPU-only loop: picks and the static
travel-time table stay resident on the device and every window's `Slice/Mask` is produced by `genie_embed_window`
(`extract_input_from_data`, `process_utils.py:460-642`), so a window costs no host->device copy at all.
"""
import numpy as np
import torch

from . import synthetic


def window_schedule(pick_times, max_t, day_len=86400.0, t_win=6.0, n_resolution=9, step_size="half"):
    """(tsteps, offsets, step, n_overlap, dt_win) exactly as process_continuous_days.py:357-381,571 builds them."""
    dt_win = float(np.diff(np.linspace(-t_win / 2.0, t_win / 2.0, n_resolution))[0])
    if step_size == "full":
        step, n_overlap = n_resolution * dt_win, 1.0
    elif step_size == "partial":
        step, n_overlap = (n_resolution / 3) * dt_win, 3.0
    elif step_size == "half":
        step, n_overlap = int(np.floor(n_resolution / 2)) * dt_win, 2.0
    else:
        raise ValueError("step_size must be 'full', 'partial' or 'half'")
    t = np.asarray(pick_times, dtype=np.float64)
    tsteps = np.arange(max(0.0, t.min() - max_t), min(day_len, t.max()), step)
    offsets = np.arange(-t_win / 2.0, t_win / 2.0 + dt_win, dt_win)[:n_resolution]
    return tsteps, offsets, step, n_overlap, dt_win


def windows_with_enough_picks(pick_times, tsteps, max_t, t_win, min_required_picks):
    """Keep window starts whose neighbourhood holds >= min_required_picks picks (process_continuous_days.py:725-741:
    ball of radius t_win + max_t/2 around t0 + max_t/2)."""
    t = np.sort(np.asarray(pick_times, dtype=np.float64))
    c = tsteps + max_t / 2.0
    r = t_win + max_t / 2.0
    n = np.searchsorted(t, c + r, side="right") - np.searchsorted(t, c - r, side="left")
    return tsteps[n >= max(1, int(min_required_picks))]


def is_ascending(grid):
    grid = np.asarray(grid)
    return bool(grid.shape[0] < 2 or np.all(grid[1:] >= grid[:-1]))


def nearest_index(grid, values, ascending=None):
    """Index of the `grid` entry nearest to every value (ties -> the lower index, as `np.abs(grid - v).argmin()` gives): a
    binary search + a choice between the two bracketing entries when `grid` is ascending (the reference's `tsteps_abs`,
    process_continuous_days.py:374-381, queried through a cKDTree at :766 / :797) -- O(log n) per value instead of a dense scan
    of a day-long grid per window; any other grid takes the dense scan."""
    grid = np.asarray(grid, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    n = grid.shape[0]
    if ascending is None:
        ascending = is_ascending(grid)
    if n < 2 or not ascending:
        return np.abs(grid.reshape(-1, 1) - values.reshape(1, -1)).argmin(0)
    hi = np.clip(np.searchsorted(grid, values, side="left"), 0, n - 1)
    lo = np.clip(hi - 1, 0, n - 1)
    take_lo = np.abs(grid[lo] - values) <= np.abs(grid[hi] - values)
    idx = np.where(take_lo, lo, hi)
    # equal grid entries: argmin returns the first of them
    first = np.searchsorted(grid, grid[idx], side="left")
    return first.astype(np.int64)


def window_columns(tsteps_abs, t0, offsets, drop_last, ascending=None):
    """Columns of `Out_2` one window adds to, as process_continuous_days.py:766,797-805 finds them: the window start is first
    SNAPPED to its nearest `tsteps_abs` entry (`tree_tsteps.query`, :766), the nine offsets are added to that entry and
    looked up again (:797), the last one is dropped for step_size 'half' (:802-803). numpy's `Out_2[:, cols] += vals` writes
    a column that appears twice only once (the LAST occurrence wins), so duplicates are reduced to their last occurrence.
    Returns (cols int64 [m], keep int64 [m]: offset index feeding each column)."""
    tsteps_abs = np.asarray(tsteps_abs, dtype=np.float64)
    if ascending is None:
        ascending = is_ascending(tsteps_abs)          # (loops over many windows pass it: one O(n) check per grid, not per window)
    i0 = int(nearest_index(tsteps_abs, np.asarray([t0], dtype=np.float64), ascending)[0])
    ip = nearest_index(tsteps_abs, tsteps_abs[i0] + np.asarray(offsets, dtype=np.float64), ascending)
    if drop_last:
        ip = ip[:-1]
    keep = np.array([k for k in range(len(ip)) if ip[k] not in ip[k + 1:]], dtype=np.int64)
    return ip[keep].astype(np.int64), keep


def picks_in_embed_range(pick_times_sorted, t0, max_t, kernel_sig_t):
    """[lo, hi) of the picks with t0 - 2 sigma < t < t0 + max_t + 2 sigma (process_utils.py:476); the reference skips a window
    whose range is empty (process_continuous_days.py:792-793)."""
    lo = int(np.searchsorted(pick_times_sorted, t0 - 2.0 * kernel_sig_t, side="right"))
    hi = int(np.searchsorted(pick_times_sorted, t0 + max_t + 2.0 * kernel_sig_t, side="left"))
    return lo, hi


def apply_windows(net, geom, P, tsteps_abs=None, t_win=6.0, step_size="half", min_required_picks=1, n_grids=1.0,
                  day_len=86400.0, device=None, embed=None):
    """Run `net.forward_fixed_source` over every kept window and stack the query read-out into `Out_2[Q, len(tsteps_abs)]`.

    net: genie_amd.module.GCN_Detection_Network_extended with adjacencies set; geom: genie_amd.synthetic.Geometry (locs,
    x_grid, x_query, travel times); P: picks [n, 5] (t, station, amp, prob, phase). Returns (Out_2 on device, times used).
    """
    dev = device or next(net.parameters()).device
    max_t = geom.max_t
    tsteps, offsets, step, n_overlap, dt_win = window_schedule(P[:, 0], max_t, day_len, t_win, 9, step_size)
    if tsteps_abs is None:
        tsteps_abs = np.arange(tsteps.min() - t_win / 2.0, tsteps.max() + t_win / 2.0 + dt_win, dt_win)
    times = windows_with_enough_picks(P[:, 0], tsteps, max_t, t_win, min_required_picks)
    Out_2 = torch.zeros((geom.x_query.shape[0], len(tsteps_abs)), dtype=torch.float32, device=dev)
    locs = torch.from_numpy(geom.locs).float().to(dev)
    xg = torch.from_numpy(geom.x_grid).float().to(dev)
    xq = torch.from_numpy(geom.x_query).float().to(dev)
    tq = torch.from_numpy(offsets.reshape(-1, 1)).float().to(dev)
    embed = embed or (lambda picks, t0: synthetic.make_slice_mask(geom, picks, t0))
    drop_last = step_size == "half"
    asc = is_ascending(tsteps_abs)
    used = []
    with torch.no_grad():
        for t0 in times:
            sel = (P[:, 0] > t0 - 2.0 * synthetic.KERNEL_SIG_T) & (P[:, 0] < t0 + max_t + 2.0 * synthetic.KERNEL_SIG_T)  # process_utils.py:476
            if not sel.any():
                continue                                                      # process_continuous_days.py:792-793
            used.append(t0)
            Slice, Mask = embed(P[sel], t0)
            if not getattr(net, "use_phase_types", True):        # process_continuous_days.py:783-786
                Slice, Mask = np.array(Slice, copy=True), np.array(Mask, copy=True)
                Slice[:, 2:] = 0.0
                Mask[:, 2:] = 0.0
            y, x = net.forward_fixed_source(torch.from_numpy(Slice).to(dev), torch.from_numpy(Mask).to(dev), None, None, None,
                                            locs, xg, xq, tq)
            cols, keep = window_columns(tsteps_abs, t0, offsets, drop_last, asc)
            Out_2.index_add_(1, torch.from_numpy(cols).to(dev), x[:, torch.from_numpy(keep).to(dev), 0] / (n_overlap * n_grids))
    return Out_2, np.asarray(used)


def apply_windows_device(net, geom, P, trv_times, tsteps_abs=None, t_win=6.0, step_size="half", min_required_picks=1,
                         n_grids=1.0, day_len=86400.0, kernel_sig_t=synthetic.KERNEL_SIG_T, dt_embed=None, max_t=None,
                         times=None, tail_batch=16, pairs=None):
    """GPU-only apply loop: `P` [n,5] (t, station index in the model's station order, amp, prob, phase) sorted by time,
    `trv_times` [G, S, 2] theoretical travel times; `pairs` [2, N] (station, source) = the product nodes of a `use_subgraph` model
    (`A_src_in_sta`), whose Slice / Mask rows follow that list. Returns (Out_2 on device, window start times used). `tail_batch`: windows
    per G-sized tail (1..16; 16 is the default of the bench and measured best with the device embedding in the loop, bench.py --mode
    stream: the tail kernels are latency-bound, their fixed costs are paid once per batch)."""
    hp = net._hip
    sharded = getattr(net, "is_sharded", False)       # source-node-sharded model: this rank embeds and runs its owned + halo rows only;
    net.window_batch = 1 if sharded else tail_batch   # the tail follows the shard's all-gather, one per window (module.py docstring)
    dev = hp.device
    max_t = float(max_t if max_t is not None else np.ceil(trv_times.max() + 1.0))
    dt_embed = float(dt_embed if dt_embed is not None else np.round(kernel_sig_t / 10.0, 2))      # process_continuous_days.py:608
    tsteps, offsets, step, n_overlap, dt_win = window_schedule(P[:, 0], max_t, day_len, t_win, 9, step_size)
    if tsteps_abs is None:
        tsteps_abs = np.arange(tsteps.min() - t_win / 2.0, tsteps.max() + t_win / 2.0 + dt_win, dt_win)
    if times is None:
        times = windows_with_enough_picks(P[:, 0], tsteps, max_t, t_win, min_required_picks)
    times = np.asarray(times, dtype=np.float64)
    order = np.argsort(P[:, 0], kind="stable")
    Ps = P[order]
    d_t = torch.from_numpy(Ps[:, 0].copy()).to(dev)
    d_sta = torch.from_numpy(Ps[:, 1].astype(np.int32)).to(dev)
    ph = Ps[:, 4].astype(np.int32)
    if not getattr(net, "use_phase_types", True):        # process_continuous_days.py:562-563 (the embedding zeroes columns 2, 3: :783-786)
        ph = np.zeros_like(ph)
    d_ph = torch.from_numpy(ph).to(dev)
    if pairs is not None:        # process_utils.py:605 `trv_times[src, ind_use[sta], :]` per listed product node
        pairs = np.asarray(pairs)
        d_trv = torch.from_numpy(np.ascontiguousarray(np.asarray(trv_times, dtype=np.float32)[pairs[1], pairs[0]])).to(dev)
    else:
        d_trv = net.node_rows(np.asarray(trv_times, dtype=np.float32), 2)          # (a shard: its own rows, cut out on the host)
    Out_2 = torch.zeros((geom.x_query.shape[0], len(tsteps_abs)), dtype=torch.float32, device=dev)
    locs = torch.from_numpy(geom.locs).float().to(dev)
    xg = torch.from_numpy(geom.x_grid).float().to(dev)
    xq = torch.from_numpy(geom.x_query).float().to(dev)
    tq = torch.from_numpy(offsets.reshape(-1, 1)).float().to(dev)
    drop_last = step_size == "half"
    # per-window pick ranges and Out_2 column indices are host arithmetic on tiny arrays, done up front
    lo = np.searchsorted(Ps[:, 0], times - 2.0 * kernel_sig_t, side="right")                      # strict >, process_utils.py:476
    hi = np.searchsorted(Ps[:, 0], times + max_t + 2.0 * kernel_sig_t, side="left")               # strict <
    nonempty = hi > lo                                                                             # process_continuous_days.py:792-793
    times, lo, hi = times[nonempty], lo[nonempty], hi[nonempty]
    asc = is_ascending(tsteps_abs)
    wc = [window_columns(tsteps_abs, t0, offsets, drop_last, asc) for t0 in times]
    n_off = len(offsets) - (1 if drop_last else 0)
    if all(len(k) == n_off for _, k in wc):            # the usual case: no duplicate column inside a window
        cols = torch.from_numpy(np.stack([c_ for c_, _ in wc]) if wc else np.zeros((0, n_off), dtype=np.int64)).to(dev)
        keeps = None
    else:
        cols = [torch.from_numpy(c_).to(dev) for c_, _ in wc]
        keeps = [torch.from_numpy(k_).to(dev) for _, k_ in wc]
    acc_done = [None]

    def window_vals(xw, w):             # xw [Q, T, 1] of window w -> the kept offsets
        if keeps is not None:
            return xw[:, keeps[w], 0]
        return xw[:, :-1, 0] if drop_last else xw[:, :, 0]

    def flush(first):
        # tail + read-outs of the pushed windows in one set of launches on a side stream; the accumulation into Out_2 follows
        # on that stream, window by window and batch by batch in order (overlapping columns: a fixed summation order)
        y, x, _ = net.flush_windows(xg, xq, tq)
        with torch.cuda.stream(hp.side_stream):
            if acc_done[0] is not None:
                hp.side_stream.wait_event(acc_done[0])
            for k in range(x.shape[0]):
                Out_2.index_add_(1, cols[first + k], window_vals(x[k], first + k) / (n_overlap * n_grids))
            acc_done[0] = torch.cuda.Event()
            acc_done[0].record(hp.side_stream)

    with torch.no_grad():
        first = 0
        for w, t0 in enumerate(times):
            a, b = int(lo[w]), int(hi[w])
            Slice, Mask = net.embed_window(d_t[a:b], d_sta[a:b], d_ph[a:b], float(t0), max_t, kernel_sig_t, dt_embed, d_trv,
                                           presplit=True)    # the push below is the only consumer of (Slice, Mask)
            if net.window_batch == 1:     # one tail per window (forward_fixed_source_pipelined), accumulated in window order
                y, x, _ = net.forward_fixed_source_pipelined(Slice, Mask, None, None, None, locs, xg, xq, tq)
                with torch.cuda.stream(hp.side_stream):
                    if acc_done[0] is not None:
                        hp.side_stream.wait_event(acc_done[0])
                    Out_2.index_add_(1, cols[w], window_vals(x, w) / (n_overlap * n_grids))
                    acc_done[0] = torch.cuda.Event()
                    acc_done[0].record(hp.side_stream)
            elif net.push_window(Slice, Mask) == net.window_batch or w == len(times) - 1:
                flush(first)
                first = w + 1
        hp.wait_tails()
    return Out_2, times


# ---- per-window pick lists (process_utils.py:644-699) and the two other per-day loops of the caller ------------------------------------

class ResidentPicks(object):
    """The picks of a day resident on one device, in the form every per-window step of the caller needs.

    `P` [n, 5] float64 (t, ABSOLUTE station index, amp, prob, phase) in the caller's order (`load_picks`, utils.py:983: the pick file's
    order, not a time order); `ind_use` = the stations of the model, positions into the absolute station set of `n_sta_all` entries
    (`perm_vec`, process_utils.py:486-487 / :678-679). Kept here: the picks of those stations (the reference drops the others per
    window, :480-483 and :684 -- the filter does not depend on the window), stable-sorted by time, so that
      * the picks a window's embedding reads, `t0 - 2 sigma < t < t0 + max_t + 2 sigma` (:476), are ONE contiguous range `embed_range`
        found by two binary searches on a host copy of the times -- the tensors handed to `genie_embed_window` are views, and
      * the lists `forward_fixed` consumes (`extract_pick_inputs_from_data`, :644-699) are that range cut by the ball query (:665) and
        stable-sorted by station on the device: `np.lexsort((times, indices))` (:690) orders by station, then time, ties in the
        caller's order, which a stable sort by station of a stable time order reproduces exactly.
    Works on CPU tensors too (the CPU tests pin it to the reference's fixtures); nothing here touches the HIP library."""

    def __init__(self, P, ind_use, n_sta_all, device, use_phase_types=True):
        P = np.asarray(P, dtype=np.float64)
        ind_use = np.asarray(ind_use).astype(np.int64)
        perm_vec = -np.ones(int(n_sta_all), dtype=np.int64)
        perm_vec[ind_use] = np.arange(len(ind_use))
        sta = perm_vec[P[:, 1].astype(np.int64)]
        keep = np.nonzero(sta > -1)[0]
        keep = keep[np.argsort(P[keep, 0], kind="stable")]
        self.P = P
        self.index_host = keep                                   # rows of the caller's P, in this object's (time) order
        self.t_host = np.ascontiguousarray(P[keep, 0])
        self.device = torch.device(device)
        self.n_sta = int(len(ind_use))
        ph = P[keep, 4].copy()
        if not use_phase_types:                                  # process_continuous_days.py:562-563
            ph[:] = 0.0
        self.t = torch.from_numpy(self.t_host).to(self.device)
        self.sta = torch.from_numpy(sta[keep].astype(np.int32)).to(self.device)
        self.phase = torch.from_numpy(ph.astype(np.int32)).to(self.device)
        self.phase_f = torch.from_numpy(ph).to(self.device)
        self.index = torch.from_numpy(keep).to(self.device)

    def __len__(self):
        return int(self.t_host.shape[0])

    def embed_range(self, t0, max_t, kernel_sig_t):
        """[lo, hi) of the picks with `t0 - 2 sigma < t < t0 + max_t + 2 sigma` (strict, process_utils.py:476)."""
        return picks_in_embed_range(self.t_host, float(t0), float(max_t), float(kernel_sig_t))

    def embed_args(self, t0, max_t, kernel_sig_t):
        """(pick_t float64, pick_sta int32, pick_phase int32) views for `HipPath.embed_window`, or None for an empty range."""
        lo, hi = self.embed_range(t0, max_t, kernel_sig_t)
        if hi <= lo:
            return None
        return self.t[lo:hi], self.sta[lo:hi], self.phase[lo:hi]

    def pick_inputs(self, t0, max_t, kernel_sig_t, t_win=10.0):
        """`lp_times, lp_stations, lp_phases` of `extract_input_from_data(...)[1]` for the window starting at `t0`
        (process_utils.py:637 -> :644-699) as device tensors (float64 [m], int64 [m], float64 [m]) plus `index` int64 [m]: the rows of
        the caller's `P` they come from (`lp_meta = P[index]`). The ball query of :665 keeps `|t - (t0 + max_t / 2)| <= t_win + max_t / 2`
        of the window's slice; with `2 sigma <= t_win` that is all of it."""
        lo, hi = self.embed_range(t0, max_t, kernel_sig_t)
        t, sta, ph, idx = self.t[lo:hi], self.sta[lo:hi], self.phase_f[lo:hi], self.index[lo:hi]
        if 2.0 * float(kernel_sig_t) > float(t_win) and hi > lo:
            inside = (t - (float(t0) + float(max_t) / 2.0)).abs() <= (float(t_win) + float(max_t) / 2.0)
            t, sta, ph, idx = t[inside], sta[inside], ph[inside], idx[inside]
        order = torch.sort(sta, stable=True)[1]
        return t[order] - float(t0), sta[order].long(), ph[order], idx[order]

    def meta(self, index):
        """`lp_meta` rows (host float64 [m, 5]) of a `pick_inputs` index."""
        return self.P[index.cpu().numpy()]


class GridLeg(object):
    """One source grid of the per-day loops (`x_grid_ind` of process_continuous_days.py:770, :950, :1020): the model whose adjacencies were
    set on that grid, the grid's Cartesian node positions and its travel-time table `x_grids_trv[x_grid_ind]` [G, S, 2] resident on the
    device ([N, 2] rows per listed product node for a `use_subgraph` model: `pairs` [2, N] = `A_src_in_sta`)."""

    def __init__(self, net, x_grid_cart, trv_times, pairs=None, ind_use=None):
        """`trv_times` [G, S', 2]: with `ind_use` the reference's table over ALL stations (`compute_travel_times(trv, locs, ...)`), cut to
        the model's stations here as process_utils.py:599 does (`trv_times[:, ind_use]`); without it the table must already be restricted
        to the model's stations, in the model's station order (its station axis is checked against the model's station count)."""
        self.net = net
        hp = net._hip
        if hp is None:
            raise RuntimeError("GridLeg: call net.set_adjacencies*(...) first")
        self.device = hp.device
        self.x_grid_cart = torch.as_tensor(x_grid_cart).float().to(self.device)
        trv_times = np.asarray(trv_times, dtype=np.float32)
        if ind_use is not None:
            trv_times = trv_times[:, np.asarray(ind_use).astype(np.int64)]
        n_sta = net._shard.n_sta if getattr(net, "_shard", None) is not None else hp.n_sta
        if trv_times.ndim != 3 or trv_times.shape[1] != n_sta or trv_times.shape[2] != 2:
            raise ValueError("GridLeg: trv_times must be [G, %d, 2] over the model's stations (pass ind_use= to cut the table of all "
                             "stations down, process_utils.py:599), got %s" % (n_sta, tuple(trv_times.shape)))
        if pairs is not None:
            pairs = np.asarray(pairs)
            self.trv = torch.from_numpy(np.ascontiguousarray(trv_times[pairs[1], pairs[0]]).reshape(-1, 2)).to(self.device)
        else:
            self.trv = net.node_rows(trv_times, 2)

    def check(self):
        """Raise what the device-side checks of the calls COMPLETED so far found (`HipPath.check_input_range` / `check_index_flags`);
        the per-day loops call it after their final copy to the host, which has waited for every window."""
        net = self.net
        for hp in ((net._hip,) if getattr(net, "_shard", None) is None else (net._shard.local, net._shard.full)):
            hp.check_input_range()

    def embed(self, picks, t0, max_t, kernel_sig_t, dt):
        """(Slice, Mask) of the window starting at t0 (genie_embed_window = extract_input_from_data, process_utils.py:460-642), or None
        when no pick falls in the embedding range."""
        args = picks.embed_args(t0, max_t, kernel_sig_t)
        if args is None:
            return None
        return self.net.embed_window(args[0], args[1], args[2], float(t0), float(max_t), float(kernel_sig_t), float(dt), self.trv)


def _dt_embed(kernel_sig_t, dt_embed):
    return float(dt_embed if dt_embed is not None else np.round(kernel_sig_t / 10.0, 2))          # process_continuous_days.py:608


_PINNED = {}


def _pinned_pair(n, device):
    """Two page-locked float64 [n, 3] staging buffers per (size, device, calling thread), kept for the life of the process (pinning
    2.7 MB costs ~20 ms: not per call). Keyed so that concurrent refine passes -- one per GPU, or one per thread -- never share a pair."""
    import threading
    key = (int(n), str(device), threading.get_ident())
    if key not in _PINNED:
        _PINNED[key] = [torch.empty((n, 3), dtype=torch.float64).pin_memory() for _ in range(2)]
    return _PINNED[key]


def refine_sources(legs, picks, srcs, locs_cart, tq, max_t, X_offset_min, X_offset_range, n_rand_query, ftrns1, ftrns2,
                   lat_range, lon_range, depth_range, kernel_sig_t=synthetic.KERNEL_SIG_T, dt_embed=None, rand=None, ftrns2_device=None):
    """The refine pass of the caller (process_continuous_days.py:926-980) on the device: for every candidate source `srcs[i]` = (lat, lon,
    depth, origin time, value) a cloud of `n_rand_query` random queries around it (`ftrns1(src) + rand(n, 3) * X_offset_range +
    X_offset_min`, kept where `ftrns2` of it lies strictly inside the three ranges, :929-936), one `forward_fixed_source` per grid
    leg on the window that STARTS at the source's origin time, read out at those queries (:971-972; the kNN of the cloud into the grid
    by `genie_knn`), the clouds' outputs averaged over the legs, and the refined source = the query and time offset of the maximum
    (`argmax` of the row maxima, then of that row, :976-978: first maximum in both). Sources whose window holds no pick keep an all-zero
    read-out (:966-967), i.e. their first query and `tq[0]`. Returns (srcs_refined float64 [n, 5] sorted by origin time (:981-982),
    `order` = that sort's permutation of the input rows). `rand(n, 3)` defaults to `np.random.rand` (the reference's draw); the
    per-source results stay on the device until one copy at the end. `ftrns2_device`: the inverse transform as a function of a float64
    GPU tensor [n, 3] (the reference also carries torch forms of its transforms, `ftrns2_diff`): the cloud's arithmetic, the region
    filter and the float32 rounding then run on the device in float64 -- the same values as the numpy path, whose 112 000 x 3 float64
    temporaries per source otherwise make the pass host-bound (37 ms per source at config 2 against ~6 ms of GPU work) -- and only the
    refined source's own query is transformed back on the host."""
    rand = rand or np.random.rand
    srcs = np.asarray(srcs, dtype=np.float64)
    tq_host = np.asarray(tq.detach().cpu() if torch.is_tensor(tq) else tq, dtype=np.float64).reshape(-1)
    dev = legs[0].device
    tq_d = torch.as_tensor(tq_host.reshape(-1, 1)).float().to(dev)
    locs_d = torch.as_tensor(locs_cart).float().to(dev)
    dt = _dt_embed(kernel_sig_t, dt_embed)
    n_scale = float(len(legs))
    clouds, found = [], []
    on_device = ftrns2_device is not None
    if on_device:      # constants of the loop and every source's Cartesian position: copied once
        off_rng_d = torch.as_tensor(np.asarray(X_offset_range, dtype=np.float64).reshape(1, 3), device=dev)
        off_min_d = torch.as_tensor(np.asarray(X_offset_min, dtype=np.float64).reshape(1, 3), device=dev)
        src_cart_d = torch.from_numpy(np.ascontiguousarray(ftrns1(srcs[:, 0:3]), dtype=np.float64)).to(dev) if srcs.shape[0] else None
        ninf = torch.full((), float("-inf"), dtype=torch.float32, device=dev)
        stage, stage_ev = _pinned_pair(n_rand_query, dev), [None, None]
    with torch.no_grad():
        for i in range(srcs.shape[0]):
            if on_device:
                # Nothing in this branch waits for the device (round 5, tools/sync_probe_day.py: eight waits per source before -- pageable
                # copies, the boolean-mask compaction, three tensor-indexed reads): the draw goes through pinned memory, the queries outside
                # the region stay in the cloud and are masked out of the argmax (a query's read-out depends on no other query: the same
                # values and the same refined query as after the reference's compaction), and the refined query's row is gathered on the
                # device. The host draws source i + 1's cloud while the GPU works on source i.
                k = i % 2
                if stage_ev[k] is not None:
                    stage_ev[k].synchronize()              # the copy that last read this staging buffer (two sources ago) has finished
                stage[k].numpy()[...] = rand(n_rand_query, 3)                                                         # the host's draw, float64
                r = stage[k].to(dev, non_blocking=True)
                stage_ev[k] = torch.cuda.Event()
                stage_ev[k].record(torch.cuda.current_stream(dev))       # the stream of `dev` the copy was issued on
                Xc_d = src_cart_d[i:i + 1] + (r * off_rng_d + off_min_d)                                                  # :929
                X1_d = ftrns2_device(Xc_d)
                keep = ((X1_d[:, 0] > lat_range[0]) & (X1_d[:, 0] < lat_range[1]) & (X1_d[:, 1] > lon_range[0]) & (X1_d[:, 1] < lon_range[1])
                        & (X1_d[:, 2] > depth_range[0]) & (X1_d[:, 2] < depth_range[1]))
                xq = Xc_d.float()
            else:
                Xc = ftrns1(srcs[i, 0:3].reshape(1, -1)) + (rand(n_rand_query, 3) * X_offset_range + X_offset_min)      # :929
                X1 = ftrns2(Xc)
                inside = np.where((X1[:, 0] > lat_range[0]) * (X1[:, 0] < lat_range[1]) * (X1[:, 1] > lon_range[0]) * (X1[:, 1] < lon_range[1])
                                  * (X1[:, 2] > depth_range[0]) * (X1[:, 2] < depth_range[1]))[0]
                X1, Xc = X1[inside], Xc[inside]
                clouds.append(X1)
                xq = torch.from_numpy(np.ascontiguousarray(Xc)).to(dev).float()                                       # torch.Tensor(...) :934 (rounded on the device)
            acc = torch.zeros((xq.shape[0], tq_host.shape[0]), dtype=torch.float32, device=dev)
            if xq.shape[0]:
                for leg in legs:
                    em = leg.embed(picks, srcs[i, 3], max_t, kernel_sig_t, dt)
                    if em is None:
                        continue                                                                                        # :966-967
                    _, x = leg.net.forward_fixed_source(em[0], em[1], None, None, None, locs_d, leg.x_grid_cart, xq, tq_d)
                    acc += x[:, :, 0] / n_scale                                                                         # :972
            if on_device:
                accm = torch.where(keep.view(-1, 1), acc, ninf)
                ip = torch.argmax(accm.max(1)[0]).view(1)                                                               # :976 (first maximum)
                row = accm.index_select(0, ip)[0]
                it = torch.argmax(row).view(1)                                                                          # :977
                found.append(torch.cat((ip.double(), it.double(), row.index_select(0, it).double(), keep.any().double().view(1),
                                        Xc_d.index_select(0, ip).view(3))))
            elif xq.shape[0]:
                ip = torch.argmax(acc.max(1)[0])
                it = torch.argmax(acc[ip])
                found.append(torch.stack((ip.double(), it.double(), acc[ip, it].double())))
            else:
                found.append(torch.full((3,), float("nan"), dtype=torch.float64, device=dev))
    found = torch.stack(found).cpu().numpy() if found else np.zeros((0, 7 if on_device else 3))
    for leg in legs:       # the copy above waited for the device: the verdicts of every window of this pass are in
        leg.check()
    out = np.zeros((srcs.shape[0], 5))
    for i in range(srcs.shape[0]):
        if (found[i, 3] == 0.0) if on_device else (clouds[i].shape[0] == 0):
            raise ValueError("refine_sources: no query of source %d lies inside the region (the reference's argmax raises here too)" % i)
        ip, it = int(found[i, 0]), int(found[i, 1])
        out[i, 0:3] = ftrns2(found[i, 4:7].reshape(1, 3))[0] if on_device else clouds[i][ip]
        out[i, 3] = srcs[i, 3] + tq_host[it]
        out[i, 4] = found[i, 2]
    order = np.argsort(out[:, 3])
    return out[order], order


def associate_sources(legs, picks, srcs_refined, locs_cart, tq, max_t, trv_out_srcs, ftrns1, x_save, kernel_sig_t=synthetic.KERNEL_SIG_T,
                      dt_embed=None, t_win=10.0):
    """The association pass of the caller (process_continuous_days.py:1020-1065) on the device: for every refined source one 4-output
    `forward_fixed` per grid leg on the window that starts at its origin time, with the window's pick lists (`ResidentPicks.pick_inputs`
    = extract_pick_inputs_from_data), ONE spatial query `x_save` (lat, lon of the first node of the reference's coarse map; its depth
    replaced by the source's, :1046) and the source itself as the only candidate (`x_query_src = ftrns1(src)`, `tq_sample = 0`,
    `trv_out_q = trv_out_srcs[[i]]` [1, S, 2], :1052). Returns (Out_p_save, Out_s_save: lists of float32 device tensors [m_i] = the
    P / S association likelihood of every pick of the window, averaged over the legs (:1054-1055); Save_picks: list of host [m_i, 2]
    (relative time, station, :1042); lp_meta: list of host [m_i, 5]). A window without picks yields empty entries (:1049-1050)."""
    srcs = np.asarray(srcs_refined, dtype=np.float64)
    dev = legs[0].device
    tq_d = torch.as_tensor(np.asarray(tq.detach().cpu() if torch.is_tensor(tq) else tq, dtype=np.float32).reshape(-1, 1)).to(dev)
    locs_d = torch.as_tensor(locs_cart).float().to(dev)
    trv_out_srcs = torch.as_tensor(trv_out_srcs).float().to(dev)
    dt = _dt_embed(kernel_sig_t, dt_embed)
    n_scale = float(len(legs))
    zero = torch.zeros(1, device=dev)
    x_save = np.array(x_save, dtype=np.float64).reshape(1, 3)
    Out_p, Out_s, Save_picks, lp_meta = [], [], [], []
    # the per-source positions of the loop, copied once (a pageable host-to-device copy per source made the host wait for the device)
    xs_all = np.repeat(x_save, max(srcs.shape[0], 1), axis=0)
    xs_all[: srcs.shape[0], 2] = srcs[:, 2]                                                                          # :1046
    xs_cart_all = torch.from_numpy(np.ascontiguousarray(ftrns1(xs_all))).float().to(dev)
    src_cart_all = torch.from_numpy(np.ascontiguousarray(ftrns1(srcs[:, 0:3]))).float().to(dev) if srcs.shape[0] else None
    with torch.no_grad():
        for i in range(srcs.shape[0]):
            tp, ip, ph, idx = picks.pick_inputs(srcs[i, 3], max_t, kernel_sig_t, t_win)
            Save_picks.append((tp, ip))
            lp_meta.append(idx)
            acc_p = torch.zeros(tp.shape[0], dtype=torch.float32, device=dev)
            acc_s = torch.zeros(tp.shape[0], dtype=torch.float32, device=dev)
            Out_p.append(acc_p)
            Out_s.append(acc_s)
            if tp.shape[0] == 0:
                continue                                                                                                # :1049-1050
            xs_cart, src_cart = xs_cart_all[i:i + 1], src_cart_all[i:i + 1]
            tpf, phf = tp.float(), ph.long().float().reshape(-1, 1)
            for leg in legs:
                em = leg.embed(picks, srcs[i, 3], max_t, kernel_sig_t, dt)
                if em is None:
                    continue
                out = leg.net.forward_fixed(em[0], em[1], tpf, ip, phf, locs_d, leg.x_grid_cart, xs_cart, src_cart, tq_d, zero,
                                            trv_out_srcs[i:i + 1])                                                      # :1052
                acc_p += out[2][0, :, 0] / n_scale                                                                      # :1054
                acc_s += out[3][0, :, 0] / n_scale                                                                      # :1055
    Save_picks = [np.stack((a.cpu().numpy(), b.cpu().numpy().astype(np.float64)), axis=1) for a, b in Save_picks]
    lp_meta = [picks.meta(ix) for ix in lp_meta]
    if srcs.shape[0]:
        torch.cuda.current_stream(dev).synchronize()      # (windows without picks copy nothing back)
    for leg in legs:       # device-side verdicts (a pick outside the time-pointer table, a station index outside the model) of every call above
        leg.check()
    return Out_p, Out_s, Save_picks, lp_meta


def retained_after_marching(srcs_refined, ftrns1, tc_win, sp_win, scale_depth_clustering=0.2, scale_time_ref=3500.0):
    """The second LocalMarching of the caller and the match back to the refined list (process_continuous_days.py:1072-1085): the
    refined sources that survive it (`n_steps_max = 2, use_directed = False`), found as the nearest refined source of each survivor in
    (Cartesian position, `scale_time_ref` * origin time), `np.unique`d. Returns the retained row indices (ascending)."""
    from scipy.spatial import cKDTree
    from . import postproc
    srcs_refined = np.asarray(srcs_refined, dtype=np.float64)
    if len(srcs_refined) == 0:
        return np.zeros(0, dtype=np.int64)
    kept = postproc.local_marching(srcs_refined, ftrns1, tc_win=tc_win, sp_win=sp_win, scale_depth=scale_depth_clustering, n_steps_max=2,
                                   use_directed=False)
    tree = cKDTree(np.concatenate((ftrns1(srcs_refined), scale_time_ref * srcs_refined[:, [3]]), axis=1))
    return np.unique(tree.query(np.concatenate((ftrns1(kept), scale_time_ref * kept[:, [3]]), axis=1))[1])


def detect_refine_associate(legs, picks, Out_2, X_query, tsteps_abs, locs, trv, tq, max_t, ftrns1, ftrns2, lat_range, lon_range,
                            depth_range, X_offset_min, X_offset_range, n_rand_query, thresh, src_t_kernel, dt_win, break_win, tc_win,
                            sp_win, scale_depth_clustering=0.2, kernel_sig_t=synthetic.KERNEL_SIG_T, dt_embed=None, t_win_assoc=10.0,
                            rand=None, ftrns2_device=None):
    """Everything the caller does between the apply loop and the competitive assignment, with the network calls on the device
    (process_continuous_days.py:811-1105): peaks of the device-resident `Out_2` -> time groups -> LocalMarching (`postproc.
    detect_sources`, :811-891), the refine pass (`refine_sources`, :926-982), travel times of the refined sources (`trv(locs, srcs)`
    [n, S, 2], :1004), the association pass (`associate_sources`, :1006-1068: `X_save` = the first node of the 15 x 15 map of the region,
    :1008-1014), the second LocalMarching with its match back (`retained_after_marching`, :1072-1090), the travel times of the retained
    sources (:1092) and the final sort by origin time (:1097-1105). `locs` [S, 3] (lat, lon, depth) of the stations in use; `trv`: the
    travel-time callable of the reference, (float tensor [S, 3], float tensor [n, 3]) -> [n, S, 2]. Returns a dict: `srcs` (after the
    first marching), `srcs_refined` [m, 5], `trv_out_srcs` (device [m, S, 2]), `Out_p_save`, `Out_s_save` (lists of device tensors),
    `Save_picks`, `lp_meta` (lists of host arrays) -- the inputs of `competitive_assignment`, which is out of scope."""
    from . import postproc
    dev = legs[0].device
    empty = {"srcs": np.zeros((0, 5)), "srcs_refined": np.zeros((0, 5)), "trv_out_srcs": None, "Out_p_save": [], "Out_s_save": [],
             "Save_picks": [], "lp_meta": []}
    srcs = postproc.detect_sources(Out_2, X_query, tsteps_abs, ftrns1, thresh, src_t_kernel, dt_win, break_win, tc_win, sp_win,
                                   scale_depth_clustering)
    if len(srcs) == 0:
        return empty                                                                                                    # :886-888
    locs = np.asarray(locs, dtype=np.float64)
    locs_cart = ftrns1(locs)
    locs_d = torch.as_tensor(locs).float().to(dev)
    srcs_refined, _ = refine_sources(legs, picks, srcs, locs_cart, tq, max_t, X_offset_min, X_offset_range, n_rand_query, ftrns1, ftrns2,
                                     lat_range, lon_range, depth_range, kernel_sig_t, dt_embed, rand, ftrns2_device)
    with torch.no_grad():
        trv_out = trv(locs_d, torch.as_tensor(srcs_refined[:, 0:3]).float().to(dev)).detach()                           # :1004
    x_save = np.array([lat_range[0], lon_range[0], 0.0])                              # xx[0] of the meshgrid of :1008-1014
    Out_p, Out_s, Save_picks, lp_meta = associate_sources(legs, picks, srcs_refined, locs_cart, tq, max_t, trv_out, ftrns1, x_save,
                                                          kernel_sig_t, dt_embed, t_win_assoc)
    keep = retained_after_marching(srcs_refined, ftrns1, tc_win, sp_win, scale_depth_clustering)
    srcs_kept = srcs_refined[keep]
    with torch.no_grad():
        trv_kept = trv(locs_d, torch.as_tensor(srcs_kept[:, 0:3]).float().to(dev)).detach()                             # :1092
    order = np.argsort(srcs_kept[:, 3])                                                                                 # :1097
    pick = [int(keep[j]) for j in order]
    return {"srcs": srcs, "srcs_refined": srcs_kept[order], "trv_out_srcs": trv_kept[torch.as_tensor(order, device=trv_kept.device)],
            "Out_p_save": [Out_p[j] for j in pick], "Out_s_save": [Out_s[j] for j in pick],
            "Save_picks": [Save_picks[j] for j in pick], "lp_meta": [lp_meta[j] for j in pick]}
