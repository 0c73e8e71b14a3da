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
    net.window_batch = tail_batch
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
        d_trv = torch.from_numpy(np.ascontiguousarray(trv_times, dtype=np.float32).reshape(-1, 2)).to(dev)
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
            Slice, Mask = hp.embed_window(d_t[a:b], d_sta[a:b], d_ph[a:b], float(t0), max_t, kernel_sig_t, dt_embed, d_trv,
                                          presplit=True)     # the push below is the only consumer of (Slice, Mask)
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
