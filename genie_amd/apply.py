"""Sliding-window apply loop around `forward_fixed_source` — the caller of the hot path
(`/root/reference/Code/process_continuous_days.py:761-810`, setup `:357-381, :571, :725-754`).

Semantics kept from the reference:
* prediction window of `n_resolution = 9` origin-time offsets `arange(-t_win/2, t_win/2 + dt_win, dt_win)` (`:357-362`);
* window starts `tsteps = arange(max(0, min(pick_t) - max_t), min(day_len, max(pick_t)), step)`, `step` and `n_overlap`
  from `step_size` in {'full', 'partial', 'half'} (`:367-381, :571`);
* windows with fewer than `min_required_picks` picks in `[t0 - t_win, t0 + max_t + t_win]` are skipped (`:725-741`);
* `Out_2[:, idx(t0 + offsets)] += x[:, :, 0] / n_overlap / n_grids`, dropping the last offset when `step_size == 'half'`
  (`:802-805`).
Differences by design: `Out_2` stays on the GPU and is accumulated with `index_add_` (the reference copies every window's
output to the host, `:803-805`); nothing in the loop synchronises with the host.

The pick -> `Slice/Mask` embedding is still a host (numpy) step here (`genie_amd.synthetic.make_slice_mask`, the exact
nearest-pick semantics of `process_utils.py:262-275`); moving it on device is row f-1 of SURVEY.md section 8.
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
    with torch.no_grad():
        for t0 in times:
            sel = (P[:, 0] > t0 - 2.0 * synthetic.KERNEL_SIG_T) & (P[:, 0] < t0 + max_t + 2.0 * synthetic.KERNEL_SIG_T)  # process_utils.py:476
            Slice, Mask = embed(P[sel], t0)
            y, x = net.forward_fixed_source(torch.from_numpy(Slice).to(dev), torch.from_numpy(Mask).to(dev), None, None, None,
                                            locs, xg, xq, tq)
            ip = np.abs(tsteps_abs.reshape(-1, 1) - (t0 + offsets).reshape(1, -1)).argmin(0)   # nearest index, tree_tsteps.query
            cols = torch.from_numpy(ip[:-1] if drop_last else ip).to(dev)
            vals = x[:, :-1, 0] if drop_last else x[:, :, 0]
            Out_2.index_add_(1, cols, vals / (n_overlap * n_grids))
    return Out_2, times
