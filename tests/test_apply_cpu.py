"""CPU: host logic of the apply loop (window schedule, quiescent-window skipping, Out_2 stacking) with a stub model."""
import numpy as np
import torch

from genie_amd import apply, synthetic


class _StubNet(torch.nn.Module):
    """Returns a known pattern so the stacking arithmetic can be checked without a GPU."""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.calls = 0

    def forward_fixed_source(self, Slice, Mask, tpick, ipick, phase, locs, xg, xq, tq):
        self.calls += 1
        Q, T = xq.shape[0], tq.shape[0]
        x = torch.ones(Q, T, 1) * torch.arange(1, T + 1).view(1, T, 1).float()
        return torch.zeros(xg.shape[0], T, 1), x


def test_window_schedule_matches_reference_constants():
    ts, off, step, n_overlap, dt_win = apply.window_schedule(np.array([100.0, 200.0]), max_t=50.0, t_win=6.0, step_size="half")
    assert np.isclose(dt_win, 0.75) and np.isclose(step, 3.0) and n_overlap == 2.0
    assert len(off) == 9 and np.isclose(off[0], -3.0) and np.isclose(off[-1], 3.0)
    assert np.isclose(ts[0], 50.0) and ts[-1] < 200.0
    _, _, step_f, n_f, _ = apply.window_schedule(np.array([100.0, 200.0]), 50.0, step_size="full")
    assert np.isclose(step_f, 6.75) and n_f == 1.0


def test_quiescent_windows_are_skipped():
    t = np.array([1000.0, 1000.5, 1001.0])
    ts = np.arange(0.0, 3000.0, 3.0)
    keep = apply.windows_with_enough_picks(t, ts, max_t=60.0, t_win=6.0, min_required_picks=3)
    assert keep.size > 0 and keep.min() >= 1000.0 - 60.0 - 6.0 - 3.0 and keep.max() <= 1001.0 + 6.0
    assert apply.windows_with_enough_picks(t, ts, 60.0, 6.0, 4).size == 0


def test_out2_stacking_half_step_is_uniform_in_the_interior():
    geom = synthetic.Geometry(6, 30, L=50e3, n_query=4, seed=3)
    P = synthetic.make_picks(geom, 80, seed=4)
    P[:, 0] += 500.0
    net = _StubNet()
    Out_2, times = apply.apply_windows(net, geom, P, step_size="half", min_required_picks=1, device="cpu",
                                       embed=lambda picks, t0: (np.zeros((180, 4), np.float32), np.zeros((180, 4), np.float32)))
    assert net.calls == len(times) > 3
    o = Out_2[0].numpy()
    # with step = 4*dt_win and the last of 9 offsets dropped, interior columns receive exactly two windows' values / 2
    interior = o[12:-12]
    assert np.all(interior > 0)
    assert np.allclose(Out_2[0], Out_2[-1])


def test_window_columns_snap_to_the_output_axis_and_write_duplicates_once():
    """process_continuous_days.py:766,797-805: the window start is snapped to `tsteps_abs` before the offsets are added, and
    numpy's fancy `+=` writes a column that two offsets map to only once (last occurrence)."""
    off = np.arange(-3.0, 3.75, 0.75)
    # aligned axis: nine distinct consecutive columns, the last dropped for 'half'
    ts_abs = np.arange(0.0, 60.0, 0.75)
    cols, keep = apply.window_columns(ts_abs, 30.0, off, drop_last=True)
    assert np.array_equal(cols, np.arange(36, 44)) and np.array_equal(keep, np.arange(8))
    # misaligned window start: snapped first (30.3 -> 30.0), so the same columns as above, not a shifted set
    cols2, _ = apply.window_columns(ts_abs, 30.3, off, drop_last=True)
    assert np.array_equal(cols2, cols)
    # a coarser output axis maps two offsets to one column: written once, by the later offset
    coarse = np.arange(0.0, 60.0, 1.5)
    cols3, keep3 = apply.window_columns(coarse, 30.0, off, drop_last=False)
    want = np.zeros(len(coarse))
    vals = np.arange(1.0, 10.0)
    ip = np.abs(coarse.reshape(-1, 1) - (30.0 + off).reshape(1, -1)).argmin(0)
    want[ip] += vals                                           # the reference's statement
    got = np.zeros(len(coarse))
    np.add.at(got, cols3, vals[keep3])
    assert len(np.unique(cols3)) == len(cols3) < 9 and np.array_equal(got, want)


def test_windows_without_picks_in_the_embedding_range_are_skipped():
    geom = synthetic.Geometry(6, 30, L=50e3, n_query=4, seed=3)
    P = synthetic.make_picks(geom, 40, seed=4)
    P[:, 0] += 500.0
    P2 = P.copy()
    P2[:, 0] += 2000.0                                          # a second burst far away: the windows in between see no pick
    P = np.concatenate([P, P2])
    net = _StubNet()
    tsteps = apply.window_schedule(P[:, 0], geom.max_t)[0]
    Out_2, used = apply.apply_windows(net, geom, P, step_size="half", min_required_picks=1, device="cpu",
                                      embed=lambda picks, t0: (np.zeros((180, 4), np.float32), np.zeros((180, 4), np.float32)))
    assert net.calls == len(used) < len(tsteps)
    for t0 in used:
        lo, hi = apply.picks_in_embed_range(np.sort(P[:, 0]), t0, geom.max_t, synthetic.KERNEL_SIG_T)
        assert hi > lo
