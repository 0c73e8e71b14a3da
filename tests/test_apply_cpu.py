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
