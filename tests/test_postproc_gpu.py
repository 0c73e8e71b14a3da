"""GPU: device half of the downstream reduction (genie_row_select_count / genie_row_select_fill through genie_amd.postproc):
`np.where(Out_2 > 0.01)` (process_continuous_days.py:812-813) and `scipy.signal.find_peaks(row, height, distance)` (:846) --
bit-exact index sets and values against numpy / scipy on the same matrix."""
import numpy as np
import pytest
import torch
from scipy.signal import find_peaks

from genie_amd import postproc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _matrix(rows, cols, seed, quantise=False):
    rng = np.random.default_rng(seed)
    x = (rng.random((rows, cols)) * (rng.random((rows, cols)) < 0.15)).astype(np.float32)
    if quantise:
        x = np.round(x * 8.0) / 8.0                       # many exact ties and flat tops
    if cols > 4:
        x[0, :3] = 0.9                                    # flat run touching the first sample: not a peak
        x[-1, -3:] = 0.9                                  # ... and the last sample
        x[rows // 2, :] = 0.0
    return x


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 2), (5, 3), (7, 255), (4, 256), (9, 257), (33, 1000), (300, 4099)])
@pytest.mark.parametrize("quantise", [False, True])
def test_threshold_compaction_matches_numpy_where(rows, cols, quantise):
    x = _matrix(rows, cols, rows * 1000 + cols, quantise)
    x[0, 0] = 0.01                                        # exactly the threshold: `>` excludes it
    iz1, iz2, v = postproc.sparse_above(torch.from_numpy(x).to(DEV), 0.01)
    w1, w2 = np.where(x > np.float32(0.01))
    assert np.array_equal(iz1, w1) and np.array_equal(iz2, w2) and np.array_equal(v, x[w1, w2])


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 2), (5, 3), (7, 255), (4, 256), (9, 257), (33, 1000), (120, 4099)])
@pytest.mark.parametrize("quantise", [False, True])
@pytest.mark.parametrize("distance", [None, 1, 6])
def test_row_peaks_match_scipy_find_peaks(rows, cols, quantise, distance):
    x = _matrix(rows, cols, rows * 77 + cols, quantise)
    h = 0.25
    r, c, v = postproc.find_peaks_rows(torch.from_numpy(x).to(DEV), h, distance)
    wr, wc, wv = [], [], []
    for i in range(rows):
        ip, props = find_peaks(x[i].astype(np.float64), height=h, distance=distance)
        wr.append(np.full(len(ip), i)); wc.append(ip); wv.append(props["peak_heights"])
    wr, wc, wv = np.concatenate(wr), np.concatenate(wc), np.concatenate(wv)
    if quantise and distance not in (None, 1):
        # equal-height peaks closer than `distance`: scipy breaks the tie by an unstable argsort; compare what is tie-free
        assert set(zip(r.tolist(), c.tolist())) <= {(a, b) for a in range(rows) for b in find_peaks(x[a], height=h)[0]}
        return
    assert np.array_equal(r, wr) and np.array_equal(c, wc) and np.array_equal(v.astype(np.float64), wv)


def test_detect_sources_end_to_end_matches_the_reference_statements():
    """process_continuous_days.py:843-891 on a synthetic Out_2 with Gaussian bumps around a few space-time centres: the device
    path (postproc.detect_sources) against the reference's own statements written with numpy / scipy on the host copy."""
    rng = np.random.default_rng(11)
    Q, T = 400, 3000
    dt_win, src_t_kernel, thresh = 0.75, 5.0, 0.15
    xq = np.c_[rng.uniform(0, 300e3, (Q, 2)), rng.uniform(-30e3, 0, Q)]
    ts = np.arange(T) * dt_win
    out = np.zeros((Q, T), dtype=np.float32)
    for _ in range(9):
        c, t0, a = xq[rng.integers(0, Q)], rng.uniform(50, T * dt_win - 50), rng.uniform(0.3, 1.0)
        d = np.linalg.norm((xq - c) * np.array([1, 1, 0.3]), axis=1)
        out += (a * np.exp(-0.5 * (d / 25e3) ** 2)[:, None] * np.exp(-0.5 * ((ts - t0) / 3.0) ** 2)[None, :]).astype(np.float32)
    out += (0.02 * rng.random((Q, T))).astype(np.float32)
    tc_win, sp_win, break_win = src_t_kernel * 1.35, 20e3 * 1.35, 15.0
    got = postproc.detect_sources(torch.from_numpy(out).to(DEV), xq, ts, lambda x: x, thresh, src_t_kernel, dt_win, break_win,
                                  tc_win, sp_win)
    rows = []
    for i in range(Q):                                                            # :844-849
        ip = find_peaks(out[i, :].astype(np.float64), height=thresh, distance=int(1.5 * src_t_kernel / dt_win))
        if len(ip[0]):
            rows.append(np.concatenate((xq[i, :].reshape(1, -1) * np.ones((len(ip[0]), 3)), ts[ip[0]].reshape(-1, 1),
                                        ip[1]["peak_heights"].reshape(-1, 1)), axis=1))
    srcs_init = np.vstack(rows)
    srcs_init = srcs_init[np.argsort(srcs_init[:, 3]), :]
    want = []
    for g in postproc.group_sources(srcs_init, break_win):
        want.append(g if len(g) == 1 else postproc.local_marching(g, lambda x: x, tc_win=tc_win, sp_win=sp_win, scale_depth=0.2,
                                                                  n_steps_max=2, use_directed=False))
    want = np.vstack(want)
    key = lambda a: a[np.lexsort(a.T[::-1])]
    assert 5 <= len(want) < len(srcs_init)
    assert got.shape == want.shape and np.allclose(key(got), key(want), rtol=0, atol=0)
    assert np.all(np.diff(got[:, 3]) >= 0)
