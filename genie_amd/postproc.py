"""Downstream reduction of the apply loop: from the stacked query output `Out_2` to initial source candidates,
`/root/reference/Code/process_continuous_days.py:812-885` and `LocalMarching` (`process_utils.py:40-100`).

What runs where:
* `Out_2 [n_query, n_time]` stays on the GPU. The threshold pre-filter (`np.where(Out_2 > 0.01)`, :812-813) and the local-maximum /
  height stage of `scipy.signal.find_peaks` (:846) are HIP kernels (`genie_row_select_count` / `genie_row_select_fill`): only
  the sparse `(query, time step, value)` triplets cross to the host (a day of 10 000 queries x 115 200 steps is 4.6 GB dense).
* The remaining steps work on those few triplets on the host, restated from the libraries the reference calls: the distance
  rule of `find_peaks` (scipy `_select_by_peak_distance`), the grouping by `break_win` (:856-870) and `LocalMarching` (max
  propagation over a space-time radius graph, `process_utils.py:40-100`; pinned to the reference by tests/golden/localmarching.npz).
"""
import ctypes

import numpy as np
import torch
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def row_select(x, threshold, mode):
    """Device selection over the rows of a contiguous fp32 GPU matrix `x [rows, cols]`; mode 0: entries > threshold; mode 1:
    local maxima (flat tops -> midpoint, never the first / last column) with value >= threshold. Returns (row int32, col int32,
    value fp32) GPU tensors in row-major order. The only host round trip is the total count (one integer)."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        raise ValueError("row_select: x must be a 2-D fp32 GPU tensor")
    if not x.is_contiguous():
        raise ValueError("row_select: x must be contiguous (a silent copy of a multi-GB Out_2 is not what the caller wants)")
    lib = _lib.load()
    rows, cols = int(x.shape[0]), int(x.shape[1])
    # the device compares in fp32 where numpy / scipy compare the fp32 entries with a float64 threshold: round the threshold to
    # the fp32 value that gives the same decisions (`>`: the largest fp32 <= threshold; `>=`: the smallest fp32 >= threshold)
    th = np.float32(threshold)
    if mode == 0 and float(th) > float(threshold):
        th = np.nextafter(th, np.float32(-np.inf))
    if mode == 1 and float(th) < float(threshold):
        th = np.nextafter(th, np.float32(np.inf))
    with torch.cuda.device(x.device):
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        counts = torch.empty(rows, dtype=torch.int32, device=x.device)
        _lib.check(lib.genie_row_select_count(_ptr(x), rows, cols, ctypes.c_float(float(th)), int(mode), _ptr(counts), st),
                   "genie_row_select_count")
        ends = torch.cumsum(counts.long(), 0)
        offsets = (ends - counts.long()).contiguous()
        n = int(ends[-1].item()) if rows else 0
        out_row = torch.empty(n, dtype=torch.int32, device=x.device)
        out_col = torch.empty(n, dtype=torch.int32, device=x.device)
        out_val = torch.empty(n, dtype=torch.float32, device=x.device)
        if n:
            _lib.check(lib.genie_row_select_fill(_ptr(x), rows, cols, ctypes.c_float(float(th)), int(mode), _ptr(offsets),
                                                 _ptr(out_row), _ptr(out_col), _ptr(out_val), st), "genie_row_select_fill")
    return out_row, out_col, out_val


def sparse_above(Out_2, thresh=0.01):
    """`iz1, iz2 = np.where(Out_2 > 0.01)`; `Out_2_sparse = [iz1, iz2, Out_2[iz1, iz2]]` (process_continuous_days.py:812-813),
    computed on the device; returns numpy (iz1, iz2, values)."""
    r, c, v = row_select(Out_2, thresh, 0)
    return r.cpu().numpy().astype(np.int64), c.cpu().numpy().astype(np.int64), v.cpu().numpy()


def select_by_peak_distance(peaks, priority, distance):
    """The distance rule of scipy.signal.find_peaks (`_select_by_peak_distance`): walking from the highest-priority peak down,
    a kept peak removes every other peak closer than ceil(distance) samples. `peaks` sorted ascending. Returns a bool mask."""
    peaks = np.asarray(peaks, dtype=np.int64)
    n = peaks.shape[0]
    d = int(np.ceil(distance))
    keep = np.ones(n, dtype=bool)
    order = np.argsort(priority)
    for i in range(n - 1, -1, -1):
        j = order[i]
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and peaks[j] - peaks[k] < d:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < n and peaks[k] - peaks[j] < d:
            keep[k] = False
            k += 1
    return keep


def find_peaks_rows(Out_2, height, distance):
    """`find_peaks(Out[i, :], height = thresh, distance = d)` for every row i (process_continuous_days.py:846): candidates (local
    maxima reaching `height`) on the device, the distance rule per row on the host. Returns numpy (row, col, peak height),
    rows ascending, columns ascending within a row."""
    if distance is not None and distance < 1:
        raise ValueError("`distance` must be greater or equal to 1")          # scipy's own check
    if not height > 0.01:
        # the reference runs find_peaks on the array REBUILT from `Out_2 > 0.01` (everything else zero, process_continuous_days.py:812-846);
        # that equals find_peaks on Out_2 itself only for thresholds above the sparsification level
        raise ValueError("find_peaks_rows: height must be > 0.01 (the reference's sparsification threshold)")
    r, c, v = row_select(Out_2, height, 1)
    r, c, v = r.cpu().numpy().astype(np.int64), c.cpu().numpy().astype(np.int64), v.cpu().numpy()
    if distance is None or r.size == 0:
        return r, c, v
    keep = np.ones(r.size, dtype=bool)
    starts = np.flatnonzero(np.r_[True, r[1:] != r[:-1]])
    stops = np.r_[starts[1:], r.size]
    for a, b in zip(starts, stops):
        if b - a > 1:
            keep[a:b] = select_by_peak_distance(c[a:b], v[a:b], distance)
    return r[keep], c[keep], v[keep]


def initial_sources(Out_2, X_query, tsteps_abs, thresh, src_t_kernel, dt_win):
    """`srcs_init` [n, 5] = (query position (3), time, peak height) of every peak of every query row, sorted by time
    (process_continuous_days.py:843-855)."""
    r, c, v = find_peaks_rows(Out_2, thresh, int(1.5 * src_t_kernel / dt_win))
    if r.size == 0:
        return np.zeros((0, 5))
    xq, ts = np.asarray(X_query, dtype=np.float64), np.asarray(tsteps_abs, dtype=np.float64)
    srcs = np.concatenate((xq[r, 0:3], ts[c].reshape(-1, 1), v.astype(np.float64).reshape(-1, 1)), axis=1)
    return srcs[np.argsort(srcs[:, 3])]


def group_sources(srcs_init, break_win):
    """Disjoint groups of time-sorted sources separated by gaps >= break_win (process_continuous_days.py:856-870)."""
    if len(srcs_init) == 0:
        return []
    ibreak = np.where(np.diff(srcs_init[:, 3]) >= break_win)[0]
    edges = np.r_[0, ibreak + 1, len(srcs_init)]
    return [srcs_init[a:b] for a, b in zip(edges[:-1], edges[1:]) if b > a]


def local_marching(srcs, ftrns1, tc_win=5, sp_win=35e3, n_steps_max=100, tol=1e-12, scale_depth=1.0, use_directed=True):
    """`LocalMarching.forward` (process_utils.py:46-100): sources (rows `[x0, x1, x2, t, value]`) linked when within `tc_win`
    in time AND `sp_win` in (depth-scaled) space; the value of every node is replaced by the maximum over its in-neighbours
    (itself included) until nothing changes or `n_steps_max` steps; a node survives when its value is still its own
    (`torch.isclose(..., rtol = tol)`, fp32, atol 1e-8). `use_directed` keeps only the edges that carry a value upwards
    (`value[target] <= value[source]`, :63-66). Returns the surviving rows of `srcs`, in index order (the reference orders them
    by connected component; every caller re-sorts by time, process_continuous_days.py:891)."""
    srcs = np.asarray(srcs, dtype=np.float64)
    n = srcs.shape[0]
    if n == 0:
        return srcs
    scale_vec = np.array([1.0, 1.0, scale_depth]).reshape(1, -1)
    xs = ftrns1(srcs[:, 0:3]) * scale_vec
    lp_t = cKDTree(srcs[:, 3].reshape(-1, 1)).query_ball_point(srcs[:, 3].reshape(-1, 1), r=tc_win)
    lp_x = cKDTree(xs).query_ball_point(xs, r=sp_win)
    src_l, dst_l = [], []
    for i in range(n):
        nb = np.array(sorted(set(lp_t[i]).intersection(lp_x[i])), dtype=np.int64)
        src_l.append(nb)
        dst_l.append(np.full(nb.size, i, dtype=np.int64))
    e0, e1 = np.concatenate(src_l), np.concatenate(dst_l)                    # edge j = e0 -> i = e1
    ncomp, comp = connected_components(coo_matrix((np.ones(e0.size), (e0, e1)), shape=(n, n)), directed=False)
    size = np.bincount(comp, minlength=ncomp)
    val0 = srcs[:, 4].astype(np.float32)
    if use_directed:
        m = val0[e1] <= val0[e0]
        e0, e1 = e0[m], e1[m]
    vals = val0.copy()
    active = size[comp] > 1                                                    # singletons are kept as they are (:72-73)
    for _ in range(int(n_steps_max)):
        new = np.zeros(n, dtype=np.float32)                                    # aggr = 'max' of an empty set is 0 (torch_scatter)
        np.maximum.at(new, e1, vals[e0])
        new = np.where(active, new, vals)
        done = float(np.abs(new - vals).max()) <= tol
        vals = new
        if done:
            break
    keep = ~active | (np.abs(val0 - vals) <= 1e-8 + tol * np.abs(vals))       # torch.isclose(vals_initial, vals, rtol = tol)
    return srcs[keep]


def detect_sources(Out_2, X_query, tsteps_abs, ftrns1, thresh, src_t_kernel, dt_win, break_win, tc_win, sp_win,
                   scale_depth_clustering=0.2):
    """process_continuous_days.py:843-891 in one call: peaks of the device-resident `Out_2` -> time groups -> LocalMarching
    (`n_steps_max = 2, use_directed = False`, :879) -> sources sorted by time, [n, 5]."""
    groups = group_sources(initial_sources(Out_2, X_query, tsteps_abs, thresh, src_t_kernel, dt_win), break_win)
    out = []
    for g in groups:
        if len(g) == 1:
            out.append(g)
        else:
            k = local_marching(g, ftrns1, tc_win=tc_win, sp_win=sp_win, scale_depth=scale_depth_clustering, n_steps_max=2,
                               use_directed=False)
            if len(k):
                out.append(k)
    if not out:
        return np.zeros((0, 5))
    srcs = np.vstack(out)
    return srcs[np.argsort(srcs[:, 3])]
