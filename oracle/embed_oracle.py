"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the pick -> per-product-node feature embedding that feeds the hot path,
`extract_input_from_data` (`/root/reference/Code/process_utils.py:460-642`, live path `use_updated_input = True`,
`process_continuous_days.py:607,776`; `use_sign_input` False or True, `trv_times` given, `batch_grids = False`).

Pinned by `tests/golden/embed_*.npz`, produced by the reference's own function in the build container
(`oracle/make_golden.py`); checked in `tests/test_embed_cpu.py`.
"""
import numpy as np


def embed_time_series(P_slice, t0, max_t, kernel_sig_t, dt, sta_index, n_sta):
    """Per-station Gaussian-kernel time series of the P- and S-labelled picks (process_utils.py:499-569).

    P_slice [n,5] (t, station, amp, prob, phase); sta_index[n] = row of each pick's station in the output (0..n_sta-1).
    Returns embed_p, embed_s float32 [n_sta, n_time], abs_time_ref float64 [n_time]."""
    t_offset = 3.0 * kernel_sig_t
    abs_time_ref = np.arange(t0 - t_offset, t0 + max_t + t_offset + dt, dt)                                  # :501
    n_time = len(abs_time_ref)
    num_index_extra = np.ceil(3 * kernel_sig_t / dt)                                                         # :518
    vec_repeat = np.arange(-num_index_extra, num_index_extra + 1).astype("int")
    out = []
    for phase in (0, 1):
        sel = np.where(P_slice[:, 4] == phase)[0]                                                            # :507-508
        emb = np.zeros((n_sta, n_time), dtype=np.float32)
        if sel.size:
            nearest = ((P_slice[sel, 0] - abs_time_ref[0]) / dt).astype("int")                               # :514-515
            idx = nearest.reshape(-1, 1) + vec_repeat.reshape(1, -1)                                         # :534
            inside = (idx >= 0) * (idx < n_time)                                                             # :537
            idx = np.minimum(np.maximum(0, idx), n_time - 1)                                                 # :540
            tv = P_slice[sel, 0].reshape(-1, 1) - abs_time_ref[idx]                                          # :543
            vals = (inside * np.exp(-0.5 * (tv ** 2) / (kernel_sig_t ** 2))).astype(np.float32)              # :545, torch.Tensor() cast :563
            rows = np.repeat(sta_index[sel], idx.shape[1])
            np.maximum.at(emb, (rows, idx.reshape(-1)), vals.reshape(-1))                                    # scatter 'max' :563
        emb[:, 0] = 0.0                                                                                      # :565-568
        emb[:, n_time - 1] = 0.0
        out.append(emb)
    return out[0], out[1], abs_time_ref


def _neg_slope_sign(emb):
    """sign(-diff(flat series)) of process_utils.py:610-614 (`use_sign_input: True`, config.yaml:93): the series of the embedded stations
    are concatenated; the element after a series' last sample is the next series' first one (both are zero), the one after the very last
    sample is linearly extrapolated. Every product node reads the sign at the index it reads the value at."""
    flat = emb.reshape(-1).astype(np.float32)
    nxt = np.concatenate((flat[1:], flat[-1:] + (flat[-1:] - flat[-2:-1])))
    return np.sign(-(nxt - flat)).reshape(emb.shape).astype(np.float32)


def extract_input_from_data(P, t0, ind_use, n_sta_all, trv_times, A_src_in_sta, max_t, kernel_sig_t, dt, use_sign_input=False):
    """Slice [P,4], Mask [P,4] float32 for the window starting at t0 (process_utils.py:460-642).

    trv_times [G, n_sta_all, 2]; A_src_in_sta [2, P] = [station index within ind_use; source node] per product node."""
    ineed = np.where((P[:, 0] > (t0 - 2.0 * kernel_sig_t)) * (P[:, 0] < (t0 + max_t + 2.0 * kernel_sig_t)))[0]   # :476
    P_slice = P[ineed]
    perm = -1 * np.ones(n_sta_all, dtype="int")
    perm[ind_use] = np.arange(len(ind_use))                                                                   # :486-487
    P_slice = P_slice[perm[P_slice[:, 1].astype("int")] > -1]                                                 # :480-483
    sta_index = perm[P_slice[:, 1].astype("int")]
    embed_p, embed_s, abs_time_ref = embed_time_series(P_slice, t0, max_t, kernel_sig_t, dt, sta_index, len(ind_use))
    embed = np.maximum(embed_p, embed_s)                                                                      # :569
    n_time = embed.shape[1]
    sta = np.asarray(A_src_in_sta[0]).astype("int")
    src = np.asarray(A_src_in_sta[1]).astype("int")
    # t0 is a float64 ARRAY in the reference (tsteps_slice), so the float32 travel times are promoted to float64 here
    trv_ind = ((trv_times[src, ind_use[sta], :].astype(np.float64) + t0 - abs_time_ref[0]) / dt).astype("int")   # :605
    ip, is_ = trv_ind[:, 0], trv_ind[:, 1]
    has_picks = np.zeros(len(ind_use), dtype=bool)
    has_picks[np.unique(sta_index)] = True        # stations without any pick in the window are not embedded -> 0 (:598-600)
    ok = has_picks[sta]
    Slice = np.zeros((len(sta), 4), dtype=np.float32)
    Slice[ok, 0] = embed[sta[ok], ip[ok]]                                                                     # :612
    Slice[ok, 1] = embed[sta[ok], is_[ok]]                                                                    # :613
    Slice[ok, 2] = embed_p[sta[ok], ip[ok]]                                                                   # :614
    Slice[ok, 3] = embed_s[sta[ok], is_[ok]]                                                                  # :615
    if use_sign_input:                                                                                        # :610-614
        # (the reference's flat series hold the stations WITH picks only; a series starts and ends with a zero sample, so which series
        # follows which does not matter)
        sg, sp, ss = _neg_slope_sign(embed), _neg_slope_sign(embed_p), _neg_slope_sign(embed_s)
        Slice[ok, 0] *= sg[sta[ok], ip[ok]]
        Slice[ok, 1] *= sg[sta[ok], is_[ok]]
        Slice[ok, 2] *= sp[sta[ok], ip[ok]]
        Slice[ok, 3] *= ss[sta[ok], is_[ok]]
    Mask = (np.abs(Slice) > 0.01).astype(np.float32)                                                          # :629
    return Slice, Mask


def extract_pick_inputs_from_data(P_slice, n_sta_all, ind_use, t0, max_t, t_win=10.0):
    """The per-window pick lists a `forward_fixed` call consumes (process_utils.py:644-699, `use_batch = False`): from the picks
    `P_slice` [n, 5] (t, ABSOLUTE station index, amp, prob, phase) in the caller's order, those within `t_win + max_t / 2` of the
    window centre `t0 + max_t / 2` (cKDTree.query_ball_point, :665: inclusive; a multi-point query returns ascending indices) whose
    station is one of `ind_use` (:680-687), station indices mapped through `perm_vec` to positions in `ind_use` (:678-679), sorted by
    `np.lexsort((times, indices))` (:690: station, then time, ties in the caller's order), times relative to the window start (:691).
    Returns (lp_times float64 [m], lp_stations int [m], lp_phases float64 [m], lp_meta float64 [m, 5])."""
    P_slice = np.asarray(P_slice, dtype=np.float64)
    centre, radius = float(t0) + max_t / 2.0, t_win + max_t / 2.0
    lp = np.where(np.abs(P_slice[:, 0] - centre) <= radius)[0]                                     # :665
    perm_vec = -1 * np.ones(n_sta_all).astype("int")
    perm_vec[ind_use] = np.arange(len(ind_use))                                                    # :678-679
    meta = P_slice[lp, :]
    phase_vals = P_slice[lp, 4]
    times = meta[:, 0]
    indices = perm_vec[meta[:, 1].astype("int")]
    ineed = np.where(indices > -1)[0]                                                              # :684
    times, indices, phase_vals, meta = times[ineed], indices[ineed], phase_vals[ineed], meta[ineed]
    lex_sort = np.lexsort((times, indices))                                                        # :690
    return times[lex_sort] - float(t0), indices[lex_sort], phase_vals[lex_sort], meta[lex_sort]    # :691-694


def window_pick_slice(P, t0, ind_use, max_t, kernel_sig_t):
    """`P_slice` as extract_input_from_data hands it to extract_pick_inputs_from_data (process_utils.py:476-483): picks strictly inside
    `(t0 - 2 sigma, t0 + max_t + 2 sigma)` whose station is in `ind_use`, in the caller's order."""
    P = np.asarray(P, dtype=np.float64)
    ineed = np.where((P[:, 0] > (t0 - 2.0 * kernel_sig_t)) * (P[:, 0] < (t0 + max_t + 2.0 * kernel_sig_t)))[0]   # :476
    P_slice = P[ineed]
    return P_slice[np.isin(P_slice[:, 1].astype("int"), np.asarray(ind_use))]                                     # :480-483
