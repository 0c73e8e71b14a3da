"""CPU: the embedding oracle (oracle/embed_oracle.py) against the reference's own extract_input_from_data
(process_utils.py:460-642) golden vectors. Exact semantics incl. discretisation; tolerance 1e-6 (float32 exp)."""
import os

import numpy as np
import pytest

from oracle import embed_oracle as E
from tests.util import GOLDEN_DIR

EMBED_CASES = ["embed_14x60_a", "embed_14x60_b"]


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    S, G = int(z["n_sta"]), int(z["n_grid"])
    A = np.stack([np.tile(np.arange(S), G), np.repeat(np.arange(G), S)], axis=0)
    return z, S, G, A


@pytest.mark.parametrize("name", EMBED_CASES)
def test_embed_oracle_matches_reference(name):
    z, S, G, A = load(name)
    Slice, Mask = E.extract_input_from_data(z["P"], float(z["t0"]), np.arange(S), S, z["trv_times"], A, float(z["max_t"]),
                                            float(z["kernel_sig_t"]), float(z["dt"]))
    assert Slice.shape == z["Slice"].shape
    assert np.abs(Slice - z["Slice"]).max() <= 1e-6
    assert np.array_equal(Mask.astype(np.uint8), z["Mask"])
    assert (z["Slice"] > 0.5).sum() > 50          # the fixture is not trivial


@pytest.mark.parametrize("name", ["embed_sign_14x60_a", "embed_sign_14x60_b"])
def test_embed_oracle_with_sign_input_matches_reference(name):
    """`use_sign_input: True` (config.yaml:93, process_utils.py:610-614): every feature carries the sign of the negative slope of the
    series it was read from. Fixtures: the reference's extract_input_from_data with the flag set (oracle/make_golden.py --embed-sign)."""
    z, S, G, A = load(name)
    Slice, Mask = E.extract_input_from_data(z["P"], float(z["t0"]), np.arange(S), S, z["trv_times"], A, float(z["max_t"]),
                                            float(z["kernel_sig_t"]), float(z["dt"]), use_sign_input=True)
    assert np.abs(Slice - z["Slice"]).max() <= 1e-6
    assert np.array_equal(Mask.astype(np.uint8), z["Mask"])
    assert (z["Slice"] < -0.5).sum() > 20 and (z["Slice"] > 0.5).sum() > 20          # both signs occur


PICK_CASES = ["picks_14x60_a", "picks_14x60_b", "picks_14x60_c"]


@pytest.mark.parametrize("name", PICK_CASES)
def test_pick_inputs_oracle_and_product_selector_match_reference(name):
    """SURVEY.md 8 f-1, the second half: the per-window pick lists `[lp_times, lp_stations, lp_phases, lp_meta]` that the reference's
    extract_input_from_data returns (process_utils.py:637 -> extract_pick_inputs_from_data :644-699). Fixtures from the reference's own
    functions (oracle/make_golden.py --picks): picks in file order (not time order), `ind_use` a subset of the station file,
    duplicated (station, time) pairs with different phases, one case where the ball query of :665 trims the window's slice, one
    where the function was called directly with a small `t_win`. Index / order work: everything EXACT, for the oracle restatement
    (oracle/embed_oracle.py) and for the product's selector (genie_amd.apply.ResidentPicks, here on CPU tensors). The same fixtures
    also pin the embedding oracle with a station subset."""
    import torch
    from genie_amd import apply
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    P, t0, ind = z["P"], float(z["t0"]), z["ind_use"]
    max_t, sig, n_all = float(z["max_t"]), float(z["kernel_sig_t"]), int(z["n_sta_all"])
    keys = ("lp_times", "lp_stations", "lp_phases", "lp_meta")
    P_slice = E.window_pick_slice(P, t0, ind, max_t, sig)
    got = E.extract_pick_inputs_from_data(P_slice, n_all, ind, t0, max_t)
    for a, k in zip(got, keys):
        assert a.shape == z[k].shape and np.array_equal(a, z[k]), k
    rp = apply.ResidentPicks(P, ind, n_all, "cpu")
    tp, ip, ph, idx = rp.pick_inputs(t0, max_t, sig)
    assert tp.dtype == torch.float64 and ip.dtype == torch.int64
    assert np.array_equal(tp.numpy(), z["lp_times"]) and np.array_equal(ip.numpy(), z["lp_stations"])
    assert np.array_equal(ph.numpy(), z["lp_phases"]) and np.array_equal(rp.meta(idx), z["lp_meta"])
    assert len(z["lp_times"]) > 100 and len(np.unique(z["lp_stations"])) == len(ind)
    if name == "picks_14x60_b":       # 2 sigma = 14 s > t_win = 10 s: the ball query drops picks of the embedding range
        lo, hi = rp.embed_range(t0, max_t, sig)
        assert hi - lo > len(z["lp_times"])
    if "lp2_times" in z.files:        # the reference function called directly with another t_win
        tw = float(z["t_win_direct"])
        got = E.extract_pick_inputs_from_data(P_slice, n_all, ind, t0, max_t, t_win=tw)
        tp, ip, ph, idx = rp.pick_inputs(t0, max_t, sig, t_win=tw)
        for a, b, k in zip(got, (tp.numpy(), ip.numpy(), ph.numpy(), rp.meta(idx)), ("lp2_times", "lp2_stations", "lp2_phases", "lp2_meta")):
            assert np.array_equal(a, z[k]) and np.array_equal(b, z[k]), k
    # the embedding range handed to genie_embed_window holds exactly the picks of the reference's P_slice (as a multiset of rows)
    lo, hi = rp.embed_range(t0, max_t, sig)
    assert sorted(rp.index_host[lo:hi].tolist()) == sorted(np.nonzero((P[:, 0] > t0 - 2 * sig) & (P[:, 0] < t0 + max_t + 2 * sig)
                                                                        & np.isin(P[:, 1].astype(int), ind))[0].tolist())
    # Slice / Mask of the same call, with the station subset
    G, nu = int(z["n_grid"]), len(ind)
    A = np.stack([np.tile(np.arange(nu), G), np.repeat(np.arange(G), nu)])
    Slice, Mask = E.extract_input_from_data(P, t0, ind, n_all, z["trv_times"], A, max_t, sig, float(z["dt"]))
    assert np.abs(Slice - z["Slice"]).max() <= 1e-6 and np.array_equal(Mask.astype(np.uint8), z["Mask"])


def test_resident_picks_without_any_pick_in_the_window():
    from genie_amd import apply
    z = np.load(os.path.join(GOLDEN_DIR, "picks_14x60_a.npz"))
    rp = apply.ResidentPicks(z["P"], z["ind_use"], int(z["n_sta_all"]), "cpu", use_phase_types=False)
    assert rp.embed_args(50000.0, float(z["max_t"]), 3.0) is None
    tp, ip, ph, idx = rp.pick_inputs(50000.0, float(z["max_t"]), 3.0)
    assert tp.numel() == 0 and ip.numel() == 0 and rp.meta(idx).shape == (0, 5)
    assert float(rp.phase_f.abs().max()) == 0.0          # process_continuous_days.py:562-563
