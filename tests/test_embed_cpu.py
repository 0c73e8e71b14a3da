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
