"""CPU: host half of the downstream reduction (genie_amd/postproc.py): `local_marching` against the reference's own
LocalMarching (tests/golden/localmarching.npz, oracle/make_golden.py --postproc), the distance rule of find_peaks against
scipy.signal.find_peaks (the library the reference calls, process_continuous_days.py:846), the grouping by break_win."""
import os

import numpy as np
import pytest
from scipy.signal import find_peaks

from genie_amd import postproc
from tests.util import GOLDEN_DIR


def _sorted_rows(a):
    a = np.asarray(a)
    return a[np.lexsort(a.T[::-1])] if len(a) else np.zeros((0, 5))


@pytest.mark.parametrize("tag", ["apply", "default", "wide"])
def test_local_marching_matches_reference(tag):
    z = np.load(os.path.join(GOLDEN_DIR, "localmarching.npz"))
    kw = {k[len("kw_%s_" % tag):]: float(z[k]) for k in z.files if k.startswith("kw_%s_" % tag)}
    for k in ("n_steps_max",):
        if k in kw:
            kw[k] = int(kw[k])
    if "use_directed" in kw:
        kw["use_directed"] = bool(kw["use_directed"])
    got = _sorted_rows(postproc.local_marching(z["srcs"], lambda x: x, **kw))
    want = z["keep_" + tag]
    assert got.shape == want.shape and np.array_equal(got, want)
    assert 0 < len(want) < len(z["srcs"])


def test_distance_rule_matches_scipy_find_peaks():
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(30, 400))
        x = rng.random(n) * (rng.random(n) < 0.6)
        if trial % 3 == 0:
            x = np.round(x, 1)                                   # flat tops and ties
        h, d = 0.3, int(rng.integers(1, 9))
        cand = find_peaks(x, height=h)[0]
        keep = postproc.select_by_peak_distance(cand, x[cand], d)
        assert np.array_equal(cand[keep], find_peaks(x, height=h, distance=d)[0]), trial


def test_grouping_by_break_window():
    t = np.array([0.0, 1.0, 2.0, 50.0, 51.0, 200.0])
    srcs = np.c_[np.zeros((6, 3)), t, np.ones(6)]
    groups = postproc.group_sources(srcs, 15.0)
    assert [len(g) for g in groups] == [3, 2, 1]
    assert postproc.group_sources(srcs[:0], 15.0) == []
    assert [len(g) for g in postproc.group_sources(srcs, 1000.0)] == [6]
