"""Pin the oracle to the real reference: golden vectors made by tests/golden/make_goldens.py.

``hist_quantile_goldens.npz`` holds outputs of the reference's own ``_rolling_histogram_quantile``
(marEx/detect.py:2465-2559) for 9 histogram regimes x 5 day-windows x 4 quantiles x 2 input dtypes
(+ float64 centres); the oracle's restatement must reproduce every one BIT FOR BIT.
"""
import json
import os

import numpy as np
import pytest

from marex_amd import binning
from oracle import marex_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "hist_quantile_goldens.npz"))


def test_bin_tables_match_reference_tables(gold):
    bt = binning.hobday_bins()
    assert bt.nb == 502 and bt.edges.dtype == np.float32 and bt.centres.dtype == np.float32
    assert np.array_equal(bt.edges.view(np.uint32), gold["edges"].view(np.uint32))
    assert np.array_equal(bt.centres.view(np.uint32), gold["centres"].view(np.uint32))
    # SURVEY.md A.4 spot values
    assert bt.edges[2] == 0.0 and bt.centres[0] == 0.0
    assert bt.lower_bound == bt.edges[3] and bt.upper_bound == bt.edges[-2]


def test_rolling_histogram_quantile_bit_exact(gold):
    cases = list(gold["cases"])
    assert len(cases) >= 360
    for key in cases:
        parts = key.split("|")
        name, wd, q, dt = parts[0], int(parts[1][2:]), float(parts[2][1:]), parts[3]
        centres = gold["centres64"] if len(parts) > 4 else gold["centres"]
        got = orc.rolling_histogram_quantile(gold["hist|" + name].astype(dt), wd, q, centres)
        exp = gold["out|" + key]
        assert got.dtype == np.float32
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), key


def test_quantile_vectorised_over_cells_equals_per_cell(gold):
    names = ["normal40", "uniform", "empty_doys", "constant_zero"]
    stack = np.stack([gold["hist|" + n] for n in names])
    got = orc.rolling_histogram_quantile(stack, 11, 0.95, gold["centres"])
    for i, n in enumerate(names):
        exp = gold[f"out|{n}|wd11|q0.95|uint16"]
        assert np.array_equal(got[i].view(np.uint32), exp.view(np.uint32))


def test_window_of_one_day_is_rejected(gold):
    # wd=1 crashes the reference itself; the oracle refuses it explicitly
    with pytest.raises(ValueError):
        orc.rolling_histogram_quantile(gold["hist|normal40"], 1, 0.95, gold["centres"])


def test_preprocessing_steps_strings():
    from marex_amd.detect import _get_preprocessing_steps

    with open(os.path.join(GOLD, "preprocessing_steps.json")) as f:
        cases = json.load(f)
    assert len(cases) == 64
    for c in cases:
        a = dict(c["args"])
        if a["reference_period"] is not None:
            a["reference_period"] = tuple(a["reference_period"])
        assert _get_preprocessing_steps(**a) == c["steps"], a
