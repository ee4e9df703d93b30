"""GPU: every entry point that turns a float into a bin / level index, fed with +-inf, NaN, +-3e38, subnormals and values
exactly on the first / last edge (the class of the round-1 `digitize_bin` fault: a float -> int conversion of an unclamped
value is undefined behaviour; all conversions are now clamped as floats or range-guarded -- this pins it per entry point)."""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar
from marex_amd.engine import HotPath
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu

SPECIALS = np.array([np.inf, -np.inf, np.nan, 3.0e38, -3.0e38, 1e-42, -1e-42, 0.0, -0.0, 4.9999995, 5.0, -0.0099999998, 2.5e9, -2.5e9],
                    dtype=np.float32)


def _field(T=3 * 365 + 1, C=260, seed=4):
    rng = np.random.default_rng(seed)
    a = rng.normal(0, 1.2, (T, C)).astype(np.float32)
    idx = rng.random((T, C)) < 0.08
    a[idx] = rng.choice(SPECIALS, int(idx.sum()))
    a[:, 3] = np.nan
    a[:, 4] = np.inf
    a[:, 5] = -3.0e38
    return a


def test_digitize_and_tails_on_special_values(hot):
    a = _field()
    tm = calendar.daily_time_axis("2001-01-01", a.shape[0])
    cal = calendar.build_calendar(tm)
    dcal = hot.upload_calendar(cal)
    ad = torch.from_numpy(a).to(hot.device)
    for bt in (binning.hobday_bins(), binning.hobday_bins(0.05, 2.0)):
        got = HotPath.bins_to_rows(hot.digitize(ad, dcal, bt), a.shape[1]).cpu().numpy().view(np.uint16)
        with np.errstate(invalid="ignore"):
            exp = (np.digitize(a, bt.edges) - 1)[cal.doy_rows]
        assert np.array_equal(got, exp.astype(np.uint16))
        tl = hot.tail_extract(ad, dcal, bt)
        hot.sync()
        aux = tl["aux"].cpu().numpy().view(np.uint32)
        valid = (exp < bt.nb)
        for d in (0, 59, 200, 365):
            rows = slice(cal.doy_start[d], cal.doy_start[d + 1])
            assert np.array_equal(aux[d] & 0x3FF, valid[rows].sum(axis=0))


@pytest.mark.parametrize("method", ["approximate", "exact"])
def test_global_thresholds_on_special_values(hot, method):
    a = _field(T=400, C=130, seed=9)
    bt = binning.hobday_bins()
    got = hot.global_threshold(torch.from_numpy(a).to(hot.device), 95.0, method, bt)["thr_f64"].cpu().numpy()
    if method == "exact":
        exp = orc.global_threshold_exact(a, 95.0)
    else:
        gb = binning.global_bins()
        exp, _ = orc.global_threshold_approx(a, 0.95, gb.edges, gb.centres)
    assert np.array_equal(got, exp, equal_nan=True)


def test_exact_hobday_and_masks_on_special_values(hot):
    a = _field(T=5 * 365 + 1, C=96, seed=2)
    tm = calendar.daily_time_axis("2001-01-01", a.shape[0])
    cal = calendar.build_calendar(tm)
    dcal = hot.upload_calendar(cal)
    ad = torch.from_numpy(a).to(hot.device)
    thr = hot.hobday_thresholds_exact(ad, dcal, 90.0, 5)
    exp = orc.hobday_thresholds_exact(a, cal.doy_out, 90.0, 5)
    assert np.array_equal(thr.cpu().numpy(), exp, equal_nan=True)
    m = hot.mask_ge_doy(ad, thr, dcal)
    hot.sync()
    with np.errstate(invalid="ignore"):
        assert np.array_equal(m["extreme"].cpu().numpy().astype(bool), a >= exp[cal.doy - 1])
