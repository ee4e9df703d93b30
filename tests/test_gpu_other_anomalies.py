"""GPU parity of the other anomaly methods (fixed_baseline, detrend_harmonic, detrend_fixed_baseline) vs the oracle."""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar, synth
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def _field(start, periods, ny, nx):
    tm = calendar.daily_time_axis(start, periods)
    tab = synth.make_tables(tm, ny, nx)
    return tm, synth.synth_field(tab)


def _same(a, b):
    return np.array_equal(np.asarray(a, np.float32), np.asarray(b, np.float32), equal_nan=True)


@pytest.mark.parametrize("ref", [None, (2004, 2008)])
def test_fixed_baseline_bit_exact(hot, ref):
    tm, x = _field("2001-03-01", 11 * 365 + 7, 7, 19)
    cal = calendar.build_calendar(tm)
    bt = binning.hobday_bins()
    exp, mask = orc.fixed_baseline_anomaly(x, cal, ref)
    dcal = hot.upload_calendar(cal)
    got = hot.fixed_baseline(torch.from_numpy(x).to(hot.device), dcal, ref, bt)
    hot.sync()
    assert _same(got["out"].cpu().numpy(), exp)
    assert np.array_equal(got["mask"].cpu().numpy().astype(bool), mask)
    assert np.array_equal(got["invalid_count"].cpu().numpy(), (~np.isfinite(x)).sum(axis=0))
    bins_exp = orc.digitize_bins(exp, bt.edges)[cal.doy_rows]
    assert np.array_equal(hot.bins_to_rows(got["bins"], x.shape[1]).cpu().numpy().view(np.uint16), bins_exp)


@pytest.mark.parametrize("orders,harm,fzm", [([1], True, True), ([1, 2], False, True), ([2, 3], True, False)])
def test_detrend_bit_exact_and_tolerance(hot, orders, harm, fzm):
    tm, x = _field("1996-01-01", 9 * 365 + 2, 5, 13)
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), orders, harm)
    exp = orc.detrend_anomaly(x, model, pmodel, fzm)
    got = hot.detrend(torch.from_numpy(x).to(hot.device), model, pmodel, fzm)
    hot.sync()
    out = got["out"].cpu().numpy()
    assert _same(out, exp)  # same summation order as the oracle => same bits
    # and independent of any summation order: close to a float64 least-squares residual
    ocean = np.isfinite(x[0])
    x64 = x[:, ocean].astype(np.float64)
    ref = x64 - model.T @ (pmodel.T @ x64)
    if fzm:
        ref = ref - ref.mean(axis=0)
    assert np.allclose(out[:, ocean], ref, rtol=0, atol=1e-5 * np.abs(x64).max())
    if fzm:
        assert np.abs(out[:, ocean].mean(axis=0)).max() < 1e-5


def test_detrend_then_fixed_baseline_and_digitize(hot):
    tm, x = _field("1999-06-01", 8 * 365 + 2, 6, 11)
    cal = calendar.build_calendar(tm)
    bt = binning.hobday_bins()
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1], False)
    exp_d = orc.detrend_anomaly(x, model, pmodel, True)
    exp, _ = orc.fixed_baseline_anomaly(exp_d, cal, None)
    dcal = hot.upload_calendar(cal)
    d = hot.detrend(torch.from_numpy(x).to(hot.device), model, pmodel, True)
    f = hot.fixed_baseline(d["out"], dcal, None, None, count_invalid=False)
    b = hot.digitize(f["out"], dcal, bt)
    hot.sync()
    assert _same(f["out"].cpu().numpy(), exp)
    assert np.array_equal(hot.bins_to_rows(b, x.shape[1]).cpu().numpy().view(np.uint16), orc.digitize_bins(exp, bt.edges)[cal.doy_rows])
