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


@pytest.mark.parametrize("orders,fzm,ref,years", [([1], True, None, 8), ([1, 2], True, (2001, 2005), 11), ([1, 2, 3], False, None, 9),
                                                  ([1, 2, 3, 4], True, None, 60), ([2], False, (1999, 2003), 6)])
def test_detrend_fixed_baseline_as_one_chain(hot, orders, fzm, ref, years):
    """marex_detrend_fixed_baseline_f32 (fit, residual mean, climatology kernel recomputing the residuals from x) against the
    oracle and against the two stages run one after the other: same bits; the deferred-mean pair likewise; 60 years exercise
    the 128-row bucket instance, gaps and a late-starting cell the NaN handling, six terms the fall-back to the two stages."""
    tm, x = _field("1998-02-01", years * 365 + years // 4, 5, 17)
    x[300:340, 3] = np.nan
    x[:500, 4] = np.nan
    cal = calendar.build_calendar(tm)
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), orders, False)
    exp_d = orc.detrend_anomaly(x, model, pmodel, fzm)
    exp, _ = orc.fixed_baseline_anomaly(exp_d, cal, ref)
    dcal = hot.upload_calendar(cal)
    xd = torch.from_numpy(x).to(hot.device)
    got = hot.detrend_fixed_baseline(xd, model, pmodel, fzm, dcal, ref)
    hot.sync()
    assert _same(got["out"].cpu().numpy(), exp)
    assert np.array_equal(got["mask"].cpu().numpy().astype(bool), np.isfinite(x[0]))
    assert np.array_equal(got["invalid_count"].cpu().numpy(), (~np.isfinite(x)).sum(axis=0))
    with hot.ctx.options(DETREND_FUSED=0):
        two = hot.detrend_fixed_baseline(xd, model, pmodel, fzm, dcal, ref)
        hot.sync()
    assert np.array_equal(got["out"].cpu().numpy(), two["out"].cpu().numpy(), equal_nan=True)
    with hot.ctx.options(DETREND_FUSED=0, FIXED_REG=0):  # and the two-read climatology kernel
        old = hot.detrend_fixed_baseline(xd, model, pmodel, fzm, dcal, ref)
        hot.sync()
    assert np.array_equal(got["out"].cpu().numpy(), old["out"].cpu().numpy(), equal_nan=True)


def test_detrend_fixed_baseline_falls_back_beyond_five_terms(hot):
    tm, x = _field("2000-01-01", 7 * 365 + 2, 4, 9)
    cal = calendar.build_calendar(tm)
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2, 3, 4, 5], False)  # six terms
    exp, _ = orc.fixed_baseline_anomaly(orc.detrend_anomaly(x, model, pmodel, True), cal, None)
    got = hot.detrend_fixed_baseline(torch.from_numpy(x).to(hot.device), model, pmodel, True, hot.upload_calendar(cal), None)
    hot.sync()
    assert _same(got["out"].cpu().numpy(), exp)


def test_two_stage_fallback_keeps_the_raw_validation_outputs_with_a_workspace(hot):
    """detrend_fixed_baseline as two stages (DETREND_FUSED=0, or more than five terms) with a SHARED workspace: the second
    stage's mask / count buffers are its own, so the returned mask and invalid counts are still those of the raw field."""
    tm, x = _field("2001-01-01", 7 * 365 + 2, 4, 11)
    ocean = np.flatnonzero(np.isfinite(x[0]))
    x[100:130, ocean[2]] = np.nan   # an ocean cell with a gap
    x[40, ocean[4]] = np.inf
    cal = calendar.build_calendar(tm)
    dcal = hot.upload_calendar(cal)
    xd = torch.from_numpy(x).to(hot.device)
    exp_mask, exp_inv = np.isfinite(x[0]), (~np.isfinite(x)).sum(axis=0)
    assert exp_inv[ocean[2]] == 30 and exp_inv[ocean[4]] == 1
    for orders, opts in (([1, 2], dict(DETREND_FUSED=0)), ([1, 2, 3, 4, 5], {})):
        model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), orders, False)
        wsp = {}
        with hot.ctx.options(**opts):
            for _ in range(2):  # the second call reuses every buffer
                got = hot.detrend_fixed_baseline(xd, model, pmodel, True, dcal, None, wsp=wsp)
                hot.sync()
                assert np.array_equal(got["mask"].cpu().numpy().astype(bool), exp_mask)
                assert np.array_equal(got["invalid_count"].cpu().numpy(), exp_inv)
        exp, _ = orc.fixed_baseline_anomaly(orc.detrend_anomaly(x, model, pmodel, True), cal, None)
        assert _same(got["out"].cpu().numpy(), exp)


def test_options_context_restores_previous_values(hot):
    """ctx.options puts back what was set before it (set_option or MAREX_<NAME> seeds), it does not just clear."""
    ctx = hot.ctx
    assert "THR_DD" not in ctx.py_opts and "THR_TILE" not in ctx.py_opts
    ctx.set_option("THR_DD", 40)
    try:
        with ctx.options(THR_DD=20, THR_TILE=16):
            assert ctx.py_opts["THR_DD"] == 20 and ctx.py_opts["THR_TILE"] == 16
        assert ctx.py_opts["THR_DD"] == 40 and "THR_TILE" not in ctx.py_opts
    finally:
        ctx.set_option("THR_DD", None)
    assert "THR_DD" not in ctx.py_opts
