"""GPU parity: exact Hobday percentile, global thresholds (exact / approximate), constant-threshold mask."""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar, synth
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def _anomalies(hot, start, periods, ny, nx, W=4, S=21, mutate=None):
    tm = calendar.daily_time_axis(start, periods)
    x = synth.synth_field(synth.make_tables(tm, ny, nx))
    if mutate is not None:
        mutate(x)
    cal_t = calendar.build_calendar(tm, window_year_baseline=W)
    anom, _ = orc.shifting_baseline_anomaly(x, cal_t, W, S)
    cal = calendar.build_calendar(tm[cal_t.kept])
    return anom, cal


@pytest.mark.parametrize("pct,wd", [(95.0, 11), (90.0, 5), (99.5, 31), (30.0, 11), (50.0, 3), (100.0, 11), (0.0, 11)])
def test_exact_hobday_matches_numpy_nanpercentile(hot, pct, wd):
    anom, cal = _anomalies(hot, "2000-01-01", 12 * 365 + 3, 6, 11)
    exp = orc.hobday_thresholds_exact(anom, cal.doy_out, pct, wd)  # np.nanpercentile itself
    dcal = hot.upload_calendar(cal)
    a = torch.from_numpy(anom).to(hot.device)
    thr = hot.hobday_thresholds_exact(a, dcal, pct, wd)
    hot.sync()
    got = thr.cpu().numpy()
    assert got.shape == (366, anom.shape[1])
    assert np.array_equal(got, exp, equal_nan=True)
    m = hot.mask_ge_doy(a, thr, dcal)
    exp_mask = orc.mask_ge_doy(anom, np.ascontiguousarray(exp.T), cal.doy_out)
    assert np.array_equal(m["extreme"].cpu().numpy().astype(bool), exp_mask)


def test_exact_hobday_with_nan_gaps_and_ties(hot):
    def mutate(x):
        ocean = np.flatnonzero(np.isfinite(x[0]))
        x[2000:2300, ocean[1]] = np.nan
        x[:, ocean[2]] = np.round(x[:, ocean[2]])  # many ties

    anom, cal = _anomalies(hot, "1998-01-01", 10 * 365 + 3, 5, 9, mutate=mutate)
    exp = orc.hobday_thresholds_exact(anom, cal.doy_out, 90.0, 11)
    thr = hot.hobday_thresholds_exact(torch.from_numpy(anom).to(hot.device), hot.upload_calendar(cal), 90.0, 11)
    hot.sync()
    assert np.array_equal(thr.cpu().numpy(), exp, equal_nan=True)


def test_percentile_lerp_mirror_is_numpy():
    """The float32 interpolation the kernel follows == np.nanpercentile (runs on the CPU, kept with its GPU user)."""
    rng = np.random.default_rng(3)
    for _ in range(300):
        m = int(rng.integers(1, 400))
        v = rng.normal(0, 1, m).astype(np.float32)
        p = float(rng.choice([60, 90, 95, 99, 12.5, 50]))
        assert orc.percentile_lerp_f32(np.sort(v), p) == np.nanpercentile(v, p)


@pytest.mark.parametrize("pct", [95.0, 60.0, 99.9])
def test_global_exact_threshold_and_mask(hot, pct):
    anom, cal = _anomalies(hot, "2002-01-01", 9 * 365 + 2, 5, 13)
    exp = orc.global_threshold_exact(anom, pct)
    a = torch.from_numpy(anom).to(hot.device)
    g = hot.global_threshold(a, pct, "exact", None)
    hot.sync()
    got = g["thr_f64"].cpu().numpy()
    assert got.dtype == np.float64 and np.array_equal(got, exp, equal_nan=True)
    m = hot.mask_ge_const(a, g["thr_f64"])
    assert np.array_equal(m["extreme"].cpu().numpy().astype(bool), orc.mask_ge_const(anom, exp))
    assert int(m["n_true"].item()) == int(orc.mask_ge_const(anom, exp).sum())


@pytest.mark.parametrize("pct", [95.0, 90.0])
def test_global_approx_threshold(hot, pct):
    anom, cal = _anomalies(hot, "2002-01-01", 9 * 365 + 2, 5, 13)
    anom[:, 7] = 0.0  # constant anomaly -> clamped to edges[3] with a "too low" count
    gb = binning.global_bins()
    exp, st = orc.global_threshold_approx(anom, pct / 100.0, gb.edges, gb.centres)
    g = hot.global_threshold(torch.from_numpy(anom).to(hot.device), pct, "approximate", binning.hobday_bins())
    hot.sync()
    got = g["thr_f64"].cpu().numpy()
    assert np.array_equal(got, exp, equal_nan=True)
    assert g["stats"]["n_too_low"] == st["n_too_low"] and g["stats"]["n_too_high"] == st["n_too_high"]
    assert g["stats"]["min"] == st["min"] and g["stats"]["max"] == st["max"]
    # within the reference's own tolerance of np.percentile for smooth data (tests/test_detect_helpers.py:172-233)
    ocean = np.isfinite(anom[0]) & (np.arange(anom.shape[1]) != 7)
    assert np.abs(got[ocean] - np.percentile(anom[:, ocean], pct, axis=0)).max() < 0.02


def test_global_threshold_fallback_kernels_agree(hot):
    """Series longer than 65 535 steps use the first-generation kernels (32-bit counters); force them on a short series."""
    anom, cal = _anomalies(hot, "2002-01-01", 9 * 365 + 2, 5, 13)
    a = torch.from_numpy(anom).to(hot.device)
    new_e = hot.global_threshold(a, 95.0, "exact", None)["thr_f64"].cpu().numpy()
    new_a = hot.global_threshold(a, 95.0, "approximate", binning.hobday_bins())["thr_f64"].cpu().numpy()
    with hot.ctx.options(GLOBAL_V1=1):
        old_e = hot.global_threshold(a, 95.0, "exact", None)["thr_f64"].cpu().numpy()
        old_a = hot.global_threshold(a, 95.0, "approximate", binning.hobday_bins())["thr_f64"].cpu().numpy()
    assert np.array_equal(new_e, old_e, equal_nan=True) and np.array_equal(new_a, old_a, equal_nan=True)


def test_global_exact_with_ties_infinities_and_tiny_series(hot):
    rng = np.random.default_rng(5)
    for T in (1, 2, 7, 40):
        x = rng.normal(0, 1, (T, 70)).astype(np.float32)
        x[:, 3] = np.round(x[:, 3])                 # ties
        x[:, 4] = 1.5                               # all equal
        if T > 2:
            x[1, 5] = np.inf
            x[2, 6] = -np.inf
            x[0, 7] = np.nan
        x[:, 8] = np.nan                            # all NaN -> NaN
        for pct in (0.0, 50.0, 95.0, 100.0):
            exp = orc.global_threshold_exact(x, pct)
            got = hot.global_threshold(torch.from_numpy(x).to(hot.device), pct, "exact", None)["thr_f64"].cpu().numpy()
            assert np.array_equal(got, exp, equal_nan=True), (T, pct)


@pytest.mark.parametrize("precision,max_anomaly", [(0.01, 5.0), (0.05, 2.0)])
def test_mask_from_bins_equals_the_value_compare(hot, precision, max_anomaly):
    """marex_mask_ge_doy_bins_f32 decides most samples from their bin; thresholds and anomalies placed exactly on bin edges,
    in the overflow bin, at +-inf and NaN must come out as `anom >= thr` does (detect.py:2003-2004)."""
    import torch

    rng = np.random.default_rng(12)
    tm = calendar.daily_time_axis("2001-01-01", 3 * 365 + 1)
    cal = calendar.build_calendar(tm)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins(precision, max_anomaly)
    T, C = len(tm), 4 * 300
    e = bt.edges[1:]
    anom = rng.normal(0, 1.5, (T, C)).astype(np.float32)
    pick = rng.random((T, C))
    anom[pick < 0.10] = rng.choice(e, int((pick < 0.10).sum()))               # exactly on an edge
    anom[(pick > 0.10) & (pick < 0.12)] = np.nan
    anom[(pick > 0.12) & (pick < 0.13)] = np.inf
    anom[(pick > 0.13) & (pick < 0.14)] = -np.inf
    anom[(pick > 0.14) & (pick < 0.16)] = np.float32(max_anomaly * 3)         # overflow bin
    anom[(pick > 0.16) & (pick < 0.18)] = np.nextafter(rng.choice(e, int(((pick > 0.16) & (pick < 0.18)).sum())), np.float32(-np.inf))
    thr = rng.normal(1.0, 1.0, (366, C)).astype(np.float32)
    tp_ = rng.random((366, C))
    thr[tp_ < 0.2] = rng.choice(e, int((tp_ < 0.2).sum()))
    thr[(tp_ > 0.2) & (tp_ < 0.3)] = np.nan
    thr[(tp_ > 0.3) & (tp_ < 0.32)] = np.inf
    thr[(tp_ > 0.32) & (tp_ < 0.34)] = -np.inf
    thr[(tp_ > 0.34) & (tp_ < 0.37)] = np.float32(max_anomaly * 2)
    thr[(tp_ > 0.37) & (tp_ < 0.40)] = np.float32(-1.0)                       # below the first finite edge
    ad, td = torch.from_numpy(anom).to(hot.device), torch.from_numpy(thr).to(hot.device)
    bd = hot.digitize(ad, dcal, bt)
    with np.errstate(invalid="ignore"):
        exp = anom >= thr[cal.doy - 1]
    with hot.ctx.options(MASK_BINS=1):  # (the automatic choice keeps the plain compare for series this short)
        for cells in (None, (4, C - 8)):
            got = hot.mask_ge_doy(ad, td, dcal, cells=cells, binned=(bd, bt))
            hot.sync()
            c0, c1 = cells or (0, C)
            assert np.array_equal(got["extreme"].cpu().numpy().astype(bool)[:, c0:c1], exp[:, c0:c1])
            assert int(got["n_true"].item()) == int(exp[:, c0:c1].sum())
    with hot.ctx.options(MASK_BINS=0):  # the plain kernel behind the same entry point
        got = hot.mask_ge_doy(ad, td, dcal, binned=(bd, bt))
        hot.sync()
    assert np.array_equal(got["extreme"].cpu().numpy().astype(bool), exp)
    # the mask from TAILS (marex_mask_ge_doy_tails_f32): same thresholds and anomalies
    tl = hot.tail_extract(ad, dcal, bt)
    for cells in (None, (4, C - 8)):
        got = hot.mask_ge_doy_tails(tl, ad, td, dcal, bt, cells=cells)
        hot.sync()
        c0, c1 = cells or (0, C)
        assert np.array_equal(got["extreme"].cpu().numpy().astype(bool)[:, c0:c1], exp[:, c0:c1])
        assert int(got["n_true"].item()) == int(exp[:, c0:c1].sum())
