"""GPU: the drop-in API (marex_amd.preprocess_data & co.) end to end on labelled arrays vs the oracle.

Mirrors what the reference's pipeline tests pin (tests/test_gridded_preprocessing.py:49-88, 736-771;
tests/test_unstructured_preprocessing.py): variables, dtypes, dims, attrs, trimmed time axis, extreme
frequency 5 % +- 1 %, and -- stronger than the reference -- bit equality with the oracle.
"""
import warnings

import numpy as np
import pytest

import marex_amd
from marex_amd import binning, calendar, synth
from marex_amd.xr_compat import DataArray
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def gridded_da(start="1990-01-01", periods=24 * 365 + 6, ny=8, nx=14, dtype=np.float32):
    tm = calendar.daily_time_axis(start, periods)
    tab = synth.make_tables(tm, ny, nx)
    x = synth.synth_field(tab).reshape(periods, ny, nx).astype(dtype)
    lat = np.linspace(-60, 60, ny)
    lon = np.linspace(0, 350, nx)
    return DataArray(x, dims=("time", "lat", "lon"), coords={"time": tm, "lat": lat, "lon": lon}, name="sst"), tm


def flat(da_or_arr):
    a = np.asarray(da_or_arr.values if hasattr(da_or_arr, "values") else da_or_arr)
    return a.reshape(a.shape[0], -1) if a.ndim == 3 else a


def test_default_pipeline_schema_and_parity(hot):
    da, tm = gridded_da()
    ds = marex_amd.preprocess_data(da, window_year_baseline=10)
    # schema (detect.py:679-783)
    assert set(ds.data_vars) >= {"dat_anomaly", "mask", "extreme_events", "thresholds"}
    assert ds.dat_anomaly.dtype == np.float32 and ds.extreme_events.dtype == bool and ds.mask.dtype == bool
    assert ds.thresholds.dtype == np.float32
    assert ds.dat_anomaly.dims == ("time", "lat", "lon") and ds.thresholds.dims == ("lat", "lon", "dayofyear")
    assert ds.mask.dims == ("lat", "lon") and ds.thresholds.shape == (8, 14, 366)
    assert ds.attrs["method_anomaly"] == "shifting_baseline" and ds.attrs["method_extreme"] == "hobday_extreme"
    assert ds.attrs["window_year_baseline"] == 10 and ds.attrs["smooth_days_baseline"] == 21
    assert ds.attrs["window_days_hobday"] == 11 and ds.attrs["method_percentile"] == "approximate"
    assert ds.attrs["preprocessing_steps"] == [
        "Rolling climatology using 10 years", "Smoothed with 21-day window", "Day-of-year thresholds with 11 day window",
    ]
    # first 10 years trimmed (tests/test_gridded_preprocessing.py:71-83)
    years = np.asarray(tm).astype("datetime64[Y]").astype(int) + 1970
    assert ds.dat_anomaly.shape[0] == int((years >= 2000).sum())
    assert np.array_equal(ds.dat_anomaly.coords["time"].values, tm[years >= 2000])
    # parity with the oracle (gridded default: 5x5 pooling, detect.py:1451-1452)
    cal = calendar.build_calendar(tm, window_year_baseline=10)
    bt = binning.hobday_bins()
    exp = orc.preprocess_arrays(flat(da), cal, ny=8, nx=14, window_year_baseline=10, edges=bt.edges, centres=bt.centres)
    assert np.array_equal(flat(ds.dat_anomaly), exp["dat_anomaly"], equal_nan=True)
    assert np.array_equal(flat(ds.extreme_events), exp["extreme_events"])
    assert np.array_equal(ds.thresholds.values.reshape(-1, 366), exp["thresholds"], equal_nan=True)
    assert np.array_equal(ds.mask.values.reshape(-1), exp["mask"])
    ocean = exp["mask"]
    freq = flat(ds.extreme_events)[:, ocean].mean()
    assert 0.04 < freq < 0.06  # reference tolerance: 5 % +- 1 % (tests/conftest.py:215-231)


@pytest.mark.parametrize("ma", ["shifting_baseline", "fixed_baseline", "detrend_harmonic", "detrend_fixed_baseline"])
@pytest.mark.parametrize("me,mp", [("hobday_extreme", "approximate"), ("hobday_extreme", "exact"),
                                    ("global_extreme", "approximate"), ("global_extreme", "exact")])
def test_all_method_combinations(hot, ma, me, mp):
    """All anomaly x extreme x percentile methods: parity with the oracle and 2.5-7.5 % extremes
    (tests/test_gridded_preprocessing.py:736-771)."""
    da, tm = gridded_da(periods=14 * 365 + 4, ny=6, nx=10)
    W = 5
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds = marex_amd.preprocess_data(
            da, method_anomaly=ma, method_extreme=me, method_percentile=mp, window_year_baseline=W,
            detrend_orders=[1, 2], threshold_percentile=95,
        )
    cal = calendar.build_calendar(tm, window_year_baseline=W if ma == "shifting_baseline" else None)
    bt = binning.hobday_bins()
    gb = binning.global_bins()
    harm = ma == "detrend_harmonic"
    model = pmodel = None
    if ma.startswith("detrend"):
        model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2], harm)
    edges, centres = (gb.edges, gb.centres) if me == "global_extreme" else (bt.edges, bt.centres)
    exp = orc.preprocess_arrays(
        flat(da), cal, ny=6, nx=10, method_anomaly=ma, method_extreme=me, method_percentile=mp,
        window_year_baseline=W, edges=edges, centres=centres, model=model, pmodel=pmodel,
    )
    assert np.array_equal(flat(ds.dat_anomaly), exp["dat_anomaly"], equal_nan=True)
    thr = ds.thresholds.values
    if me == "hobday_extreme" and mp == "approximate":
        assert ds.thresholds.dims == ("lat", "lon", "dayofyear")
        assert np.array_equal(thr.reshape(-1, 366), exp["thresholds"], equal_nan=True)
    elif me == "hobday_extreme":
        assert ds.thresholds.dims == ("dayofyear", "lat", "lon")  # detect.py:1956
        assert np.array_equal(thr.reshape(366, -1), exp["thresholds"], equal_nan=True)
    else:
        assert ds.thresholds.dims == ("lat", "lon") and thr.dtype == np.float64
        assert np.array_equal(thr.reshape(-1), exp["thresholds"], equal_nan=True)
    assert np.array_equal(flat(ds.extreme_events), exp["extreme_events"])
    freq = flat(ds.extreme_events)[:, exp["mask"]].mean()
    assert 0.025 < freq < 0.075
    if ma != "shifting_baseline" and ma != "fixed_baseline":
        assert abs(float(np.nanmean(flat(ds.dat_anomaly)))) < 0.01  # force_zero_mean (664-666)


def test_unstructured_pipeline(hot):
    tm = calendar.daily_time_axis("1995-01-01", 16 * 365 + 4)
    tab = synth.make_tables(tm, 0, 405, unstructured=True)
    x = synth.synth_field(tab)
    da = DataArray(x, dims=("time", "ncells"), coords={
        "time": tm, "lon": ("ncells", np.linspace(0, 359, 405)), "lat": ("ncells", np.linspace(-80, 80, 405))})
    dims = {"time": "time", "x": "ncells"}
    coords = {"time": "time", "x": "lon", "y": "lat"}
    ds = marex_amd.preprocess_data(da, window_year_baseline=6, dimensions=dims, coordinates=coords)
    assert ds.dat_anomaly.dims == ("time", "ncells") and ds.thresholds.dims == ("ncells", "dayofyear")
    cal = calendar.build_calendar(tm, window_year_baseline=6)
    bt = binning.hobday_bins()
    exp = orc.preprocess_arrays(x, cal, ny=0, nx=405, window_year_baseline=6, edges=bt.edges, centres=bt.centres)
    assert np.array_equal(ds.extreme_events.values, exp["extreme_events"])
    assert np.array_equal(ds.thresholds.values, exp["thresholds"], equal_nan=True)
    with pytest.raises(marex_amd.ConfigurationError, match="window_spatial_hobday is not supported for unstructured grids"):
        marex_amd.preprocess_data(da, window_year_baseline=6, dimensions=dims, coordinates=coords, window_spatial_hobday=5)


def test_invalid_data_is_rejected_like_the_reference(hot):
    da, _ = gridded_da(periods=12 * 365 + 3, ny=5, nx=6)
    bad = da.copy()
    ocean = np.argwhere(np.isfinite(bad.values[0]))
    j, i = ocean[0]
    bad.values[100:103, j, i] = np.nan
    with pytest.raises(marex_amd.DataValidationError, match="contains.*invalid values.*ocean locations"):
        marex_amd.preprocess_data(bad, window_year_baseline=5)
    allnan = DataArray(np.full(da.shape, np.nan, np.float32), dims=da.dims, coords=da.coords)
    with pytest.raises(marex_amd.DataValidationError, match="contains no valid.*finite.*data"):
        marex_amd.preprocess_data(allnan, window_year_baseline=5)
    with pytest.raises(marex_amd.DataValidationError, match="Insufficient data for shifting_baseline"):
        marex_amd.preprocess_data(da, window_year_baseline=15)


def test_public_subfunctions(hot):
    da, tm = gridded_da(periods=9 * 365 + 2, ny=4, nx=6)
    x = flat(da)
    cal = calendar.build_calendar(tm, window_year_baseline=3)
    clim = marex_amd.smoothed_rolling_climatology(da, window_year_baseline=3, smooth_days_baseline=21)
    exp = orc.rolling_climatology(orc.rolling_mean_centered(x, 21), cal.tindex, 3)
    assert np.array_equal(flat(clim), exp, equal_nan=True)
    raw = marex_amd.rolling_climatology(da, window_year_baseline=3)
    assert np.array_equal(flat(raw), orc.rolling_climatology(x, cal.tindex, 3), equal_nan=True)
    an = marex_amd.compute_normalised_anomaly(da, "shifting_baseline", window_year_baseline=3)
    assert an.dat_anomaly.shape == da.shape and np.isnan(flat(an.dat_anomaly)[~cal.kept]).all()
    ext, thr = marex_amd.identify_extremes(an.dat_anomaly.isel(time=slice(int((~cal.kept).sum()), None)),
                                           method_extreme="hobday_extreme", threshold_percentile=95)
    assert ext.dtype == bool and thr.dims == ("lat", "lon", "dayofyear")


def test_validation_summary_kernel(hot):
    """marex_validation_summary == the NumPy form of _validate_data_values' counts (detect.py:224-279), any cell range."""
    import torch

    rng = np.random.default_rng(4)
    for n in (1, 63, 64, 1000, 300001):
        mask = (rng.random(n) < 0.7).astype(np.uint8)
        inv = (rng.integers(0, 50, n) * (rng.random(n) < 0.1)).astype(np.int32)
        md, iv = torch.from_numpy(mask).to(hot.device), torch.from_numpy(inv).to(hot.device)
        for c0, c1 in ((0, n), (n // 3, n - n // 4), (n // 2, n // 2)):
            got = hot.validation_summary(md, iv, (c0, c1)).cpu().tolist()
            m, v = mask[c0:c1].astype(bool), inv[c0:c1]
            exp = [int(m.sum()), int(v[m].sum()), int((v[m] > 0).sum()), int(v[m].max()) if m.any() else 0]
            assert got == exp, (n, c0, c1)
