"""GPU parity: std_normalise branch of detrend_harmonic (marEx/detect.py:2257-2293, 686-715) vs the oracle.

Reference pins mirrored: tests/test_gridded_preprocessing.py:202-272 (variables, dtypes, dims, both extreme
frequencies 5 % +- 1 %); preprocessing step text "Normalised by 30-day rolling STD" (tests/test_detect_helpers.py:319).
"""
import numpy as np
import pytest
import torch

import marex_amd
from marex_amd import calendar, synth
from marex_amd.xr_compat import DataArray
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def _detrended(start, periods, ny, nx, orders=(1, 2)):
    tm = calendar.daily_time_axis(start, periods)
    x = synth.synth_field(synth.make_tables(tm, ny, nx))
    cal = calendar.build_calendar(tm)
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), list(orders), True)
    anom = orc.detrend_anomaly(x, model, pmodel, True)
    anom = anom[0] if isinstance(anom, tuple) else anom
    return tm, x, cal, np.asarray(anom, dtype=np.float32)


def test_std_rolling_and_division_bit_exact(hot):
    tm, x, cal, anom = _detrended("1996-01-01", 11 * 365 + 3, 6, 10)
    ocean = np.flatnonzero(np.isfinite(anom[0]))
    anom[:, ocean[0]] = np.float32(0.25)   # constant cell: STD = 0 -> division by NaN (detect.py:2275)
    anom[300:310, ocean[1]] = np.nan       # NaN terms poison their dayofyear groups and the windows over them
    exp_stn, exp_std = orc.std_normalise(anom, cal.doy)
    dcal = hot.upload_calendar(cal)
    r = hot.std_normalise(torch.from_numpy(anom).to(hot.device), dcal)
    hot.sync()
    got_std = r["STD"].cpu().numpy()
    got_stn = r["dat_stn"].cpu().numpy()
    assert np.array_equal(got_std, exp_std, equal_nan=True)
    assert np.array_equal(got_stn, exp_stn, equal_nan=True)
    assert np.isnan(got_stn[:, ocean[0]]).all() and got_std[:, ocean[0]].max() == 0.0
    # sanity of the semantics themselves: STD is close to the plain per-dayofyear standard deviation of the data
    c = ocean[3]
    sd = np.array([anom[cal.doy == d + 1, c].std() for d in range(365)])
    assert abs(np.nanmean(got_std[:365, c]) / sd.mean() - 1.0) < 0.05


def test_leap_day_only_series_and_short_series(hot):
    """dayofyear 366 occurs in leap years only; a 1-year series gives std 0 everywhere -> all-NaN dat_stn."""
    tm, x, cal, anom = _detrended("2003-01-01", 6 * 365 + 2, 4, 6, orders=(1,))
    exp_stn, exp_std = orc.std_normalise(anom, cal.doy)
    r = hot.std_normalise(torch.from_numpy(anom).to(hot.device), hot.upload_calendar(cal))
    hot.sync()
    assert np.array_equal(r["STD"].cpu().numpy(), exp_std, equal_nan=True)
    assert np.array_equal(r["dat_stn"].cpu().numpy(), exp_stn, equal_nan=True)


@pytest.mark.parametrize("method_extreme", ["global_extreme", "hobday_extreme"])
def test_api_std_normalise(hot, method_extreme):
    periods, ny, nx = 20 * 365 + 5, 8, 12
    tm = calendar.daily_time_axis("1995-01-01", periods)
    x = synth.synth_field(synth.make_tables(tm, ny, nx)).reshape(periods, ny, nx)
    da = DataArray(x, dims=("time", "lat", "lon"),
                   coords={"time": tm, "lat": np.linspace(-50, 50, ny), "lon": np.linspace(0, 330, nx)}, name="sst")
    ds = marex_amd.preprocess_data(da, method_anomaly="detrend_harmonic", method_extreme=method_extreme,
                                   threshold_percentile=95, std_normalise=True, detrend_orders=[1, 2])
    for v in ("extreme_events", "dat_anomaly", "thresholds", "mask", "dat_stn", "STD", "extreme_events_stn", "thresholds_stn"):
        assert v in ds.data_vars, v
    assert ds.attrs["std_normalise"] is True and "Normalised by 30-day rolling STD" in ds.attrs["preprocessing_steps"]
    assert ds.extreme_events_stn.dtype == bool and ds.dat_stn.dtype == np.float32 and ds.STD.dtype == np.float32
    assert ds.dat_stn.dims == ("time", "lat", "lon") and ds.STD.dims == ("lat", "lon", "dayofyear")
    ocean = ds.mask.values
    for name in ("extreme_events", "extreme_events_stn"):
        freq = ds[name].values[:, ocean].mean()
        assert 0.04 < freq < 0.06, (name, freq)  # tests/conftest.py:215-231 of the reference
    # parity of the extra variables with the oracle
    cal = calendar.build_calendar(tm)
    exp_stn, exp_std = orc.std_normalise(ds.dat_anomaly.values.reshape(periods, -1), cal.doy)
    assert np.array_equal(ds.dat_stn.values.reshape(periods, -1), exp_stn, equal_nan=True)
    assert np.array_equal(ds.STD.values.reshape(-1, 366), exp_std.T, equal_nan=True)
    da2 = marex_amd.compute_normalised_anomaly(da, method_anomaly="detrend_harmonic", std_normalise=True, detrend_orders=[1, 2])
    assert np.array_equal(da2.dat_stn.values.reshape(periods, -1), exp_stn, equal_nan=True)
