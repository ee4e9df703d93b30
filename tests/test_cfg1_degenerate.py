"""BASELINE.json configs[0] at its exact shape: 365-day x 45x90 synthetic gridded SST, fixed_baseline + global_extreme p95.

SURVEY.md 8(d) "Degenerate cfg1": with one year of data every dayofyear holds one sample, so the daily climatology IS the
data (detect.py:2365-2379), anomalies are identically 0 on the ocean, the 1-D histogram quantile is the centre of the bin
[0, 0.01) = 0.005 (detect.py:2789-2832), which lies below edges[3] = 0.01 and is clamped to it with a UserWarning
(detect.py:2853-2863); the mask `0 >= 0.01` is all False (detect.py:2915).  The oracle states it on the CPU; the GPU test
runs the public API and the engine on the same field.
"""
import warnings

import numpy as np
import pytest

import marex_amd
from marex_amd import binning, calendar, synth
from marex_amd.xr_compat import DataArray
from oracle import marex_oracle as orc

NY, NX, T = 45, 90, 365


def _field():
    tm = calendar.daily_time_axis("2001-01-01", T)
    x = synth.synth_field(synth.make_tables(tm, NY, NX))
    return tm, x


def test_oracle_cfg1_is_degenerate():
    tm, x = _field()
    cal = calendar.build_calendar(tm)
    gb = binning.global_bins()
    r = orc.preprocess_arrays(x, cal, ny=NY, nx=NX, method_anomaly="fixed_baseline", method_extreme="global_extreme",
                              threshold_percentile=95, edges=gb.edges, centres=gb.centres)
    ocean = r["mask"]
    assert 0.5 < ocean.mean() < 0.8 and not ocean[:NX].any()          # 30 % land + the whole first row
    assert r["dat_anomaly"].shape == (T, NY * NX) and (r["dat_anomaly"][:, ocean] == 0).all()
    assert np.isnan(r["dat_anomaly"][:, ~ocean]).all()
    assert r["thresholds"].dtype == np.float64 and (r["thresholds"][ocean] == gb.edges[3]).all()
    assert np.isnan(r["thresholds"][~ocean]).all()
    assert not r["extreme_events"].any()
    assert r["stats"]["n_too_low"] == int(ocean.sum()) and r["stats"]["n_too_high"] == 0


@pytest.mark.gpu
def test_cfg1_through_the_public_api(hot):
    tm, x = _field()
    lat, lon = np.linspace(-88, 88, NY), np.linspace(0, 356, NX)
    da = DataArray(x.reshape(T, NY, NX), dims=("time", "lat", "lon"), coords={"time": tm, "lat": lat, "lon": lon}, name="sst")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        ds = marex_amd.preprocess_data(da, method_anomaly="fixed_baseline", method_extreme="global_extreme", threshold_percentile=95)
    msgs = [str(w.message) for w in caught if issubclass(w.category, UserWarning)]
    assert any("Quantile values below expected range" in m for m in msgs), msgs  # detect.py:2853-2863
    gb = binning.global_bins()
    cal = calendar.build_calendar(tm)
    exp = orc.preprocess_arrays(x, cal, ny=NY, nx=NX, method_anomaly="fixed_baseline", method_extreme="global_extreme",
                                threshold_percentile=95, edges=gb.edges, centres=gb.centres)
    ocean = exp["mask"]
    assert ds.dat_anomaly.shape == (T, NY, NX) and ds.dat_anomaly.dtype == np.float32      # no trim for fixed baselines
    assert ds.thresholds.dims == ("lat", "lon") and ds.thresholds.dtype == np.float64      # detect.py:2772 float64 edges
    assert np.array_equal(ds.dat_anomaly.values.reshape(T, -1), exp["dat_anomaly"], equal_nan=True)
    assert (ds.dat_anomaly.values.reshape(T, -1)[:, ocean] == 0).all()
    assert np.array_equal(ds.thresholds.values.reshape(-1), exp["thresholds"], equal_nan=True)
    assert (ds.thresholds.values.reshape(-1)[ocean] == gb.edges[3]).all()
    assert ds.extreme_events.dtype == bool and not ds.extreme_events.values.any()
    assert np.array_equal(ds.mask.values.reshape(-1), ocean)
    assert ds.attrs["method_anomaly"] == "fixed_baseline" and ds.attrs["method_extreme"] == "global_extreme"
