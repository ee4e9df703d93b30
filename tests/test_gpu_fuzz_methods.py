"""GPU: seeded random draws over ALL method pairs of preprocess_data (anomaly x extreme x percentile method, detrend orders,
reference periods, windows, percentiles, precision / max_anomaly, grid kind and shape, block-wise execution) through the
public API against the oracle, bit for bit.  Complements tests/test_gpu_fuzz.py, which stresses the shifting_baseline +
hobday_extreme kernels with damaged fields."""
import os
import warnings

import numpy as np
import pytest

import marex_amd
from marex_amd import binning, calendar, synth
from marex_amd.xr_compat import DataArray
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu

_lo, _hi = (int(v) for v in os.environ.get("MAREX_FUZZ_METHOD_SEEDS", "0:32").split(":"))


@pytest.mark.parametrize("seed", list(range(_lo, _hi)))
def test_random_method_pair_matches_the_oracle(hot, monkeypatch, seed):
    rng = np.random.default_rng(1000 + seed)
    ma = str(rng.choice(["shifting_baseline", "fixed_baseline", "detrend_harmonic", "detrend_fixed_baseline"]))
    me = str(rng.choice(["hobday_extreme", "global_extreme"]))
    mp = str(rng.choice(["approximate", "exact"]))
    unstructured = rng.random() < 0.3
    W = int(rng.choice([2, 3, 5, 7]))
    years = W + int(rng.integers(4, 9))
    start = f"{int(rng.integers(1960, 2010))}-{int(rng.integers(1, 13)):02d}-01"
    periods = years * 365 + int(rng.integers(0, 200))
    tm = calendar.daily_time_axis(start, periods)
    if unstructured:
        ny, nx = 0, int(rng.integers(5, 300))
        x = synth.synth_field(synth.make_tables(tm, 0, nx, 777 + seed, unstructured=True))
        da = DataArray(x, dims=("time", "ncells"), coords={"time": tm, "lon": ("ncells", np.linspace(0, 359, nx)),
                                                            "lat": ("ncells", np.linspace(-80, 80, nx))})
        extra = dict(dimensions={"time": "time", "x": "ncells"}, coordinates={"time": "time", "x": "lon", "y": "lat"})
        ws = None
    else:
        ny, nx = int(rng.integers(3, 14)), int(rng.integers(4, 20))
        x = synth.synth_field(synth.make_tables(tm, ny, nx, 777 + seed))
        da = DataArray(x.reshape(periods, ny, nx), dims=("time", "lat", "lon"),
                       coords={"time": tm, "lat": np.linspace(-60, 60, ny), "lon": np.linspace(0, 350, nx)})
        extra = {}
        ws = None if rng.random() < 0.5 else int(rng.choice([1, 3, 5]))
    pct = float(rng.choice([90.0, 95.0, 99.0] if mp == "approximate" else [75.0, 90.0, 95.0, 99.0]))
    orders = [[1], [1, 2], [1, 2, 3]][int(rng.integers(0, 3))]
    fzm = bool(rng.random() < 0.7)
    y0 = int(str(tm[0])[:4])
    ref = None if rng.random() < 0.5 else (y0 + 1, y0 + int(rng.integers(2, years - 1)))
    S = int(rng.choice([21, 11, 31]))
    wd = int(rng.choice([3, 5, 11, 21]))
    prec, maxa = (0.01, 5.0) if rng.random() < 0.6 else (float(rng.choice([0.02, 0.05])), float(rng.choice([4.0, 8.0])))
    kw = dict(method_anomaly=ma, method_extreme=me, method_percentile=mp, threshold_percentile=pct, window_year_baseline=W,
              smooth_days_baseline=S, window_days_hobday=wd, detrend_orders=orders, force_zero_mean=fzm)
    if ma in ("fixed_baseline", "detrend_fixed_baseline") and ref is not None:
        kw["reference_period"] = ref
    if me == "hobday_extreme" and mp == "approximate" and ws is not None and not unstructured:
        kw["window_spatial_hobday"] = ws
    if mp == "approximate":
        kw.update(precision=prec, max_anomaly=maxa)
    monkeypatch.setenv("MAREX_BLOCKS", str(int(rng.choice([1, 1, 2, 3]))))
    hot.hobday_path = (None, "tails", "bins")[seed % 3]  # representation of the dayofyear histograms (same results)
    try:
        with hot.ctx.options(**({"MASK_BINS": 1} if seed % 2 else {})), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ds = marex_amd.preprocess_data(da, **kw, **extra)
    finally:
        hot.hobday_path = None

    cal = calendar.build_calendar(tm, window_year_baseline=W if ma == "shifting_baseline" else None)
    model = pmodel = None
    if ma.startswith("detrend"):
        model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), orders, ma == "detrend_harmonic")
    edges = centres = None
    if mp == "approximate":
        tab = binning.global_bins(prec, maxa) if me == "global_extreme" else binning.hobday_bins(prec, maxa)
        edges, centres = tab.edges, tab.centres
    ws_eff = kw.get("window_spatial_hobday")
    exp = orc.preprocess_arrays(
        x.reshape(periods, -1), cal, ny=ny, nx=nx, method_anomaly=ma, method_extreme=me, method_percentile=mp,
        threshold_percentile=pct, window_year_baseline=W, smooth_days_baseline=S, window_days_hobday=wd,
        window_spatial_hobday=ws_eff, edges=edges, centres=centres, model=model, pmodel=pmodel, force_zero_mean=fzm,
        reference_period=kw.get("reference_period"),
    )
    C = x.reshape(periods, -1).shape[1]
    case = f"{ma}/{me}/{mp} ny={ny} nx={nx} W={W} S={S} wd={wd} ws={ws_eff} p={pct} orders={orders} ref={kw.get('reference_period')}"
    assert np.array_equal(ds.dat_anomaly.values.reshape(-1, C), exp["dat_anomaly"], equal_nan=True), case
    assert np.array_equal(ds.mask.values.reshape(-1), exp["mask"]), case
    thr = ds.thresholds.values
    if me == "hobday_extreme" and mp == "approximate":
        assert np.array_equal(thr.reshape(C, 366), exp["thresholds"], equal_nan=True), case
    elif me == "hobday_extreme":
        assert np.array_equal(thr.reshape(366, C), exp["thresholds"], equal_nan=True), case
    else:
        assert thr.dtype == np.float64 and np.array_equal(thr.reshape(C), exp["thresholds"], equal_nan=True), case
    assert np.array_equal(ds.extreme_events.values.reshape(-1, C), exp["extreme_events"]), case
