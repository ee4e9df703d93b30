"""GPU: the reference's own tests, re-expressed on the reference's own fixture data (tests/golden/ref_fixtures, read with
marex_amd.zarr_io) -- plus bit parity with the oracle on that real SST field.

* tests/test_gridded_preprocessing.py:35-88  (shifting_baseline + hobday_extreme on sst_gridded.zarr)
* tests/test_gridded_preprocessing.py:735-771 (all anomaly x extreme method pairs: frequency in 2.5 % .. 7.5 %)
* the tracker's pre-processing on extremes_gridded.zarr with the parameters of tests/test_gridded_tracking.py:28-35, 85-91
The SST fixtures are the complete 40-year stores (14 611 days); every call uses the parameters the reference's test
uses (the reference's defaults -- window_year_baseline = 15 -- in test_with_all_extreme_methods).
"""
import os
import warnings

import numpy as np
import pytest

import marex_amd
import marex_amd.track_pre as tp
from marex_amd import binning, calendar, zarr_io
from marex_amd.xr_compat import DataArray
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")


@pytest.fixture(scope="module")
def sst():
    p = os.path.join(FIX, "sst_gridded.zarr")
    x = zarr_io.read_array(os.path.join(p, "to")).copy()
    assert x.shape == (14611, 20, 40)
    tm = zarr_io.decode_cf_time(zarr_io.read_array(os.path.join(p, "time")), zarr_io.array_attrs(os.path.join(p, "time")))
    x[:, 1, 1] = np.nan  # the reference's setup_class masks the 2nd lat / 2nd lon point (test_gridded_preprocessing.py:22-25)
    lat, lon = zarr_io.read_array(os.path.join(p, "lat")), zarr_io.read_array(os.path.join(p, "lon"))  # the store's own axes
    assert lat[0] == np.float32(35.125) and lon[0] == np.float32(-39.875)
    return DataArray(x, dims=("time", "lat", "lon"), coords={"time": tm, "lat": lat, "lon": lon}, name="to"), tm


def test_shifting_baseline_hobday_extreme_like_the_reference(hot, sst):
    da, tm = sst
    W = 5
    ds = marex_amd.preprocess_data(da, method_anomaly="shifting_baseline", method_extreme="hobday_extreme", threshold_percentile=95,
                                   window_year_baseline=W, smooth_days_baseline=11, window_days_hobday=3,
                                   dimensions={"time": "time", "x": "lon", "y": "lat"}, dask_chunks={"time": 25})
    for v in ("extreme_events", "dat_anomaly", "thresholds", "mask"):
        assert v in ds.data_vars
    assert ds.attrs["method_anomaly"] == "shifting_baseline" and ds.attrs["method_extreme"] == "hobday_extreme"
    assert ds.attrs["threshold_percentile"] == 95
    assert ds.extreme_events.dtype == bool and ds.dat_anomaly.dtype == np.float32
    assert ds.extreme_events.dims == ("time", "lat", "lon") and "dayofyear" in ds.thresholds.dims
    reduction = da.shape[0] - ds.extreme_events.shape[0]
    assert abs(reduction - W * 365) <= 10                       # test_gridded_preprocessing.py:71-83
    freq = float(ds.extreme_events.values.mean())
    assert 0.04 <= freq <= 0.06, freq                           # conftest.py:215-231 (5 % +- max(0.5 %, 20 % rel))
    # bit parity with the oracle on this real field (gridded default 5x5 pooling, detect.py:1451-1452)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    exp = orc.preprocess_arrays(da.values.reshape(da.shape[0], -1), cal, ny=20, nx=40, window_year_baseline=W,
                                smooth_days_baseline=11, window_days_hobday=3, window_spatial_hobday=5,
                                threshold_percentile=95.0, edges=bt.edges, centres=bt.centres)
    assert np.array_equal(ds.dat_anomaly.values.reshape(-1, 800), exp["dat_anomaly"], equal_nan=True)
    assert np.array_equal(ds.thresholds.values.reshape(800, 366), exp["thresholds"], equal_nan=True)
    assert np.array_equal(ds.extreme_events.values.reshape(-1, 800), exp["extreme_events"])
    assert not ds.mask.values[1, 1] and ds.mask.values.sum() == 799


@pytest.mark.parametrize("ma,me", [
    ("fixed_baseline", "global_extreme"), ("fixed_baseline", "hobday_extreme"),
    ("detrend_fixed_baseline", "global_extreme"), ("detrend_fixed_baseline", "hobday_extreme"),
    ("shifting_baseline", "global_extreme"), ("shifting_baseline", "hobday_extreme"),
    ("detrend_harmonic", "global_extreme"), ("detrend_harmonic", "hobday_extreme"),
])
def test_all_method_pairs_like_the_reference(hot, sst, ma, me):
    da, _ = sst
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds = marex_amd.preprocess_data(da, method_anomaly=ma, method_extreme=me, threshold_percentile=95,
                                       detrend_orders=None, window_days_hobday=11 if me == "hobday_extreme" else None,
                                       dimensions={"time": "time", "x": "lon", "y": "lat"},
                                       dask_chunks={"time": 25})  # defaults: window_year_baseline = 15, smoothing 21
    if ma == "shifting_baseline":
        assert abs((da.shape[0] - ds.extreme_events.shape[0]) - 15 * 365) <= 10
    assert "extreme_events" in ds.data_vars and ds.attrs["method_anomaly"] == ma and ds.attrs["method_extreme"] == me
    freq = float(ds.extreme_events.values.mean())
    assert 0.025 < freq < 0.075, (ma, me, freq)                 # test_gridded_preprocessing.py:767-771


@pytest.mark.parametrize("R_fill,T_fill", [(4, 0), (4, 2), (8, 2)])
def test_tracker_preprocessing_on_the_reference_extremes(hot, R_fill, T_fill):
    p = os.path.join(FIX, "extremes_gridded.zarr")
    ev = zarr_io.read_array(os.path.join(p, "extreme_events")).astype(bool)
    mask = zarr_io.read_array(os.path.join(p, "mask")).astype(bool)
    assert ev.shape == (32, 180, 360) and 0.05 < ev.mean() < 0.10
    got, stats = tp.run_preprocess(ev, mask, R_fill=R_fill, T_fill=T_fill, area_filter_quartile=0.5)
    a = orc.fill_holes(ev, mask, R_fill)
    g = orc.fill_time_gaps(a, mask, R_fill, T_fill)
    e, thr, areas, n0, n1 = orc.filter_small_objects(g, 0.5)
    assert np.array_equal(got, e)
    assert (stats[1], stats[2], stats[3]) == (n0, n1, thr) and n0 > n1 > 0
    assert not got[:, ~mask].any()


def test_device_chunk_decoder_equals_the_host_decoder(hot):
    """Compressed chunks -> HBM (LZ4 streams decoded one wave each, byte shuffle undone on placement) == host decode."""
    for sub, lead in (("sst_gridded.zarr/to", None), ("sst_gridded.zarr/to", 47), ("extremes_gridded.zarr/extreme_events", None),
                      ("extremes_gridded.zarr/extreme_events", 3), ("extremes_gridded.zarr/mask", None)):
        p = os.path.join(FIX, *sub.split("/"))
        host = zarr_io.read_array(p)
        host = host if lead is None else host[:lead]
        dev = zarr_io.read_array_to_device(p, hot, lead)
        assert tuple(dev.shape) == host.shape
        assert np.array_equal(dev.cpu().numpy(), host), sub


def test_preprocess_straight_from_the_compressed_store(hot, sst):
    """Chunks decoded in HBM feed preprocess_data without a host copy of the field: same Dataset as from the host array."""
    da_host, tm = sst
    p = os.path.join(FIX, "sst_gridded.zarr")
    lat, lon = da_host.coords["lat"].values, da_host.coords["lon"].values
    da_dev = zarr_io.open_dataarray_device(p, "to", hot, dims=("time", "lat", "lon"), coords={"lat": lat, "lon": lon})
    assert da_dev.device_tensor.is_cuda and da_dev.shape == (14611, 20, 40)
    da_dev.device_tensor[:, 1, 1] = float("nan")  # the same masked point as the host fixture
    kw = dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", threshold_percentile=95,
              window_year_baseline=5, smooth_days_baseline=11, window_days_hobday=3)
    a = marex_amd.preprocess_data(da_host, **kw)
    b = marex_amd.preprocess_data(da_dev, **kw)
    for v in ("dat_anomaly", "extreme_events", "thresholds", "mask"):
        assert np.array_equal(a[v].values, b[v].values, equal_nan=True), v
    assert np.array_equal(a.dat_anomaly.coords["time"].values, b.dat_anomaly.coords["time"].values)


def test_result_written_like_the_reference_and_read_back_in_hbm(hot, sst, tmp_path):
    """examples/batch jobs/run_detect.py:60-83: preprocess_data(...).to_zarr(store); the store is then the tracker's input.
    Chunks our writer compressed decode on the device (one wave per LZ4 stream) to the arrays that were written, and a
    device tensor is written chunk by chunk without a host copy of the whole array."""
    import torch

    da_host, _ = sst
    ds = marex_amd.preprocess_data(da_host, method_anomaly="shifting_baseline", method_extreme="hobday_extreme",
                                   threshold_percentile=95, window_year_baseline=5, smooth_days_baseline=11, window_days_hobday=3)
    store = str(tmp_path / "extremes.zarr")
    ds.to_zarr(store, mode="w")
    back = zarr_io.read_dataset(store)
    for v in ("dat_anomaly", "extreme_events", "thresholds", "mask"):
        assert back[v].dims == ds[v].dims and back[v].values.dtype == ds[v].values.dtype
        assert np.array_equal(back[v].values, ds[v].values, equal_nan=True), v
    assert np.array_equal(back.time.values.astype("datetime64[D]"), ds.dat_anomaly.coords["time"].values.astype("datetime64[D]"))
    assert back.attrs["method_anomaly"] == "shifting_baseline"
    for v in ("dat_anomaly", "extreme_events"):
        dev = zarr_io.read_array_to_device(os.path.join(store, v), hot)
        assert np.array_equal(dev.cpu().numpy().astype(ds[v].values.dtype), ds[v].values, equal_nan=True), v
    t = torch.from_numpy(ds.dat_anomaly.values).to(hot.device)
    zarr_io.write_array(os.path.join(store, "from_device"), t, (25,) + tuple(t.shape[1:]), dims=ds.dat_anomaly.dims)
    assert np.array_equal(zarr_io.read_array(os.path.join(store, "from_device")), ds.dat_anomaly.values, equal_nan=True)


@pytest.fixture(scope="module")
def sst_unstructured():
    p = os.path.join(FIX, "sst_unstructured.zarr")
    x = zarr_io.read_array(os.path.join(p, "to")).copy()
    assert x.shape == (14611, 405)
    tm = zarr_io.decode_cf_time(np.round(zarr_io.read_array(os.path.join(p, "time")) * 60.0),
                                {"units": "seconds since 1950-01-01"})  # "minutes since 1950-01-01", fractional minutes
    x[:, 2] = np.nan  # the reference's setup_class masks cell 2 (test_unstructured_preprocessing.py:28-29)
    n = x.shape[1]
    return DataArray(x, dims=("time", "ncells"), coords={"time": tm, "lat": ("ncells", np.linspace(-90, 90, n)),
                                                           "lon": ("ncells", np.linspace(-180, 180, n))}, name="to"), tm


def test_unstructured_shifting_baseline_hobday_like_the_reference(hot, sst_unstructured):
    """tests/test_unstructured_preprocessing.py:57-112 on the reference's unstructured SST fixture + oracle parity."""
    da, tm = sst_unstructured
    n = da.shape[1]
    nb = DataArray(np.random.default_rng(0).integers(0, n, (3, n)), dims=("nv", "ncells"))
    ca = DataArray(np.ones(n) * 1000.0, dims=("ncells",))
    ds = marex_amd.preprocess_data(da, method_anomaly="shifting_baseline", method_extreme="hobday_extreme", threshold_percentile=95,
                                   window_year_baseline=5, smooth_days_baseline=5, window_days_hobday=3,
                                   dimensions={"time": "time", "x": "ncells"}, coordinates={"time": "time", "x": "lon", "y": "lat"},
                                   dask_chunks={"time": 25}, neighbours=nb, cell_areas=ca)
    for v in ("extreme_events", "dat_anomaly", "thresholds", "mask", "neighbours", "cell_areas"):
        assert v in ds.data_vars, v
    assert ds.extreme_events.dtype == bool and ds.dat_anomaly.dtype == np.float32
    assert ds.extreme_events.dims == ("time", "ncells") and set(ds.thresholds.dims) == {"ncells", "dayofyear"}
    freq = float(ds.extreme_events.values.mean())
    assert 0.04 <= freq <= 0.06, freq                           # conftest.py:215-231: 5 % +- max(0.5 %, 20 % rel)
    cal = calendar.build_calendar(tm, window_year_baseline=5)
    bt = binning.hobday_bins()
    exp = orc.preprocess_arrays(da.values, cal, ny=0, nx=n, window_year_baseline=5, smooth_days_baseline=5, window_days_hobday=3,
                                window_spatial_hobday=None, threshold_percentile=95.0, edges=bt.edges, centres=bt.centres)
    assert np.array_equal(ds.dat_anomaly.values, exp["dat_anomaly"], equal_nan=True)
    assert np.array_equal(ds.thresholds.values, exp["thresholds"], equal_nan=True)
    assert np.array_equal(ds.extreme_events.values, exp["extreme_events"])


@pytest.mark.parametrize("R_fill,T_fill,q", [(2, 0, 0.1), (4, 0, 0.5), (2, 2, 0.1)])
def test_tracker_preprocessing_on_the_reference_mesh_extremes(hot, R_fill, T_fill, q):
    """The mesh branches on extremes_unstructured.zarr with the parameters of tests/test_unstructured_tracking.py:57-99."""
    p = os.path.join(FIX, "extremes_unstructured.zarr")
    ev = zarr_io.read_array(os.path.join(p, "extreme_events")).astype(bool)
    mask = zarr_io.read_array(os.path.join(p, "mask")).astype(bool)
    nb1 = zarr_io.read_array(os.path.join(p, "neighbours"))  # 1-based, as the tracker takes it
    nb0 = nb1.astype(np.int32) - 1
    a = tp.fill_holes(ev, mask, R_fill, neighbours=nb1)
    assert np.array_equal(a, orc.fill_holes_mesh(ev, mask, nb0, R_fill))
    g = tp.fill_time_gaps(a, mask, R_fill, T_fill=T_fill, neighbours=nb1)
    if T_fill:
        from scipy import ndimage as ndi
        k = T_fill + 1
        closed = ndi.binary_closing(np.pad(a, ((k, k), (0, 0))), structure=np.ones(k, dtype=bool)[:, None])[k:-k]
        assert np.array_equal(g, orc.fill_holes_mesh(closed, mask, nb0, R_fill // 2))
    try:
        e = orc.filter_small_objects_mesh(g, mask, nb0, q)
    except ValueError:
        with pytest.raises(Exception, match="No objects found"):
            tp.filter_small_objects(g, q, mask=mask, neighbours=nb1)
        return
    f, thr, big, n0, n1 = tp.filter_small_objects(g, q, mask=mask, neighbours=nb1)
    assert np.array_equal(f, e[0]) and thr == e[1] and (n0, n1) == (e[3], e[4])
