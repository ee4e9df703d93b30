"""GPU: the lazy, block-wise ``preprocess_data`` of marex_amd.dask_adapter (spatial blocks with
``time: -1`` like marEx/detect.py:2617-2620, output chunked like detect.py:785-792) gives the Dataset of the eager call, bit for
bit, for grids (latitude bands with overlap rows) and meshes (cell ranges).

With the real ``dask`` the graph is built and computed by Dask.  The build image has no Dask: there the same wiring runs on a
tiny EAGER stand-in for the four Dask entry points the adapter uses (``delayed``, ``array.from_delayed``, ``array.concatenate``,
``base.is_dask_collection`` + slicing / transpose / rechunk / compute on arrays) -- test infrastructure that lets the block
planning, the owned-row slicing, the concatenation axes, the coordinates and the validation reduction execute against the
device results."""
import sys
import types
import warnings

import numpy as np
import pytest

import marex_amd
from marex_amd import calendar, synth
from marex_amd.exceptions import DataValidationError
from marex_amd.xr_compat import DataArray

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ eager stand-in for dask
class _Lazy:
    def __init__(self, fn):
        self._fn, self._val, self._done = fn, None, False

    def compute(self):
        if not self._done:
            self._val, self._done = self._fn(), True
        return self._val


def _resolve(a):
    if isinstance(a, _Lazy):
        return a.compute()
    if isinstance(a, (list, tuple)):
        return type(a)(_resolve(v) for v in a)
    return a


class _LazyArray(_Lazy):
    def __init__(self, fn, shape, dtype):
        super().__init__(fn)
        self.shape, self.dtype, self.ndim = tuple(int(v) for v in shape), np.dtype(dtype), len(shape)

    def __getitem__(self, key):
        shape = np.empty(self.shape, dtype=np.bool_)[key].shape
        return _LazyArray(lambda: self.compute()[key], shape, self.dtype)

    def transpose(self, axes):
        return _LazyArray(lambda: self.compute().transpose(axes), [self.shape[a] for a in axes], self.dtype)

    def rechunk(self, *_a, **_k):
        return self

    def __array__(self, dtype=None):
        return np.asarray(self.compute(), dtype=dtype)


def _install_fake_dask():
    dask = types.ModuleType("dask")
    dsa = types.ModuleType("dask.array")
    base = types.ModuleType("dask.base")
    dask.delayed = lambda f, pure=False: (lambda *a: _Lazy(lambda: f(*[_resolve(v) for v in a])))
    dsa.from_delayed = lambda d, shape, dtype: _LazyArray(lambda: np.asarray(d.compute()), shape, dtype)

    def concatenate(parts, axis=0):
        shape = list(parts[0].shape)
        shape[axis] = sum(p.shape[axis] for p in parts)
        return _LazyArray(lambda: np.concatenate([p.compute() for p in parts], axis=axis), shape, parts[0].dtype)

    dsa.concatenate = concatenate
    dsa.from_array = lambda a, chunks=None: _LazyArray(lambda: a, a.shape, a.dtype)
    base.is_dask_collection = lambda x: isinstance(x, _Lazy)
    dask.array, dask.base = dsa, base
    sys.modules.update({"dask": dask, "dask.array": dsa, "dask.base": base})
    return dsa


@pytest.fixture(scope="module")
def dsa():
    try:
        import dask.array as real

        yield real
    except ImportError:
        fake = _install_fake_dask()
        yield fake
        for k in ("dask", "dask.array", "dask.base"):
            sys.modules.pop(k, None)


def gridded(ny=17, nx=16, years=12):
    tm = calendar.daily_time_axis("1991-01-01", years * 365 + 3)
    x = synth.synth_field(synth.make_tables(tm, ny, nx)).reshape(len(tm), ny, nx)
    return tm, x, {"time": tm, "lat": np.linspace(-60, 60, ny), "lon": np.linspace(0, 350, nx)}


def same(a, b):
    assert set(a.data_vars) == set(b.data_vars)
    for k in a.data_vars:
        va, vb = np.asarray(a[k].values), np.asarray(b[k].values)
        assert tuple(a[k].dims) == tuple(b[k].dims) and va.dtype == vb.dtype, k
        assert np.array_equal(va, vb, equal_nan=va.dtype.kind == "f"), k


@pytest.mark.parametrize("kw", [dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=5),
                                dict(method_anomaly="detrend_harmonic", method_extreme="global_extreme"),
                                dict(method_anomaly="fixed_baseline", method_extreme="hobday_extreme", method_percentile="exact")])
def test_lazy_bands_equal_the_eager_call(hot, dsa, kw):
    from marex_amd.dask_adapter import preprocess_data_lazy, validation_summary

    tm, x, coords = gridded()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eager = marex_amd.preprocess_data(DataArray(x, dims=("time", "lat", "lon"), coords=coords), **kw)
        lazy = preprocess_data_lazy(DataArray(dsa.from_array(x, chunks=(100, 5, 16)), dims=("time", "lat", "lon"), coords=coords),
                                    block_rows=5, **kw)
        for name in lazy.data_vars:  # still a graph: nothing ran while it was built
            assert hasattr(lazy[name].data, "compute")
            lazy[name] = DataArray(np.asarray(lazy[name].data.compute()), dims=lazy[name].dims, coords=lazy[name].coords)
        v = validation_summary(lazy)
    same(eager, lazy)
    assert v["n_ocean"] == int(np.asarray(eager["mask"].values).sum()) and v["max_invalid"] == 0
    assert np.array_equal(np.asarray(lazy["dat_anomaly"].coords["time"].values), np.asarray(eager["dat_anomaly"].coords["time"].values))
    # the attrs of the eager Dataset (detect.py:731-783), nothing in them that a zarr / netCDF writer cannot store
    assert dict(lazy.attrs) == dict(eager.attrs)
    import json

    json.dumps(dict(lazy.attrs))
    if kw["method_anomaly"] == "shifting_baseline":
        # ... and the ORACLE itself, not only the eager HIP call: the block planning, the overlap rows and the stitching are part
        # of what is compared (detect.py:2617-2620, 785-808, 2708-2732)
        from marex_amd import binning
        from oracle import marex_oracle as orc

        T, ny, nx = x.shape
        cal = calendar.build_calendar(tm, window_year_baseline=kw["window_year_baseline"])
        bt = binning.hobday_bins()
        exp = orc.preprocess_arrays(x.reshape(T, ny * nx), cal, ny=ny, nx=nx, window_year_baseline=kw["window_year_baseline"],
                                    smooth_days_baseline=21, window_days_hobday=11, window_spatial_hobday=5,
                                    threshold_percentile=95.0, edges=bt.edges, centres=bt.centres)
        assert np.array_equal(np.asarray(lazy["dat_anomaly"].values).reshape(cal.T_out, -1), exp["dat_anomaly"], equal_nan=True)
        assert np.array_equal(np.asarray(lazy["thresholds"].values).reshape(ny * nx, 366), exp["thresholds"], equal_nan=True)
        assert np.array_equal(np.asarray(lazy["extreme_events"].values).reshape(cal.T_out, -1), exp["extreme_events"])
        assert np.array_equal(np.asarray(lazy["mask"].values).reshape(-1), exp["mask"])


def test_one_threshold_range_warning_for_the_whole_field(hot, dsa):
    """Cells with a constant anomaly (sea ice) put thresholds below the table's lower bound in SEVERAL blocks: the eager call warns
    once (detect.py:2711-2730), and so does the lazy Dataset -- when its thresholds are computed, with the field's minimum."""
    from marex_amd.dask_adapter import preprocess_data_lazy

    tm, x, coords = gridded(ny=17, nx=16, years=12)
    x[:, 0:5, 2:9] = np.float32(271.35)      # a constant patch (whole 5 x 5 neighbourhoods) in the first block
    x[:, 11:16, 3:12] = np.float32(271.35)   # ... and one across the third and the fourth
    kw = dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=5)
    with warnings.catch_warnings(record=True) as w_eager:
        warnings.simplefilter("always")
        marex_amd.preprocess_data(DataArray(x, dims=("time", "lat", "lon"), coords=coords), **kw)
    low_e = [str(m.message) for m in w_eager if "below expected range" in str(m.message)]
    assert len(low_e) == 1
    with warnings.catch_warnings(record=True) as w_lazy:
        warnings.simplefilter("always")
        lazy = preprocess_data_lazy(DataArray(dsa.from_array(x, chunks=(100, 5, 16)), dims=("time", "lat", "lon"), coords=coords),
                                    block_rows=5, **kw)
        assert not [m for m in w_lazy if "below expected range" in str(m.message)]   # nothing has run yet
        np.asarray(lazy["thresholds"].data.compute())
    low_l = [str(m.message) for m in w_lazy if "below expected range" in str(m.message)]
    assert low_l == low_e  # one warning, the same text (the field's minimum, not a block's)


def test_blocks_from_a_thread_pool_on_one_device(hot):
    """Dask's threaded scheduler runs ``run_block`` tasks of the same device concurrently (blocks i and i + len(devices)): four
    unequal blocks from four threads at once, all on device 0 -- serialised inside ``run_block`` (one engine, one stream binding,
    scratch that grows with the block) -- give the eager Dataset."""
    from concurrent.futures import ThreadPoolExecutor

    from marex_amd.dask_adapter import plan_spatial_blocks, run_block

    tm, x, coords = gridded(ny=23, nx=16, years=12)
    kw = dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=5,
              dimensions={"time": "time", "x": "lon", "y": "lat"}, coordinates={"time": "time", "x": "lon", "y": "lat"})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eager = marex_amd.preprocess_data(DataArray(x, dims=("time", "lat", "lon"), coords=coords), **kw)
        shards = plan_spatial_blocks(23, 16, 2, block_rows=7)            # 7 + 6 + 5 + 5 rows: unequal, with overlap rows
        assert len({s.in1 - s.in0 for s in shards}) > 1 and len(shards) >= 3

        def one(sh):
            bc = {"lat": coords["lat"][sh.in0:sh.in1], "lon": coords["lon"]}
            return run_block(x[:, sh.in0:sh.in1], tm, sh, True, ("time", "lat", "lon"), bc, kw, 0)

        with ThreadPoolExecutor(max_workers=len(shards)) as pool:
            res = list(pool.map(one, shards))
    for name, axis in (("dat_anomaly", 1), ("extreme_events", 1), ("mask", 0), ("thresholds", 0)):
        got = np.concatenate([r[name] for r in res], axis=axis)
        exp = np.asarray(eager[name].values)
        assert got.shape == exp.shape and np.array_equal(got, exp, equal_nan=got.dtype.kind == "f"), name
    assert sum(r["_validation"]["n_ocean"] for r in res) == int(np.asarray(eager["mask"].values).sum())


def test_lazy_cell_ranges_on_a_mesh_and_validation(hot, dsa):
    from marex_amd.dask_adapter import preprocess_data_lazy, raise_if_invalid

    n = 301
    tm = calendar.daily_time_axis("1991-01-01", 12 * 365 + 3)
    x = synth.synth_field(synth.make_tables(tm, 0, n))
    coords = {"time": tm, "lat": ("ncells", np.linspace(-80, 80, n)), "lon": ("ncells", np.linspace(0, 359, n))}
    names = dict(dimensions={"time": "time", "x": "ncells"}, coordinates={"time": "time", "x": "lon", "y": "lat"})
    kw = dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=5, **names)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eager = marex_amd.preprocess_data(DataArray(x, dims=("time", "ncells"), coords=coords), **kw)
        lazy = preprocess_data_lazy(DataArray(dsa.from_array(x, chunks=(100, 64)), dims=("time", "ncells"), coords=coords), block_cells=97, **kw)
        for name in lazy.data_vars:
            lazy[name] = DataArray(np.asarray(lazy[name].data.compute()), dims=lazy[name].dims, coords=lazy[name].coords)
    same(eager, lazy)
    raise_if_invalid(lazy)  # a clean field
    # a gap in an ocean cell: the whole-field verdict of detect.py:224-279, reduced over the blocks
    ocean = np.flatnonzero(np.isfinite(x[0]))
    bad = x.copy()
    bad[100:130, ocean[5]] = np.nan
    bad[7, ocean[-3]] = np.inf
    lz = preprocess_data_lazy(DataArray(dsa.from_array(bad, chunks=(100, 64)), dims=("time", "ncells"), coords=coords), block_cells=97, **kw)
    with pytest.raises(DataValidationError, match=r"contains 31 invalid values in 2 ocean locations"):
        raise_if_invalid(lz)
    with pytest.raises(DataValidationError, match="must be Dask-backed"):
        preprocess_data_lazy(DataArray(x, dims=("time", "ncells"), coords=coords), **kw)
