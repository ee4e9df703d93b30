"""GPU: block-wise execution of preprocess_data (spatial blocks sized to the free HBM -- the device-side counterpart of the
reference's Dask layout for this path: space chunked, ``time: -1``, detect.py:2617-2620) gives the very same Dataset as a
single block.  The reference pins the analogous property for its chunking (identical extreme counts for different chunk
sizes, tests/test_integration.py:219-226); here every variable is compared bit for bit."""
import warnings

import numpy as np
import pytest

import marex_amd
from marex_amd import calendar, synth
from marex_amd.exceptions import DataValidationError
from marex_amd.xr_compat import DataArray

pytestmark = pytest.mark.gpu


def gridded(ny=13, nx=14, years=12):
    tm = calendar.daily_time_axis("1990-01-01", years * 365 + 3)
    x = synth.synth_field(synth.make_tables(tm, ny, nx)).reshape(len(tm), ny, nx)
    return DataArray(x, dims=("time", "lat", "lon"), coords={"time": tm, "lat": np.linspace(-60, 60, ny), "lon": np.linspace(0, 350, nx)})


def mesh(n=301, years=12):
    tm = calendar.daily_time_axis("1990-01-01", years * 365 + 3)
    x = synth.synth_field(synth.make_tables(tm, 0, n))
    return DataArray(x, dims=("time", "ncells"), coords={"time": tm, "lat": ("ncells", np.linspace(-80, 80, n)), "lon": ("ncells", np.linspace(0, 359, n))})


def run(da, monkeypatch, blocks, **kw):
    monkeypatch.setenv("MAREX_BLOCKS", str(blocks))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ds = marex_amd.preprocess_data(da, **kw)
    return ds, sorted(str(x.message)[:40] for x in w)


def same(a, b):
    assert set(a.data_vars) == set(b.data_vars)
    for k in a.data_vars:
        assert a[k].dims == b[k].dims and a[k].values.dtype == b[k].values.dtype, k
        assert np.array_equal(a[k].values, b[k].values, equal_nan=a[k].values.dtype.kind == "f"), k
    assert a.attrs == b.attrs


CASES = [
    dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=5),
    dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=4, smooth_days_baseline=11,
         window_days_hobday=5, window_spatial_hobday=3),
    dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=5, method_percentile="exact"),
    dict(method_anomaly="shifting_baseline", method_extreme="global_extreme", window_year_baseline=5),
    dict(method_anomaly="detrend_harmonic", method_extreme="hobday_extreme", std_normalise=True),
    dict(method_anomaly="detrend_harmonic", method_extreme="global_extreme", method_percentile="exact", detrend_orders=[1, 2]),
    dict(method_anomaly="fixed_baseline", method_extreme="hobday_extreme", reference_period=(1992, 1998)),
    dict(method_anomaly="detrend_fixed_baseline", method_extreme="global_extreme"),
]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: f"{k['method_anomaly']}-{k['method_extreme']}-{k.get('method_percentile', 'approx')}")
def test_blocks_equal_single_block_gridded(hot, monkeypatch, kw):
    da = gridded()
    ref, wref = run(da, monkeypatch, 1, **kw)
    for blocks in (2, 5, 13):
        got, wgot = run(da, monkeypatch, blocks, **kw)
        same(ref, got)
        assert wgot == wref  # the threshold-range warnings are raised once for the whole field, not once per block


@pytest.mark.parametrize("kw", [CASES[0], CASES[3], CASES[4]], ids=["shifting-hobday", "shifting-global", "detrend-stn"])
def test_blocks_equal_single_block_mesh(hot, monkeypatch, kw):
    da = mesh()
    dims, coords = {"time": "time", "x": "ncells"}, {"time": "time", "x": "lon", "y": "lat"}
    ref, _ = run(da, monkeypatch, 1, dimensions=dims, coordinates=coords, **kw)
    for blocks in (3, 10):
        got, _ = run(da, monkeypatch, blocks, dimensions=dims, coordinates=coords, **kw)
        same(ref, got)


def test_validation_error_counts_cover_all_blocks(hot, monkeypatch):
    """detect.py:224-279: the counts in the message are those of the whole field however it is cut."""
    da = gridded()
    x = da.values.copy()
    ocean = np.argwhere(np.isfinite(x[0]))
    (j0, i0), (j1, i1) = ocean[1], ocean[-2]
    x[100:103, j0, i0] = np.nan
    x[7, j1, i1] = np.inf
    bad = DataArray(x, dims=da.dims, coords={k: v for k, v in da.coords.items()})
    msgs = []
    for blocks in (1, 4):
        monkeypatch.setenv("MAREX_BLOCKS", str(blocks))
        with pytest.raises(DataValidationError) as e:
            marex_amd.preprocess_data(bad, window_year_baseline=5)
        msgs.append(str(e.value))
    assert "contains 4 invalid values in 2 ocean locations" in msgs[0]
    assert msgs[0] == msgs[1]


def test_automatic_plan_is_one_block_when_the_field_fits(hot, monkeypatch):
    from marex_amd import detect

    monkeypatch.delenv("MAREX_BLOCKS", raising=False)
    da = gridded()
    f = detect._Field(da, {"time": "time", "x": "lon", "y": "lat"}, {"time": "time", "x": "lon", "y": "lat"})
    assert len(detect.plan_blocks(f, hot, 2, 11 * f.shape[0])) == 1
    # a per-cell footprint that cannot fit forces bands; overlap rows are ingested on interior sides only
    free = __import__("torch").cuda.mem_get_info(hot.device)[0]
    shards = detect.plan_blocks(f, hot, 2, int(free * 0.75 / (7 * 14)))  # room for 7 rows: <= 3 owned + 4 overlap
    assert len(shards) > 1 and shards[0].in0 == 0 and shards[-1].in1 == 13
    assert all(s.in0 == max(0, s.own0 - 2) and s.in1 == min(13, s.own1 + 2) for s in shards)
    assert [s.own0 for s in shards[1:]] == [s.own1 for s in shards[:-1]]


def test_block_budget_is_shared_by_the_engines_of_a_card(hot, monkeypatch):
    """devices=[0, 0]: two engines hold a block each on the same card, so a block may take half of what one engine could."""
    from marex_amd import detect

    monkeypatch.delenv("MAREX_BLOCKS", raising=False)
    da = gridded()
    f = detect._Field(da, {"time": "time", "x": "lon", "y": "lat"}, {"time": "time", "x": "lon", "y": "lat"})
    free = __import__("torch").cuda.mem_get_info(hot.device)[0]
    per_cell = int(free * 0.75 / (13 * 14))  # the whole 13-row field fits one engine's budget ...
    assert len(detect.plan_blocks(f, hot, 2, per_cell)) == 1
    second = detect.get_engine(0, 1)
    two = detect.plan_blocks(f, hot, 2, per_cell, min_blocks=2, engines=[hot, second])
    assert len(two) >= 2 and max(s.cells_in for s in two) * per_cell <= free * 0.8 / 2  # ... but not half of it


def test_pinned_pipe_round_trips_strided_arrays(hot):
    """marex_amd.transfer.PinnedPipe: chunked, multi-threaded staging in both directions, strided host views, several
    dtypes, more chunks than staging buffers, a row larger than a buffer."""
    import torch

    from marex_amd.transfer import PinnedPipe

    pipe = PinnedPipe(hot.device, chunk_bytes=1 << 20, nbuf=3, threads=4)
    rng = np.random.default_rng(8)
    big = rng.normal(size=(700, 5000)).astype(np.float32)
    for view in (big, big[:, 100:4100], big[5:, ::1][:, 7:8], big[:0], big[:, :0]):
        dev = pipe.upload(view, np.float32)
        assert tuple(dev.shape) == view.shape and np.array_equal(dev.cpu().numpy(), view)
        back = np.full((view.shape[0], view.shape[1] + 3), -1, dtype=np.float32)
        pipe.download(dev, back[:, 1:-2])
        assert np.array_equal(back[:, 1:-2], view) and (back[:, 0] == -1).all() and (back[:, -2:] == -1).all()
    f64 = rng.normal(size=(300, 1000))
    assert np.array_equal(pipe.upload(f64, np.float32).cpu().numpy(), f64.astype(np.float32))  # cast on the way in
    u8 = (rng.random((3000, 2048)) < 0.3).astype(np.uint8)
    out = np.zeros_like(u8)
    pipe.download(torch.from_numpy(u8).to(hot.device)[:, 8:2000], out[:, 8:2000])  # strided device view
    assert np.array_equal(out[:, 8:2000], u8[:, 8:2000]) and not out[:, :8].any()
    v = torch.arange(100000, dtype=torch.float64, device=hot.device)
    o1 = np.empty(100000)
    pipe.download(v, o1)
    assert np.array_equal(o1, np.arange(100000.0))
    wide = rng.normal(size=(3, 400000)).astype(np.float32)  # one row > chunk_bytes
    assert np.array_equal(pipe.upload(wide, np.float32).cpu().numpy(), wide)
    o2 = np.empty_like(wide)
    pipe.download(torch.from_numpy(wide).to(hot.device), o2)
    assert np.array_equal(o2, wide)
