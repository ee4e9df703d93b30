"""CPU: planning of the spatial blocks preprocess_data cuts a field into (marex_amd.detect.plan_blocks; the device-side
counterpart of the reference's space-chunked, ``time: -1`` Dask layout, detect.py:2617-2620)."""
import numpy as np

from marex_amd import calendar, detect
from marex_amd.xr_compat import DataArray


def _field(ny, nx, T=40, gridded=True):
    tm = calendar.daily_time_axis("2001-01-01", T)
    if gridded:
        da = DataArray(np.zeros((T, ny, nx), np.float32), dims=("time", "lat", "lon"),
                       coords={"time": tm, "lat": np.arange(ny), "lon": np.arange(nx)})
        d = {"time": "time", "x": "lon", "y": "lat"}
        return detect._Field(da, d, d)
    da = DataArray(np.zeros((T, nx), np.float32), dims=("time", "ncells"), coords={"time": tm})
    return detect._Field(da, {"time": "time", "x": "ncells"}, {"time": "time", "x": "lon", "y": "lat"})


def test_forced_bands_cover_the_grid_with_overlap_rows(monkeypatch):
    f = _field(23, 7)
    for n in (1, 2, 5, 23, 99):
        monkeypatch.setenv("MAREX_BLOCKS", str(n))
        sh = detect.plan_blocks(f, None, 2, 1)
        assert len(sh) == min(n, 23)
        assert sh[0].own0 == 0 and sh[-1].own1 == 23
        assert all(a.own1 == b.own0 for a, b in zip(sh, sh[1:]))
        assert all(s.in0 == max(0, s.own0 - 2) and s.in1 == min(23, s.own1 + 2) for s in sh)
        for s in sh:  # a block is a contiguous range of the flattened cell axis
            fb = f.block(s)
            assert (fb.c0, fb.c1) == (s.in0 * 7, s.in1 * 7) and fb.shape == (40, (s.in1 - s.in0) * 7)
            assert (fb.ny, fb.nx) == (s.in1 - s.in0, 7)
            own = s.own_cell_slice()
            assert fb.c0 + own.start == s.own0 * 7 and fb.c0 + own.stop == s.own1 * 7


def test_forced_cell_ranges_on_a_mesh(monkeypatch):
    f = _field(0, 101, gridded=False)
    monkeypatch.setenv("MAREX_BLOCKS", "4")
    sh = detect.plan_blocks(f, None, 2, 1)
    assert [s.own0 for s in sh] == [0, 26, 51, 76] and sh[-1].own1 == 101
    assert all((s.in0, s.in1) == (s.own0, s.own1) for s in sh)  # no pooling on meshes: no overlap
    fb = f.block(sh[2])
    assert (fb.ny, fb.nx, fb.c0, fb.c1) == (0, 25, 51, 76)
