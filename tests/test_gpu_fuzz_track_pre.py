"""GPU: seeded random draws over the tracker pre-processing stage (fill_holes -> fill_time_gaps -> filter_small_objects and
the 2-D labelling; track.py:1283-1360, 1520-1727, 1912-2049) on grids and on meshes against the scipy-based oracle:
random shapes (down to single rows / columns), densities, radii, gap lengths, quantiles, periodic or regional."""
import os

import numpy as np
import pytest
import torch

import marex_amd.track_pre as tp
from marex_amd.exceptions import ProcessingError
from oracle import marex_oracle as orc
from tests.test_gpu_track_pre import _blobs, _same_partition, _tri_mesh

pytestmark = pytest.mark.gpu

_lo, _hi = (int(v) for v in os.environ.get("MAREX_FUZZ_TRACK_SEEDS", "0:24").split(":"))


@pytest.mark.parametrize("seed", list(range(_lo, _hi)))
def test_random_gridded_chain(hot, seed):
    rng = np.random.default_rng(5000 + seed)
    T = int(rng.integers(1, 14))
    ny, nx = int(rng.integers(1, 70)), int(rng.integers(1, 140))
    dens = float(rng.choice([0.02, 0.1, 0.25, 0.5, 0.8]))
    regional = bool(rng.random() < 0.4)
    if min(ny, nx) > 6:
        x, mask = _blobs(rng, T, ny, nx, dens, land=float(rng.choice([0.0, 0.2, 0.5])))
    else:
        x, mask = rng.random((T, ny, nx)) < dens, rng.random((ny, nx)) > 0.2
        x &= mask
    if T > 3:
        x[rng.random(T) < 0.25] = False
    R = int(rng.choice([0, 1, 2, 3, 5, 8]))
    T_fill = int(rng.choice([0, 2, 4]))
    q = float(rng.choice([0.0, 0.25, 0.5, 0.9]))
    case = (seed, T, ny, nx, dens, regional, R, T_fill, q)
    a = orc.fill_holes(x, mask, R, regional)
    assert np.array_equal(tp.fill_holes(x, mask, R, regional_mode=regional), a), case
    g = orc.fill_time_gaps(a, mask, R, T_fill, regional)
    assert np.array_equal(tp.fill_time_gaps(a, mask, R, T_fill, regional_mode=regional), g), case
    exp = orc.label_objects_2d(g, wrap_x=not regional)
    ids, n = tp.identify_objects_2d(g, regional)
    assert n == int(exp.max()) and _same_partition(ids, exp), case
    if exp.max() > 0:
        e, thr, areas, n0, n1 = orc.filter_small_objects(g, q, None, regional)
        f, g_thr, g_areas, g0, g1 = tp.filter_small_objects(g, q, None, regional)
        assert g_thr == thr and (g0, g1) == (n0, n1) and np.array_equal(f, e), case
        assert np.array_equal(np.sort(g_areas), np.sort(areas)), case


@pytest.mark.parametrize("seed", list(range(_lo, _hi)))
def test_random_mesh_chain(hot, seed):
    rng = np.random.default_rng(9000 + seed)
    T, C = int(rng.integers(1, 20)), int(rng.integers(4, 3000))
    nb0 = _tri_mesh(rng, C)
    if rng.random() < 0.5:
        nb0[2] = np.where(rng.random(C) < 0.7, -1, nb0[2])
    nb1 = nb0 + 1  # the reference's tables are 1-based (0 = no neighbour)
    mask = rng.random(C) > float(rng.choice([0.0, 0.1, 0.4]))
    x = (rng.random((T, C)) < float(rng.choice([0.05, 0.3, 0.6, 0.95]))) & mask
    R = int(rng.choice([0, 1, 2, 4]))
    q = float(rng.choice([0.0, 0.3, 0.5, 0.8]))
    case = (seed, T, C, R, q)
    a = orc.fill_holes_mesh(x, mask, nb0, R)
    assert np.array_equal(tp.fill_holes(x, mask, R, neighbours=nb1), a), case
    exp = orc.label_objects_mesh(a, mask, nb0)
    dev = lambda v, dt: torch.from_numpy(np.ascontiguousarray(v, dtype=dt)).to(hot.device)  # noqa: E731
    sizes = np.bincount(exp.reshape(-1))[1:]
    if not (sizes > 50).any():  # track.py:1819-1830: nothing large enough to take a percentile of -> an error, both sides
        with pytest.raises(ValueError):
            orc.filter_small_objects_mesh(a, mask, nb0, q)
        with pytest.raises(ProcessingError, match="No objects found"):
            hot.filter_small_objects_mesh(dev(a, np.uint8), dev(mask, np.uint8), dev(nb0, np.int32), q, None)
        q_abs = 3.0  # the absolute threshold lowers the bar to 5 cells
        if not (sizes > 5).any():
            return
        e = orc.filter_small_objects_mesh(a, mask, nb0, q, q_abs)
        r = hot.filter_small_objects_mesh(dev(a, np.uint8), dev(mask, np.uint8), dev(nb0, np.int32), q, q_abs)
    else:
        e = orc.filter_small_objects_mesh(a, mask, nb0, q)
        r = hot.filter_small_objects_mesh(dev(a, np.uint8), dev(mask, np.uint8), dev(nb0, np.int32), q, None)
        f, thr, _, n0, n1 = tp.filter_small_objects(a, q, mask=mask, neighbours=nb1)
        assert np.array_equal(f, e[0]) and thr == e[1] and (n0, n1) == (e[3], e[4]), case
    hot.sync()
    assert _same_partition(r["labels"].cpu().numpy().reshape(T, C), exp), case
    assert r["area_threshold"] == e[1] and (r["n_before"], r["n_after"]) == (e[3], e[4]), case
    assert np.array_equal(r["filtered"].cpu().numpy().astype(bool), e[0]), case
