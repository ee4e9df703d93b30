"""GPU parity: tracker pre-processing (fill_holes, fill_time_gaps; marEx/track.py:1520-1726) vs scipy.ndimage, the
library the reference itself calls through dask_image.  Boolean outputs: bit-exact."""
import numpy as np
import pytest
import torch

import marex_amd.track_pre as tp
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def _blobs(rng, T, ny, nx, density, land=0.25):
    """Random blobs with holes and specks: thresholded smoothed noise plus salt-and-pepper."""
    from scipy import ndimage as ndi

    f = ndi.gaussian_filter(rng.normal(0, 1, (T, ny, nx)), sigma=(0.7, 2.0, 2.5), mode="wrap")
    x = f > np.quantile(f, 1.0 - density)
    x ^= rng.random((T, ny, nx)) < 0.01
    lm = ndi.gaussian_filter(rng.normal(0, 1, (ny, nx)), sigma=3.0, mode="wrap")
    mask = lm > np.quantile(lm, land)
    return x & mask, mask


@pytest.mark.parametrize("R,regional", [(0, False), (1, False), (2, True), (3, False), (5, True), (8, False), (8, True), (12, False)])
def test_fill_holes_matches_scipy(hot, R, regional):
    rng = np.random.default_rng(100 + R)
    for (T, ny, nx, dens) in ((5, 37, 70, 0.15), (3, 64, 128, 0.4), (4, 20, 33, 0.05), (2, 90, 181, 0.25)):
        x, mask = _blobs(rng, T, ny, nx, dens)
        exp = orc.fill_holes(x, mask, R, regional)
        got = hot.fill_holes(torch.from_numpy(x.reshape(T, -1).astype(np.uint8)).to(hot.device),
                             torch.from_numpy(mask.reshape(-1).astype(np.uint8)).to(hot.device), ny, nx, R, regional)
        hot.sync()
        assert np.array_equal(got.cpu().numpy().astype(bool).reshape(T, ny, nx), exp), (R, regional, T, ny, nx)


@pytest.mark.parametrize("T_fill,R", [(0, 4), (2, 8), (2, 3), (4, 6), (6, 2)])
def test_fill_time_gaps_matches_scipy(hot, T_fill, R):
    rng = np.random.default_rng(7 + T_fill + R)
    T, ny, nx = 30, 40, 76
    x, mask = _blobs(rng, T, ny, nx, 0.2)
    x[rng.random(T) < 0.3] = False  # whole timesteps missing: gaps to close
    exp = orc.fill_time_gaps(x, mask, R, T_fill)
    got = hot.fill_time_gaps(torch.from_numpy(x.reshape(T, -1).astype(np.uint8)).to(hot.device),
                             torch.from_numpy(mask.reshape(-1).astype(np.uint8)).to(hot.device), ny, nx, R, T_fill)
    hot.sync()
    assert np.array_equal(got.cpu().numpy().astype(bool).reshape(T, ny, nx), exp)


def test_api_and_errors(hot):
    rng = np.random.default_rng(3)
    x, mask = _blobs(rng, 6, 24, 48, 0.2)
    a = tp.fill_holes(x, mask, 4)
    assert a.dtype == bool and np.array_equal(a, orc.fill_holes(x, mask, 4))
    b = tp.fill_time_gaps(a, mask, 4, T_fill=2)
    assert np.array_equal(b, orc.fill_time_gaps(a, mask, 4, 2))
    with pytest.raises(Exception, match="T_fill must be even"):
        tp.fill_time_gaps(x, mask, 4, T_fill=3)
    with pytest.raises(Exception, match="gridded"):
        tp.fill_holes(x[:, 0], mask, 4)


def _same_partition(a, b):
    """Two label fields describe the same objects (labels may be numbered differently)."""
    if not np.array_equal(a > 0, b > 0):
        return False
    fa, fb = a[a > 0], b[b > 0]
    pairs = np.unique(np.stack([fa, fb], axis=1), axis=0)
    return pairs.shape[0] == np.unique(fa).size == np.unique(fb).size


@pytest.mark.parametrize("regional", [False, True])
def test_connected_components_match_scipy(hot, regional):
    rng = np.random.default_rng(11)
    for (T, ny, nx, dens) in ((4, 33, 70, 0.3), (3, 64, 128, 0.55), (5, 9, 17, 0.5), (2, 1, 40, 0.5), (2, 40, 1, 0.5), (2, 5, 2, 0.6)):
        x, _ = _blobs(rng, T, ny, nx, dens, land=0.0) if min(ny, nx) > 4 else (rng.random((T, ny, nx)) < dens, None)
        x[:, :, 0] |= rng.random((T, ny)) < 0.3   # plenty of objects on the seam
        x[:, :, -1] |= rng.random((T, ny)) < 0.3
        exp = orc.label_objects_2d(x, wrap_x=not regional)
        r = hot.label_objects_2d(torch.from_numpy(x.reshape(T, -1).astype(np.uint8)).to(hot.device), ny, nx, wrap_x=not regional)
        hot.sync()
        got = r["labels"].cpu().numpy().reshape(T, ny, nx)
        assert _same_partition(got, exp), (regional, T, ny, nx)
        areas = r["areas"].cpu().numpy().reshape(-1)
        assert np.array_equal(np.sort(areas[areas > 0]), np.sort(np.bincount(exp.reshape(-1))[1:]))
        assert int(exp.max()) == int((areas > 0).sum())


@pytest.mark.parametrize("q,absolute,regional", [(0.5, None, False), (0.25, None, True), (0.9, None, False), (0.5, 12, False), (0.0, None, False)])
def test_filter_small_objects_matches_the_oracle(hot, q, absolute, regional):
    rng = np.random.default_rng(23)
    T, ny, nx = 6, 48, 96
    x, _ = _blobs(rng, T, ny, nx, 0.25, land=0.0)
    exp, thr, areas, n0, n1 = orc.filter_small_objects(x, q, absolute, regional)
    got, g_thr, g_areas, g0, g1 = tp.filter_small_objects(x, q, absolute, regional)
    assert g_thr == thr and g0 == n0 and g1 == n1
    assert np.array_equal(np.sort(g_areas), np.sort(areas))
    assert np.array_equal(got, exp)
    ids, n = tp.identify_objects_2d(x, regional)
    assert n == n0 and _same_partition(ids, orc.label_objects_2d(x, wrap_x=not regional))


def _tri_mesh(rng, C):
    """A random 3-regular-ish neighbour table: cells on a ring with one random chord each, some neighbours missing,
    some listed from one end only (the reference treats listed pairs as undirected for clustering)."""
    nb = np.full((3, C), -1, dtype=np.int32)
    nb[0] = (np.arange(C) + 1) % C
    nb[1] = (np.arange(C) - 1) % C
    nb[2] = rng.integers(0, C, C)
    nb[2][rng.random(C) < 0.2] = -1
    nb[1][rng.random(C) < 0.05] = -1
    return nb


@pytest.mark.parametrize("R", [0, 1, 3, 6])
def test_mesh_fill_holes_matches_the_sparse_matrix_form(hot, R):
    rng = np.random.default_rng(40 + R)
    T, C = 7, 613
    nb = _tri_mesh(rng, C)
    x = rng.random((T, C)) < 0.2
    mask = rng.random(C) > 0.15
    x &= mask
    exp = orc.fill_holes_mesh(x, mask, nb, R)
    got = hot.fill_holes_mesh(torch.from_numpy(x.astype(np.uint8)).to(hot.device), torch.from_numpy(mask.astype(np.uint8)).to(hot.device),
                              torch.from_numpy(nb).to(hot.device), R)
    hot.sync()
    assert np.array_equal(got.cpu().numpy().astype(bool), exp)


@pytest.mark.parametrize("q,absolute", [(0.5, None), (0.2, None), (0.5, 30)])
def test_mesh_components_and_filter_match_scipy(hot, q, absolute):
    rng = np.random.default_rng(77)
    T, C = 5, 4000
    nb = _tri_mesh(rng, C)
    nb[2] = np.where(rng.random(C) < 0.7, -1, nb[2])  # mostly a ring: long clusters
    x = rng.random((T, C)) < 0.93
    mask = rng.random(C) > 0.02
    exp, thr, big, n0, n1 = orc.filter_small_objects_mesh(x, mask, nb, q, absolute)
    r = hot.filter_small_objects_mesh(torch.from_numpy(x.astype(np.uint8)).to(hot.device),
                                      torch.from_numpy(mask.astype(np.uint8)).to(hot.device), torch.from_numpy(nb).to(hot.device),
                                      q, absolute)
    hot.sync()
    assert r["area_threshold"] == thr and r["n_before"] == n0 and r["n_after"] == n1
    assert np.array_equal(np.sort(r["object_areas"].cpu().numpy()), np.sort(big))
    assert np.array_equal(r["filtered"].cpu().numpy().astype(bool), exp)
    assert _same_partition(r["labels"].cpu().numpy(), orc.label_objects_mesh(x, mask, nb))


def test_mesh_api(hot):
    rng = np.random.default_rng(5)
    T, C = 9, 900
    nb0 = _tri_mesh(rng, C)
    nb1 = nb0 + 1  # the reference's 1-based table, 0 = none
    mask = rng.random(C) > 0.1
    x = (rng.random((T, C)) < 0.3) & mask
    a = tp.fill_holes(x, mask, 2, neighbours=nb1)
    assert np.array_equal(a, orc.fill_holes_mesh(x, mask, nb0, 2))
    b = tp.fill_time_gaps(a, mask, 4, T_fill=2, neighbours=nb1)
    from scipy import ndimage as ndi
    closed = ndi.binary_closing(np.pad(a, ((3, 3), (0, 0))), structure=np.ones(3, dtype=bool)[:, None])[3:-3]
    assert np.array_equal(b, orc.fill_holes_mesh(closed, mask, nb0, 2))
    f, thr, big, n0, n1 = tp.filter_small_objects(b, 0.5, mask=mask, neighbours=nb1)
    e = orc.filter_small_objects_mesh(b, mask, nb0, 0.5)
    assert np.array_equal(f, e[0]) and thr == e[1] and (n0, n1) == (e[3], e[4])


def test_run_preprocess_chain(hot):
    """fill_holes -> fill_time_gaps -> filter_small_objects in one call (track.py:1283-1360) vs the oracle chain."""
    rng = np.random.default_rng(9)
    T, ny, nx = 16, 60, 120
    x, mask = _blobs(rng, T, ny, nx, 0.12)
    got, stats = tp.run_preprocess(x, mask, R_fill=3, T_fill=2, area_filter_quartile=0.5)
    a = orc.fill_holes(x, mask, 3)
    g = orc.fill_time_gaps(a, mask, 3, 2)
    e, thr, areas, n0, n1 = orc.filter_small_objects(g, 0.5)
    assert np.array_equal(got, e)
    assert stats[1] == n0 and stats[2] == n1 and stats[3] == thr and stats[0] == areas.sum()
    assert abs(stats[4] - areas[areas > thr].sum() / areas.sum()) < 1e-12
    assert abs(stats[5] - x.sum() / e.sum()) < 1e-12


@pytest.mark.parametrize("regional", [False, True])
def test_connected_components_dense_fields(hot, regional):
    """Dense noise, full rows and an all-True image: long overlaps between rows, including ones that go all the way
    round a periodic row (the union-per-overlap shortcut must still link them)."""
    rng = np.random.default_rng(31)
    for (T, ny, nx, dens) in ((3, 40, 130, 0.7), (3, 33, 64, 0.9), (2, 20, 70, 1.0), (4, 25, 129, 0.5)):
        x = rng.random((T, ny, nx)) < dens
        x[0, 3:6, :] = True          # full rows, stacked
        x[-1, :, 0] = True
        x[-1, :, -1] = True
        exp = orc.label_objects_2d(x, wrap_x=not regional)
        r = hot.label_objects_2d(torch.from_numpy(x.reshape(T, -1).astype(np.uint8)).to(hot.device), ny, nx, wrap_x=not regional)
        hot.sync()
        assert _same_partition(r["labels"].cpu().numpy().reshape(T, ny, nx), exp), (regional, T, ny, nx, dens)
