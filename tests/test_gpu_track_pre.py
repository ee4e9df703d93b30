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
