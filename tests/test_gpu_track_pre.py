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
