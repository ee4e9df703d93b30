"""CPU: how far is the oracle's arithmetic CONTRACT (C1 / C2, oracle/marex_oracle.py) from what the reference's
third-party kernels compute for the same stage?

The reference smooths with ``da.rolling(time=S, center=True).mean()`` (marEx/detect.py:1810-1812), which xarray hands to
bottleneck's ``move_mean`` per Dask time chunk with ``S // 2`` ghost steps on either side (``dask_chunks = {"time": 25}``,
detect.py:532-535), and forms the climatology with flox's grouped ``nanmean`` (detect.py:1659-1669).  Neither package is
under /root/reference nor installed here (both unpinned, pyproject.toml:28-43), so their published algorithms are
restated below:

* bottleneck ``move_mean`` (move_template.c, any 1.3.x / 1.4.x): a running sum in the INPUT dtype (float32),
  ``asum += ai`` while the window fills, then ``asum += ai - aold`` per step, result ``asum * (1 / count)``;
* flox ``nanmean``: per-group float32 sums whose order depends on the chunking -- bracketed here by summing the window
  in DESCENDING year order (the contract sums in ascending order).

The oracle re-sums every window from scratch (C1, C2), which no chunking can perturb; the reference's running sum
carries its rounding error along a chunk.  north_star asks for "fp32 anomalies within 1e-5 relative" of the reference:
this test bounds the distance between the two formulations on the synthetic SST field (degrees C) and on
the reference's own 40-year fixture (kelvin), and reports how many mask bits the difference flips.
"""
import os

import numpy as np
import pytest

from marex_amd import binning, calendar, synth, zarr_io
from oracle import marex_oracle as orc

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")


def move_mean_bottleneck_f32(a: np.ndarray, window: int) -> np.ndarray:
    """bottleneck.move_mean(a, window, min_count=window) along axis 0 for float32 input WITHOUT NaNs inside the ocean
    columns (NaN columns stay NaN): float32 running sum, add the newest and subtract the oldest, times float32(1/window)."""
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[0]
    out = np.full(a.shape, np.nan, dtype=np.float32)
    asum = np.zeros(a.shape[1:], dtype=np.float32)
    for i in range(min(window, n)):
        asum = (asum + a[i]).astype(np.float32)
    if n >= window:
        out[window - 1] = asum / np.float32(window)  # WHILE1 branch divides by the count
        inv = np.float32(1.0) / np.float32(window)
        for i in range(window, n):
            asum = (asum + (a[i] - a[i - window]).astype(np.float32)).astype(np.float32)
            out[i] = asum * inv
    return out


def smooth_like_the_reference(x: np.ndarray, S: int, chunk: int = 25) -> np.ndarray:
    """``rolling(time=S, center=True).mean()`` the way xarray + dask run it: every time chunk of ``chunk`` steps is
    extended by the ghost steps it needs, bottleneck's trailing ``move_mean`` runs on that piece alone (its running sum
    starts afresh per chunk), the result is shifted to centre the window and trimmed to the chunk."""
    T = x.shape[0]
    lo, hi = S // 2, S - 1 - S // 2
    out = np.full(x.shape, np.nan, dtype=np.float32)
    for a in range(0, T, chunk):
        b = min(a + chunk, T)
        p0, p1 = max(a - lo, 0), min(b + hi, T)
        mm = move_mean_bottleneck_f32(x[p0:p1], S)  # mm[k] = mean of piece rows k-S+1 .. k
        for t in range(a, b):
            k = t + hi - p0  # trailing index of the window centred on t
            if t - lo >= 0 and t + hi < T:
                out[t] = mm[k]
    return out


def climatology_descending(s: np.ndarray, tindex: np.ndarray, W: int) -> np.ndarray:
    """C2 with the window summed from the newest to the oldest year (another order flox's chunked combine can take)."""
    T, C = s.shape
    clim = np.full((T, C), np.nan, dtype=np.float32)
    for Y in range(W, tindex.shape[0]):
        tgt = tindex[Y]
        acc = np.zeros((366, C), dtype=np.float32)
        cnt = np.zeros((366, C), dtype=np.int32)
        for yy in range(Y - 1, Y - W - 1, -1):
            ts = tindex[yy]
            present = ts >= 0
            v = s[np.where(present, ts, 0)]
            valid = present[:, None] & ~np.isnan(v)
            acc = np.where(valid, acc + v, acc)
            cnt += valid
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = np.where(cnt > 0, acc / cnt.astype(np.float32), np.float32(np.nan)).astype(np.float32)
        have = tgt >= 0
        clim[tgt[have]] = mean[have]
    return clim


def _compare(x, tm, ny, nx, W, S, wd, ws):
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    ref = orc.preprocess_arrays(x, cal, ny=ny, nx=nx, window_year_baseline=W, smooth_days_baseline=S,
                                window_days_hobday=wd, window_spatial_hobday=ws, edges=bt.edges, centres=bt.centres)
    s_alt = smooth_like_the_reference(x, S, chunk=25)
    s_con = orc.rolling_mean_centered(x, S)
    assert np.array_equal(np.isnan(s_alt), np.isnan(s_con))  # same NaN rim: first S//2 and last S-1-S//2 steps
    clim_alt = climatology_descending(s_alt, cal.tindex, W)
    with np.errstate(invalid="ignore"):
        anom_alt = (x - clim_alt)[cal.kept].astype(np.float32)
    thr_alt, _ = orc.hobday_thresholds_approx(anom_alt, cal.doy_out, 0.95, wd, ws, bt.edges, bt.centres, ny, nx)
    ext_alt = orc.mask_ge_doy(anom_alt, thr_alt, cal.doy_out)
    ocean = ref["mask"]
    scale = float(np.nanmax(np.abs(x)))
    d_anom = float(np.nanmax(np.abs(anom_alt[:, ocean] - ref["dat_anomaly"][:, ocean])))
    d_thr = float(np.nanmax(np.abs(thr_alt[ocean] - ref["thresholds"][ocean])))
    flips = int((ext_alt[:, ocean] != ref["extreme_events"][:, ocean]).sum())
    n = int(ref["extreme_events"][:, ocean].size)
    return d_anom, d_thr, flips, n, scale


def test_contract_vs_running_sum_on_the_synthetic_field(capsys):
    tm = calendar.daily_time_axis("1990-01-01", 22 * 365 + 5)
    x = synth.synth_field(synth.make_tables(tm, 6, 8))
    d_anom, d_thr, flips, n, scale = _compare(x, tm, 6, 8, W=15, S=21, wd=11, ws=5)
    with capsys.disabled():
        print(f"\n[contract bound, synthetic degC] max|d anom| = {d_anom:.3e} ({d_anom / scale:.2e} of max|x| = {scale:.1f}), "
              f"max|d thr| = {d_thr:.3e}, mask bits flipped: {flips} of {n} ({flips / n:.2e})")
    assert d_anom <= 1e-5 * scale            # north_star: fp32 anomalies within 1e-5 relative
    assert d_thr <= 0.01 + 1e-6              # thresholds move by less than one histogram bin (precision 0.01)
    assert flips / n <= 2e-4                 # a handful of samples sitting within the rounding noise of their threshold


def test_contract_vs_running_sum_on_the_reference_fixture(capsys):
    """The same bound on the reference's own SST fixture (kelvin: float32 spacing 3e-5, the harder case)."""
    p = os.path.join(FIX, "sst_gridded.zarr")
    x = zarr_io.read_array(os.path.join(p, "to"))[:, :6, :8].reshape(14611, 48)[: 25 * 365 + 7].copy()
    tm = zarr_io.decode_cf_time(zarr_io.read_array(os.path.join(p, "time")), zarr_io.array_attrs(os.path.join(p, "time")))[: x.shape[0]]
    d_anom, d_thr, flips, n, scale = _compare(x, tm, 6, 8, W=15, S=21, wd=11, ws=5)
    with capsys.disabled():
        print(f"[contract bound, reference fixture K] max|d anom| = {d_anom:.3e} ({d_anom / scale:.2e} of max|x| = {scale:.1f}), "
              f"max|d thr| = {d_thr:.3e}, mask bits flipped: {flips} of {n} ({flips / n:.2e})")
    assert d_anom <= 1e-5 * scale
    assert d_thr <= 0.01 + 1e-6
    assert flips / n <= 1e-3
