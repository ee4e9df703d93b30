"""CPU oracle for the ``preprocess_data`` hot path -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the reference algorithm
(wienkers/marEx, ``marEx/detect.py``).  It is the *checker* for the HIP
kernels: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped package
``marex_amd`` never imports, calls or falls back to anything in ``oracle/``.

Parity status
-------------
* ``rolling_histogram_quantile``  -- PINNED: bit-for-bit against the real
  reference leaf ``_rolling_histogram_quantile`` (detect.py:2465-2559) through
  ``tests/golden/hist_quantile_*.npz`` (made by ``tests/golden/make_goldens.py``,
  which imports the reference in the build container).
* ``preprocessing_steps`` strings -- PINNED the same way (detect.py:844-888);
  the restatement lives in the shipped package (host logic), the golden in tests.
* ``np.digitize`` / ``np.arange(float32)`` / ``np.nanpercentile`` / ``np.linalg.pinv``
  are NumPy itself (same 2.2.6 here and on the GPU box).
* Rolling mean, grouped ``nanmean`` and the spatial pooling restate the public
  semantics of third-party libraries that are NOT under /root/reference and not
  installed (xarray -> bottleneck ``move_mean``; ``flox.xarray.xarray_reduce``;
  ``xhistogram``; all unpinned in the reference's pyproject.toml:28-43).  For
  those stages the reference's own result depends on Dask chunking (fp32
  running sums per chunk), so parity is defined against the ARITHMETIC CONTRACT
  below and is otherwise UNPINNED beyond the reference tests' statistical pins
  (extreme frequency, means; see tests/test_oracle_stats.py).

Arithmetic contract (what "bit-exact" means for the GPU path)
------------------------------------------------------------
C1  smoothing (detect.py:1810-1812):  ``s[t] = fl32( SUM / fl32(S) )`` where SUM is
    the float32 sequential sum ``((x[t-l] + x[t-l+1]) + ...) + x[t+h]``,
    ``l = S//2``, ``h = S-1-l``; NaN when the window leaves ``[0,T)`` or holds a
    NaN (``min_periods = S``).  float32 like the reference (bottleneck works in
    the input dtype) but re-summed per output, so independent of any chunking.
C2  rolling climatology (detect.py:1622-1669): for target year ``Y >= min_year+W``
    and dayofyear ``d``: ``clim = fl32( ACC / fl32(n) )``, ``ACC = 0f`` then ``ACC += s``
    over the finite ``s`` of calendar years ``Y-W .. Y-1`` at dayofyear ``d`` in
    ascending year, ``n`` = number of terms, NaN when ``n == 0`` (nanmean, float32).
C3  anomaly (detect.py:1844): single float32 subtraction ``x - clim``.
C4  bins (detect.py:2603-2631): ``np.digitize(anom, edges) - 1`` with the float32
    edge table of ``marex_amd.binning.hobday_bins``; index ``nb`` (NaN, +inf,
    ``>= edges[-1]``) is dropped from every count.
C5  counts, pooling and the count-interpolated quantile are integer / float64
    exactly as in detect.py:2494-2559, 2652-2668, 2704-2732.
C6  mask (detect.py:2004): ``anom >= thr`` (False when either side is NaN).
C7  detrend (detect.py:2143-2224): float64 reductions over time as partial sums over blocks of
    ``DETREND_TBLOCK`` = 1024 consecutive timesteps (ascending t, from 0.0) combined in ascending
    block order (from 0.0); trend = float64 sum over coefficients in ascending k, rounded to float32
    once; time mean rounded to float32 once.  Against the reference (BLAS order unspecified) this
    stage is tolerance-only.
C8  std_normalise (detect.py:2257-2278): see ``std_normalise`` -- float64 two-pass population std
    per dayofyear, float32 squares, float64 30-term wrapped window mean, float32 sqrt and division.
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

N_DOY = 366


# --------------------------------------------------------------------------------------
# a3  validation                                                     detect.py:205-279
# --------------------------------------------------------------------------------------
def validate_data_values(x: np.ndarray) -> Dict[str, int]:
    """Counts behind ``_validate_data_values`` (detect.py:222-256).  ``x`` is ``[T, C]``."""
    ocean = np.isfinite(x[0])
    invalid_per_cell = (~np.isfinite(x)).sum(axis=0)
    invalid_in_ocean = np.where(ocean, invalid_per_cell, 0)
    return {
        "n_ocean": int(ocean.sum()),
        "max_invalid": int(invalid_in_ocean.max()) if x.shape[1] else 0,
        "total_invalid_in_ocean": int(invalid_in_ocean.sum()),
        "locations_affected": int((invalid_in_ocean > 0).sum()),
    }


# --------------------------------------------------------------------------------------
# a5  smoothing                                                      detect.py:1810-1812
# --------------------------------------------------------------------------------------
def rolling_mean_centered(x: np.ndarray, S: int) -> np.ndarray:
    """Contract C1: centred ``S``-step mean over axis 0, float32, ``min_periods=S``."""
    x = np.asarray(x, dtype=np.float32)
    T = x.shape[0]
    out = np.full(x.shape, np.nan, dtype=np.float32)
    if S > T:
        return out
    lo = S // 2
    n = T - S + 1
    acc = x[0:n].copy()
    for k in range(1, S):
        acc += x[k : k + n]  # float32 add, ascending k
    with np.errstate(invalid="ignore"):
        out[lo : lo + n] = acc / np.float32(S)
    return out


# --------------------------------------------------------------------------------------
# a6  rolling climatology                                            detect.py:1511-1688
# --------------------------------------------------------------------------------------
def rolling_climatology(s: np.ndarray, tindex: np.ndarray, W: int) -> np.ndarray:
    """Contract C2.  ``s`` is ``[T, C]`` (already smoothed or raw); returns ``[T, C]`` float32.

    ``tindex[y, d-1]`` is the timestep of calendar year index ``y`` and dayofyear ``d`` (or -1).
    Rows of years ``< W`` (no full history) are NaN, as detect.py:1634 + fill_value=NaN.
    """
    s = np.asarray(s, dtype=np.float32)
    T, C = s.shape
    n_cal = tindex.shape[0]
    clim = np.full((T, C), np.nan, dtype=np.float32)
    for Y in range(W, n_cal):
        tgt = tindex[Y]
        if not (tgt >= 0).any():
            continue
        acc = np.zeros((N_DOY, C), dtype=np.float32)
        cnt = np.zeros((N_DOY, C), dtype=np.int32)
        for yy in range(Y - W, Y):
            ts = tindex[yy]
            present = ts >= 0
            v = s[np.where(present, ts, 0)]
            valid = present[:, None] & ~np.isnan(v)
            with np.errstate(invalid="ignore", over="ignore"):
                acc = np.where(valid, acc + v, acc)
            cnt += valid
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = np.where(cnt > 0, acc / cnt.astype(np.float32), np.float32(np.nan)).astype(np.float32)
        have = tgt >= 0
        clim[tgt[have]] = mean[have]
    return clim


# --------------------------------------------------------------------------------------
# a7  shifting-baseline anomaly                                      detect.py:1819-1850, 615-641
# --------------------------------------------------------------------------------------
def shifting_baseline_anomaly(x: np.ndarray, cal, W: int, S: int) -> Tuple[np.ndarray, np.ndarray]:
    """Returns ``(anom [T_out, C] float32, mask [C] bool)`` for the rows kept by the trim."""
    x = np.asarray(x, dtype=np.float32)
    s = rolling_mean_centered(x, S)
    clim = rolling_climatology(s, cal.tindex, W)
    with np.errstate(invalid="ignore"):
        anom = (x - clim)[cal.kept]
    mask = np.isfinite(x[0])
    return anom.astype(np.float32), mask


# --------------------------------------------------------------------------------------
# a13 fixed baseline                                                 detect.py:2299-2397
# --------------------------------------------------------------------------------------
def fixed_baseline_anomaly(
    x: np.ndarray, cal, reference_period: Optional[Tuple[int, int]] = None
) -> Tuple[np.ndarray, np.ndarray]:
    """``clim[d] = nanmean_{t: doy=d (, year in ref)} x`` in float32 (ascending t), ``anom = x - clim[doy]``."""
    x = np.asarray(x, dtype=np.float32)
    T, C = x.shape
    use = np.ones(T, dtype=bool)
    if reference_period is not None:
        use = (cal.year >= reference_period[0]) & (cal.year <= reference_period[1])
    acc = np.zeros((N_DOY, C), dtype=np.float32)
    cnt = np.zeros((N_DOY, C), dtype=np.int32)
    for t in range(T):
        if not use[t]:
            continue
        d = int(cal.doy[t]) - 1
        valid = ~np.isnan(x[t])
        with np.errstate(invalid="ignore", over="ignore"):
            acc[d] = np.where(valid, acc[d] + x[t], acc[d])
        cnt[d] += valid
    with np.errstate(invalid="ignore", divide="ignore"):
        clim = np.where(cnt > 0, acc / cnt.astype(np.float32), np.float32(np.nan)).astype(np.float32)
        anom = (x - clim[cal.doy.astype(np.int64) - 1]).astype(np.float32)
    return anom, np.isfinite(x[0])


# --------------------------------------------------------------------------------------
# a12 polynomial / harmonic detrend                                  detect.py:2061-2296
# --------------------------------------------------------------------------------------
DETREND_TBLOCK = 1024  # timesteps per partial sum of the detrend reductions (part of the arithmetic contract)


def detrend_anomaly(x: np.ndarray, model: np.ndarray, pmodel: np.ndarray, force_zero_mean: bool) -> np.ndarray:
    """``coef = pmodel^T x`` (float64), ``resid = x - fl32(model^T coef)``, optional ``- mean_t``.

    The reference's BLAS summation order is unspecified, so against the reference this stage is tolerance-only
    (SURVEY A.9).  Contract of the two reductions over time (what the kernels reproduce bit for bit): partial sums
    over blocks of ``DETREND_TBLOCK`` consecutive timesteps -- float64, ascending t, starting from 0.0 -- combined
    in ascending block order starting from 0.0.  The trend is the float64 sum over coefficients in ascending k,
    rounded to float32 once; the time mean is rounded to float32 once.
    """
    x = np.asarray(x, dtype=np.float32)
    T, C = x.shape
    n_coef = model.shape[0]
    coef = np.zeros((n_coef, C), dtype=np.float64)
    for t0 in range(0, T, DETREND_TBLOCK):
        part = np.zeros((n_coef, C), dtype=np.float64)
        for t in range(t0, min(t0 + DETREND_TBLOCK, T)):
            part += pmodel[t][:, None] * x[t].astype(np.float64)[None, :]
        coef += part
    resid = np.empty_like(x)
    for t in range(T):
        trend = np.zeros(C, dtype=np.float64)
        for k in range(n_coef):
            trend += model[k, t] * coef[k]
        resid[t] = x[t] - trend.astype(np.float32)
    if force_zero_mean:
        mean = np.zeros(C, dtype=np.float64)
        for t0 in range(0, T, DETREND_TBLOCK):
            part = np.zeros(C, dtype=np.float64)
            for t in range(t0, min(t0 + DETREND_TBLOCK, T)):
                part += resid[t].astype(np.float64)
            mean += part
        mean = (mean / np.float64(T)).astype(np.float32)
        resid = (resid - mean[None, :]).astype(np.float32)
    return resid


# --------------------------------------------------------------------------------------
# a10 binning + counts + pooling                                     detect.py:2601-2668
# --------------------------------------------------------------------------------------
def digitize_bins(anom: np.ndarray, edges: np.ndarray) -> np.ndarray:
    """Contract C4 (detect.py:2622-2631).  Index ``nb = len(edges)-1`` means "dropped"."""
    return (np.digitize(anom, edges) - 1).astype(np.uint16)


def doy_bin_counts(bins: np.ndarray, doy_out: np.ndarray, nb: int) -> np.ndarray:
    """``hist[c, d-1, b]`` = number of kept timesteps with that dayofyear and bin (detect.py:2638-2648)."""
    T, C = bins.shape
    valid = bins < nb
    d0 = (doy_out.astype(np.int64) - 1)[:, None]
    flat = (np.arange(C, dtype=np.int64)[None, :] * N_DOY + d0) * nb + bins.astype(np.int64)
    hist = np.bincount(flat[valid], minlength=C * N_DOY * nb)
    return hist.reshape(C, N_DOY, nb)


def spatial_pool(hist: np.ndarray, ny: int, nx: int, ws: int) -> np.ndarray:
    """Lon-periodic, lat-truncated ``ws x ws`` box sum of the histograms (detect.py:2652-2668)."""
    if ws is None or ws <= 1:
        return hist
    p = ws // 2
    h = hist.reshape(ny, nx, *hist.shape[1:])
    lon = np.zeros_like(h)
    for o in range(-p, p + 1):
        lon += np.roll(h, -o, axis=1)  # h[j, (i+o) mod nx]
    out = np.zeros_like(h)
    for j in range(ny):
        out[j] = lon[max(0, j - p) : min(ny, j + p + 1)].sum(axis=0)
    return out.reshape(hist.shape)


# --------------------------------------------------------------------------------------
# a11 count-interpolated quantile                                    detect.py:2465-2559
# --------------------------------------------------------------------------------------
def rolling_histogram_quantile(hist: np.ndarray, wd: int, q: float, centres: np.ndarray) -> np.ndarray:
    """Restatement of ``_rolling_histogram_quantile`` vectorised over leading axes.

    ``hist`` is ``[..., 366, nb]`` integer counts, returns ``[..., 366]`` float32.
    Pinned bit-for-bit against the reference leaf (tests/test_oracle_golden.py).
    """
    if wd < 3 or wd % 2 == 0:
        # wd=1 crashes the reference (pad_size=0 makes hist_chunk[-0:] the whole array)
        raise ValueError("window_days_hobday must be an odd number >= 3")
    hist = np.asarray(hist)
    n_doy, nb = hist.shape[-2:]
    pad = wd // 2
    hp = np.concatenate([hist[..., -pad:, :], hist, hist[..., :pad, :]], axis=-2).astype(np.int64)
    hw = np.zeros(hist.shape, dtype=np.int64)
    for o in range(wd):  # detect.py:2499-2500
        hw += hp[..., o : o + n_doy, :]
    cs = np.cumsum(hw, axis=-1).astype(np.int32)  # detect.py:2510
    tot = cs[..., -1]
    qpos = q * tot  # float64, detect.py:2516
    iu = (cs <= qpos[..., None]).sum(axis=-1).astype(np.int32)  # searchsorted(side="right"), 2527
    iu = np.where(tot <= 0, 0, iu)
    iu = np.clip(iu, 0, nb - 1)
    il = np.maximum(0, iu - 1)
    cl = np.take_along_axis(cs, il[..., None].astype(np.int64), axis=-1)[..., 0]
    cu = np.take_along_axis(cs, iu[..., None].astype(np.int64), axis=-1)[..., 0]
    bl = centres[il]
    bu = centres[iu]
    eps = 1e-10
    diff = cu - cl
    safe = np.where(diff > eps, diff, 1.0)
    frac = np.where(diff > eps, (qpos - cl) / safe, 0.5)
    thr = bl + frac * (bu - bl)  # (bu-bl) in the centres' dtype, product and sum in float64 (2550)
    thr = np.where(tot > 0, thr, np.nan)
    thr = np.where((iu == 0) & (tot > 0), centres[0], thr)
    return thr.astype(np.float32)


def hobday_thresholds_approx(
    anom: np.ndarray,
    doy_out: np.ndarray,
    q: float,
    wd: int,
    ws: Optional[int],
    edges: np.ndarray,
    centres: np.ndarray,
    ny: int = 0,
    nx: int = 0,
    row_block: int = 8,
) -> Tuple[np.ndarray, Dict[str, float]]:
    """``_compute_histogram_quantile_2d`` (detect.py:2562-2734) -> ``thr [C, 366]`` float32 + warning stats.

    ``ny == 0`` means unstructured (no spatial pooling allowed).  Processed in latitude blocks
    with a ``ws//2`` halo so the dense ``[cells, 366, nb]`` histogram stays small.
    """
    anom = np.asarray(anom, dtype=np.float32)
    T, C = anom.shape
    nb = len(edges) - 1
    bins = digitize_bins(anom, edges)
    thr = np.empty((C, N_DOY), dtype=np.float32)
    p = (ws // 2) if (ws is not None and ws > 1 and ny > 0) else 0
    if ny > 0:
        assert ny * nx == C
        for j0 in range(0, ny, row_block):
            j1 = min(ny, j0 + row_block)
            a0, a1 = max(0, j0 - p), min(ny, j1 + p)
            hb = doy_bin_counts(bins[:, a0 * nx : a1 * nx], doy_out, nb)
            hb = spatial_pool(hb, a1 - a0, nx, ws if p else 1)
            # rows of the block inside the halo'd slab see the full (lat-truncated) window
            sl = hb.reshape(a1 - a0, nx, N_DOY, nb)[j0 - a0 : j1 - a0].reshape(-1, N_DOY, nb)
            thr[j0 * nx : j1 * nx] = rolling_histogram_quantile(sl, wd, q, centres)
    else:
        step = 4096
        for c0 in range(0, C, step):
            hb = doy_bin_counts(bins[:, c0 : c0 + step], doy_out, nb)
            thr[c0 : c0 + step] = rolling_histogram_quantile(hb, wd, q, centres)
    # NaN where the first kept anomaly is NaN (detect.py:2704-2705)
    thr[np.isnan(anom[0])] = np.nan
    upper, lower = edges[-2], edges[3]
    with np.errstate(invalid="ignore"):
        too_high = thr > upper
        too_low = thr < lower
    stats = {
        "n_too_high": int(too_high.sum()),
        "n_too_low": int(too_low.sum()),
        "max": float(np.nanmax(thr)) if np.isfinite(thr).any() else float("nan"),
        "min": float(np.nanmin(thr)) if np.isfinite(thr).any() else float("nan"),
    }
    thr = np.where(too_low, lower, thr).astype(np.float32)  # detect.py:2732
    return thr, stats


# --------------------------------------------------------------------------------------
# a9  exact Hobday percentile                                        detect.py:1921-1956
# --------------------------------------------------------------------------------------
def doy_window_masks(doy_out: np.ndarray, wd: int) -> np.ndarray:
    """``masks[d-1, t]`` -- timesteps whose dayofyear lies in the wrapped ``wd`` window of ``d`` (1929-1934)."""
    half = wd // 2
    masks = np.zeros((N_DOY, doy_out.size), dtype=bool)
    for d in range(1, N_DOY + 1):
        for o in range(-half, half + 1):
            masks[d - 1] |= doy_out == ((d - 1 + o) % N_DOY) + 1
    return masks


def hobday_thresholds_exact(anom: np.ndarray, doy_out: np.ndarray, percentile: float, wd: int) -> np.ndarray:
    """``np.nanpercentile`` per (dayofyear window, cell) exactly as the reference calls it -> ``[366, C]`` float32."""
    import warnings

    anom = np.asarray(anom, dtype=np.float32)
    masks = doy_window_masks(doy_out, wd)
    data = np.ascontiguousarray(anom.T)  # (*spatial, time)
    out = np.full((data.shape[0], N_DOY), np.nan, dtype=np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)  # all-NaN slices (land)
        for i in range(N_DOY):
            if masks[i].any():
                out[:, i] = np.nanpercentile(data[:, masks[i]], percentile, axis=-1)
    return np.ascontiguousarray(out.T)


def percentile_lerp_f32(v_sorted: np.ndarray, percentile: float) -> np.float32:
    """Float32 mirror of NumPy 2.x "linear" percentile on sorted finite float32 data.

    Every product / sum is rounded to float32 separately; this is the arithmetic the exact GPU
    kernel follows.  Checked against ``np.nanpercentile`` in tests/test_oracle_exact.py.
    """
    m = v_sorted.size
    if m == 0:
        return np.float32(np.nan)
    q = np.float32(percentile) / np.float32(100)
    h = np.float32(m - 1) * q
    lo = int(np.floor(h))
    if lo >= m - 1:
        return np.float32(v_sorted[m - 1])
    g = np.float32(h - np.float32(lo))
    a = np.float32(v_sorted[lo])
    b = np.float32(v_sorted[lo + 1])
    d = np.float32(b - a)
    if g >= np.float32(0.5):
        return np.float32(b - np.float32(d * np.float32(np.float32(1) - g)))
    return np.float32(a + np.float32(d * g))


# --------------------------------------------------------------------------------------
# a14 global (constant-in-time) threshold                            detect.py:2737-2923
# --------------------------------------------------------------------------------------
def global_threshold_exact(anom: np.ndarray, percentile: float) -> np.ndarray:
    """``DataArray.quantile(p/100, dim=time)`` -> float64 ``[C]`` (detect.py:2899).

    xarray (absent here; restated from its documented behaviour) calls ``np.nanquantile`` with
    ``q = np.atleast_1d(np.asarray(q, dtype=np.float64))`` for float data: NaNs are skipped, the
    interpolation runs in float64 on the float32 order statistics, all-NaN cells give NaN.
    Contract: NumPy's ``_quantile`` arithmetic applied PER CELL.  (``np.nanquantile`` itself runs
    ``apply_along_axis``, whose output dtype is taken from the first slice -- float32 when that cell is
    all-NaN -- so the reference's own result is float32- or float64-rounded depending on which cell
    starts a Dask chunk; that accident is not reproduced.)
    """
    anom = np.asarray(anom, dtype=np.float32)
    q = np.atleast_1d(np.asarray(percentile / 100.0, dtype=np.float64))
    out = np.full(anom.shape[1], np.nan, dtype=np.float64)
    for c in range(anom.shape[1]):
        v = anom[:, c]
        v = v[~np.isnan(v)]
        if v.size:
            out[c] = np.quantile(v, q)[0]
    return out


def global_threshold_approx(anom: np.ndarray, q: float, edges: np.ndarray, centres: np.ndarray):
    """``_compute_histogram_quantile_1d`` (detect.py:2737-2865) -> float64 ``[C]`` + warning stats.

    ``xhistogram`` is absent; its documented semantics are those of ``np.histogram`` (half-open bins,
    last bin closed, NaN ignored), which is what is restated here.
    """
    anom = np.asarray(anom, dtype=np.float32)
    T, C = anom.shape
    nb = len(edges) - 1
    idx = np.digitize(anom, edges) - 1
    idx = np.where(anom == edges[-1], nb - 1, idx)  # right edge belongs to the last bin
    valid = (idx >= 0) & (idx < nb)
    flat = np.arange(C, dtype=np.int64)[None, :] * nb + np.where(valid, idx, 0)
    hist = np.bincount(flat[valid], minlength=C * nb).reshape(C, nb).astype(np.float64)
    hsum = hist.sum(axis=1) + 1e-10
    cdf = np.cumsum(hist / hsum[:, None], axis=1)
    eps = 1e-10
    iu = np.argmax(cdf >= (q - eps), axis=1)
    ib = np.where(iu - 1 > 0, iu - 1, 0)
    cdf_t = np.take_along_axis(cdf, ib[:, None], axis=1)
    il = np.argmax(cdf > cdf_t, axis=1)
    il = np.where(il < 0, 0, np.where(il > nb - 2, nb - 2, il))
    iu = np.where(iu < 1, 1, np.where(iu > nb - 1, nb - 1, iu))
    cl = np.take_along_axis(cdf, il[:, None], axis=1)[:, 0]
    cu = np.take_along_axis(cdf, iu[:, None], axis=1)[:, 0]
    bl, bu = centres[il], centres[iu]
    denom = cu - cl
    exact = np.abs(cl - q) < eps
    zero = np.abs(denom) <= eps
    frac = (q - cl) / np.where(np.abs(denom) > eps, denom, 1.0)
    thr = bl + frac * (bu - bl)
    thr = np.where(exact, bl, thr)
    thr = np.where(zero & ~exact, (bl + bu) / 2, thr)
    thr = np.where(np.isnan(anom).any(axis=0), np.nan, thr)
    upper, lower = edges[-2], edges[3]
    with np.errstate(invalid="ignore"):
        too_high = (thr > upper) & ~np.isnan(thr)
        too_low = (thr < lower) & ~np.isnan(thr)
    stats = {
        "n_too_high": int(too_high.sum()),
        "n_too_low": int(too_low.sum()),
        "max": float(np.nanmax(thr)) if np.isfinite(thr).any() else float("nan"),
        "min": float(np.nanmin(thr)) if np.isfinite(thr).any() else float("nan"),
    }
    thr = np.where(too_low, lower, thr)
    return thr, stats


# --------------------------------------------------------------------------------------
# a9  mask                                                           detect.py:2003-2004, 2915
# --------------------------------------------------------------------------------------
def mask_ge_doy(anom: np.ndarray, thr_c_doy: np.ndarray, doy_out: np.ndarray) -> np.ndarray:
    """``extreme[t, c] = anom[t, c] >= thr[c, doy(t)]`` (contract C6)."""
    with np.errstate(invalid="ignore"):
        return anom >= thr_c_doy[:, doy_out.astype(np.int64) - 1].T


def mask_ge_const(anom: np.ndarray, thr_c: np.ndarray) -> np.ndarray:
    with np.errstate(invalid="ignore"):
        return anom >= thr_c[None, :]


# --------------------------------------------------------------------------------------
# a1  whole path on plain arrays
# --------------------------------------------------------------------------------------
def preprocess_arrays(
    x: np.ndarray,
    cal,
    *,
    ny: int = 0,
    nx: int = 0,
    method_anomaly: str = "shifting_baseline",
    method_extreme: str = "hobday_extreme",
    threshold_percentile: float = 95,
    window_year_baseline: int = 15,
    smooth_days_baseline: int = 21,
    window_days_hobday: int = 11,
    window_spatial_hobday: Optional[int] = None,
    method_percentile: str = "approximate",
    edges: Optional[np.ndarray] = None,
    centres: Optional[np.ndarray] = None,
    model: Optional[np.ndarray] = None,
    pmodel: Optional[np.ndarray] = None,
    force_zero_mean: bool = True,
    reference_period: Optional[Tuple[int, int]] = None,
) -> Dict[str, object]:
    """The numeric core of ``preprocess_data`` (detect.py:287-841) on ``x [T, C]`` float32.

    ``cal`` is a ``marex_amd.calendar.CalendarPlan`` built with the matching trim; tables
    (``edges``/``centres``, ``model``/``pmodel``) come from ``marex_amd.binning`` / ``calendar``.
    ``window_spatial_hobday=None`` on a gridded input means 5 (detect.py:1451-1452).
    """
    x = np.asarray(x, dtype=np.float32)
    if method_anomaly == "shifting_baseline":
        anom, mask = shifting_baseline_anomaly(x, cal, window_year_baseline, smooth_days_baseline)
    elif method_anomaly == "fixed_baseline":
        anom, mask = fixed_baseline_anomaly(x, cal, reference_period)
    elif method_anomaly in ("detrend_harmonic", "detrend_fixed_baseline"):
        anom = detrend_anomaly(x, model, pmodel, force_zero_mean)
        mask = np.isfinite(x[0])
        if method_anomaly == "detrend_fixed_baseline":
            anom, _ = fixed_baseline_anomaly(anom, cal, reference_period)
    else:
        raise ValueError(method_anomaly)

    stats: Dict[str, float] = {}
    if method_extreme == "hobday_extreme":
        ws = window_spatial_hobday
        if ws is None and ny > 0:
            ws = 5
        if method_percentile == "exact":
            thr = hobday_thresholds_exact(anom, cal.doy_out, threshold_percentile, window_days_hobday)
            extremes = mask_ge_doy(anom, np.ascontiguousarray(thr.T), cal.doy_out)
        else:
            thr, stats = hobday_thresholds_approx(
                anom, cal.doy_out, threshold_percentile / 100.0, window_days_hobday, ws, edges, centres, ny, nx
            )
            extremes = mask_ge_doy(anom, thr, cal.doy_out)
    elif method_extreme == "global_extreme":
        if method_percentile == "exact":
            thr = global_threshold_exact(anom, threshold_percentile)
        else:
            thr, stats = global_threshold_approx(anom, threshold_percentile / 100.0, edges, centres)
        extremes = mask_ge_const(anom, thr)
    else:
        raise ValueError(method_extreme)
    return {"dat_anomaly": anom, "mask": mask, "thresholds": thr, "extreme_events": extremes, "stats": stats}


def std_normalise(anom: np.ndarray, doy: np.ndarray, window: int = 30):
    """``std_normalise`` branch of detrend_harmonic (marEx/detect.py:2257-2278).

    * ``std_day[d]`` = population standard deviation (flox ``func="std"``, ddof 0, NaN-propagating, NaN for a
      dayofyear that never occurs) of the anomalies with dayofyear ``d+1``.  Contract: float64 two-pass -- mean from
      a sequential sum in ascending time, then the sequential sum of squared deviations, one float64 sqrt, rounded
      to float32 (the reference's flox/float32 reduction differs in the last bits and with its chunking).
    * ``STD[d]`` = sqrt of the centred ``window``-day mean of ``float32(std_day**2)`` on the wrapped dayofyear axis:
      ``pad(16, wrap) -> rolling(30, center=True).mean() -> isel(16:382)`` = offsets ``-window//2 .. window-1-window//2``;
      float64 sequential sum, ``/ window``, rounded to float32, float32 sqrt.
    * ``dat_stn = anom / where(STD > 1e-10, STD, NaN)[dayofyear]`` (float32 division).

    Returns ``(dat_stn [T, C] float32, STD [366, C] float32)``.
    """
    anom = np.asarray(anom, dtype=np.float32)
    doy = np.asarray(doy)
    C = anom.shape[1]
    std_day = np.full((366, C), np.nan, dtype=np.float32)
    for d in range(366):
        rows = np.flatnonzero(doy == d + 1)
        if rows.size == 0:
            continue
        x = anom[rows].astype(np.float64)
        n = float(rows.size)
        with np.errstate(invalid="ignore"):
            mean = np.cumsum(x, axis=0)[-1] / n
            dv = x - mean
            ss = np.cumsum(dv * dv, axis=0)[-1]
            std_day[d] = np.sqrt(ss / n).astype(np.float32)
    with np.errstate(invalid="ignore"):
        sq = std_day * std_day  # float32
        lo = window // 2
        acc = np.zeros((366, C), dtype=np.float64)
        idx = np.arange(366)
        for k in range(window):
            acc = acc + sq[(idx - lo + k) % 366].astype(np.float64)
        std_roll = np.sqrt((acc / float(window)).astype(np.float32))
        safe = np.where(std_roll > np.float32(1e-10), std_roll, np.float32(np.nan)).astype(np.float32)
        dat_stn = (anom / safe[np.asarray(doy, dtype=np.int64) - 1]).astype(np.float32)
    return dat_stn, std_roll


# --------------------------------------------------------------------------------------
# tracker pre-processing (first half)              track.py:1520-1726
# --------------------------------------------------------------------------------------
def fill_holes(data_bin: np.ndarray, mask: np.ndarray, R_fill: int, regional_mode: bool = False) -> np.ndarray:
    """``tracker.fill_holes`` on gridded data (marEx/track.py:1608-1676), literally: pad ``(y, x)`` by ``2 R`` with
    ``np.pad`` (``wrap`` / ``edge``), ``binary_closing`` then ``binary_opening`` with the disk ``x^2 + y^2 < R^2 + 1``
    applied per timestep, trim, ``where(mask, False)``.  The reference calls ``dask_image.ndmorph``, a chunked wrapper
    of the very ``scipy.ndimage`` functions used here (border_value 0), so this oracle is the reference's own library.
    ``data_bin``: bool ``[T, ny, nx]``; ``mask``: bool ``[ny, nx]``."""
    from scipy import ndimage as ndi

    data_bin = np.asarray(data_bin).astype(bool)
    mask = np.asarray(mask).astype(bool)
    R = int(R_fill)
    if R > 0:
        y, x = np.ogrid[-R:R + 1, -R:R + 1]
        se = (x ** 2 + y ** 2) < (R ** 2) + 1
        d = 2 * R
        p = np.pad(data_bin, ((0, 0), (d, d), (d, d)), mode="edge" if regional_mode else "wrap")
        p = ndi.binary_closing(p, structure=se[np.newaxis, :, :])
        p = ndi.binary_opening(p, structure=se[np.newaxis, :, :])
        data_bin = p[:, d:-d, d:-d]
    return data_bin & mask[np.newaxis, :, :]


def fill_time_gaps(data_bin: np.ndarray, mask: np.ndarray, R_fill: int, T_fill: int, regional_mode: bool = False) -> np.ndarray:
    """``tracker.fill_time_gaps`` (track.py:1678-1726): pad time by ``T_fill + 1`` False steps, ``binary_closing`` with
    ``T_fill + 1`` ones along time, trim, then ``fill_holes(R_fill // 2)``."""
    from scipy import ndimage as ndi

    data_bin = np.asarray(data_bin).astype(bool)
    if T_fill == 0:
        return data_bin
    k = int(T_fill) + 1
    p = np.pad(data_bin, ((k, k), (0, 0), (0, 0)), mode="constant", constant_values=False)
    p = ndi.binary_closing(p, structure=np.ones(k, dtype=bool)[:, np.newaxis, np.newaxis])
    return fill_holes(p[k:-k], mask, int(R_fill) // 2, regional_mode)


def label_objects_2d(data_bin: np.ndarray, wrap_x: bool = True) -> np.ndarray:
    """Connected components of every timestep on its own (track.py:2013-2031, ``time_connectivity=False``): 8-connected
    in ``(y, x)``, periodic in ``x`` unless regional; ``scipy.ndimage.label`` per timestep plus a union over the seam.
    Returns int64 labels, 0 = background, unique across time (numbering: scan order per timestep, offset by the number
    of objects before -- the reference's dask_image numbering may differ; only memberships are contractual)."""
    from scipy import ndimage as ndi

    data_bin = np.asarray(data_bin).astype(bool)
    T, ny, nx = data_bin.shape
    out = np.zeros((T, ny, nx), dtype=np.int64)
    offset = 0
    for t in range(T):
        lab, n = ndi.label(data_bin[t], structure=np.ones((3, 3), dtype=bool))
        if wrap_x and n > 0 and nx > 1:
            parent = np.arange(n + 1)

            def find(a):
                while parent[a] != a:
                    parent[a] = parent[parent[a]]
                    a = parent[a]
                return a

            left, right = lab[:, 0], lab[:, nx - 1]
            for y in range(ny):
                if left[y] == 0:
                    continue
                for yy in (y - 1, y, y + 1):
                    if 0 <= yy < ny and right[yy] != 0:
                        a, b = find(left[y]), find(right[yy])
                        if a != b:
                            parent[max(a, b)] = min(a, b)
            roots = np.array([find(i) for i in range(n + 1)])
            uniq, inv = np.unique(roots[1:], return_inverse=True)
            remap = np.concatenate([[0], inv + 1])
            lab = remap[lab]
            n = uniq.size
        out[t] = np.where(lab > 0, lab + offset, 0)
        offset += n
    return out


def filter_small_objects(data_bin: np.ndarray, area_filter_quartile: float = 0.5, area_filter_absolute=None,
                         regional_mode: bool = False):
    """``tracker.filter_small_objects`` on gridded data (track.py:1873-1911): areas in cells per 2-D object, threshold =
    ``np.percentile(areas, 100 q)`` (or the absolute one), keep ``area >= threshold`` -- and, as the reference's
    ``object_ids_keep[0] = -1`` does, never keep the first object of the list (here: the object that contains the
    first True cell in C order; the reference's list order comes from dask_image's labelling and is not pinned).
    Returns ``(filtered bool, threshold, areas, n_before, n_after)``."""
    data_bin = np.asarray(data_bin).astype(bool)
    lab = label_objects_2d(data_bin, wrap_x=not regional_mode)
    n = int(lab.max())
    areas = np.bincount(lab.reshape(-1), minlength=n + 1)[1:].astype(np.float64)
    if n == 0:
        raise ValueError("No objects found for area-based filtering")
    thr = float(area_filter_absolute) if area_filter_absolute is not None else float(np.percentile(areas, area_filter_quartile * 100.0))
    keep = areas >= thr
    keep[0] = False  # the first object of the list (label 1 = first True cell in scan order)
    out = np.concatenate([[False], keep])[lab]
    return out, thr, areas, n, int(keep.sum())


def _mesh_dilation_matrix(neighbours_int: np.ndarray):
    """``tracker._build_sparse_dilation_matrix`` (track.py:1093-1117): (cell, listed neighbour) entries plus identity."""
    from scipy.sparse import coo_matrix, csr_matrix, eye

    nb = np.asarray(neighbours_int)
    n = nb.shape[1]
    rows = np.repeat(np.arange(n), 3)
    cols = nb.T.flatten()
    ok = cols >= 0
    m = csr_matrix(coo_matrix((np.ones(int(ok.sum()), dtype=bool), (rows[ok], cols[ok])), shape=(n, n)))
    return m + eye(n, dtype=bool, format="csr")


def fill_holes_mesh(data_bin: np.ndarray, mask: np.ndarray, neighbours_int: np.ndarray, R_fill: int) -> np.ndarray:
    """``tracker.fill_holes`` on an unstructured mesh (track.py:1543-1606; ``sparse_bool_power`` = ``R`` boolean products
    with the dilation matrix).  ``data_bin`` bool ``[T, C]``, ``neighbours_int`` int ``[3, C]`` 0-based, -1 = none.  The
    reference does not re-apply the land mask at the end of this branch."""
    m = _mesh_dilation_matrix(neighbours_int).astype(np.int32)
    mask = np.asarray(mask).astype(bool)

    def power(b):
        v = b.T.astype(np.int32)  # [C, T]
        for _ in range(int(R_fill)):
            v = (m @ v > 0).astype(np.int32)
        return v.T.astype(bool)

    b = power(np.asarray(data_bin).astype(bool))
    b[:, ~mask] = True
    b = ~power(~b)
    b[:, ~mask] = True
    b = ~power(~b)
    return power(b)


def label_objects_mesh(data_bin: np.ndarray, mask: np.ndarray, neighbours_int: np.ndarray) -> np.ndarray:
    """Connected components per timestep over the mesh edges (track.py:1947-1985), land excluded.  int64 labels, 0 =
    background, made unique across time here (the reference restarts at 1 in every slice)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    d = np.asarray(data_bin).astype(bool) & np.asarray(mask).astype(bool)[None, :]
    nb = np.asarray(neighbours_int)
    T, C = d.shape
    out = np.zeros((T, C), dtype=np.int64)
    offset = 0
    for t in range(T):
        k, c = np.nonzero((nb != -1) & d[t][np.where(nb >= 0, nb, 0)] & d[t][None, :])
        g = coo_matrix((np.ones(k.size, dtype=np.int8), (nb[k, c], c)), shape=(C, C))
        n, lab = connected_components(g, directed=False)
        ids = np.unique(lab[d[t]])
        remap = np.zeros(n, dtype=np.int64)
        remap[ids] = np.arange(1, ids.size + 1)
        out[t] = np.where(d[t], remap[lab] + offset, 0)
        offset += ids.size
    return out


def filter_small_objects_mesh(data_bin, mask, neighbours_int, area_filter_quartile: float = 0.5, area_filter_absolute=None):
    """``tracker.filter_small_objects`` on an unstructured mesh (track.py:1776-1857): sizes in cells per (timestep, cluster);
    the percentile is taken over the clusters larger than 50 cells (5 with an absolute threshold); clusters are kept when
    STRICTLY larger than the threshold.  Returns ``(filtered, threshold, sizes of the clusters that entered the
    percentile, their number, number kept among them)``."""
    lab = label_objects_mesh(data_bin, mask, neighbours_int)
    n = int(lab.max())
    sizes = np.bincount(lab.reshape(-1), minlength=n + 1)[1:]
    big = sizes[sizes > (5 if area_filter_absolute is not None else 50)].astype(np.float64)
    if big.size == 0:
        raise ValueError("No objects found for area-based filtering")
    thr = float(area_filter_absolute) if area_filter_absolute is not None else float(np.percentile(big, area_filter_quartile * 100))
    keep = np.concatenate([[False], sizes > thr])
    return keep[lab], thr, big, int(big.size), int((big > thr).sum())
