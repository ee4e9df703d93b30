"""GPU: the tail kernels (marex_tail_extract_f32, marex_hobday_thresholds_tails_f32, marex_mask_ge_doy_tails_f32).

Tails are an internal representation (include/marex_hip.h, TAILS): what is checked against the oracle is what the
reference defines -- thresholds and the extreme mask -- plus the tails themselves against a NumPy statement of their
definition.  The cases force the parts of the kernels that ordinary fields rarely reach: tails much shorter than the
number of samples above the band (buckets re-read from the anomalies), thresholds that drift across many bands over
the year and jump between neighbouring cells (band rebuilds, extra passes), values beyond the edge table.
"""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def tails_reference(anom, cal, edges):
    """NumPy statement of the tails of ``anom`` [T_out, C]: per bucket the sorted (descending) keys of ALL its countable
    samples, keys[366, max_bucket, C] (0 = none), and aux[366, C] (uint32: count, "beyond the table" flag, the positions
    of the first two such samples of the bucket, "second exists", "more than two" -- csrc/marex_tails.hip.h)."""
    nb = edges.size - 1
    T_out, C = anom.shape
    nmax = int(np.diff(cal.doy_start).max())
    with np.errstate(invalid="ignore"):
        bins = np.digitize(anom, edges) - 1
        over = (anom >= edges[-1]) & ~np.isnan(anom)
    keys = np.zeros((366, nmax, C), dtype=np.uint16)
    aux = np.zeros((366, C), dtype=np.uint32)
    for d in range(366):
        rows = cal.doy_rows[cal.doy_start[d]:cal.doy_start[d + 1]]
        if rows.size == 0:
            continue
        b = bins[rows]                                    # [n, C]
        valid = b < nb
        k = np.where(valid, ((b + 1) << 7) | np.arange(rows.size)[:, None], 0).astype(np.uint16)
        keys[d, :rows.size] = -np.sort(-k.astype(np.int32), axis=0)
        ov = over[rows]                                   # [n, C] beyond the table, in time order
        n_ov = ov.sum(axis=0)
        order = np.argsort(~ov, axis=0, kind="stable")    # positions of the True entries first, ascending
        p1 = np.where(n_ov >= 1, order[0], 0).astype(np.uint32)
        p2 = np.where(n_ov >= 2, order[1] if rows.size > 1 else 0, 0).astype(np.uint32)
        aux[d] = (valid.sum(axis=0).astype(np.uint32) | np.where(n_ov >= 1, 0x8000, 0).astype(np.uint32) | (p1 << 16) | (p2 << 23)
                  | np.where(n_ov >= 2, 0x40000000, 0).astype(np.uint32) | np.where(n_ov >= 3, 0x80000000, 0).astype(np.uint32))
    return keys, aux


def device_tails_to_keys(tl, C):
    """Device lists [366, NPER, NCH, C, 8] -> (all keys of a bucket sorted descending [366, NPER*NCH*8, C], every list sorted?)."""
    t = tl["tails"].cpu().numpy().view(np.uint16)          # [366, NPER, NCH, C, 8]
    nper, nch = t.shape[1], t.shape[2]
    per_list = np.ascontiguousarray(t.transpose(0, 1, 2, 4, 3)).reshape(366, nper, nch * 8, C)   # keys of list p in stored order
    lists_sorted = bool((np.diff(per_list.astype(np.int32), axis=2) <= 0).all())
    allk = per_list.reshape(366, nper * nch * 8, C)
    allk = -np.sort(-allk.astype(np.int32), axis=1)
    return allk.astype(np.uint16), tl["aux"].cpu().numpy().view(np.uint32), lists_sorted


def make_anomalies(T_years=12, C=300, seed=3, start="2000-01-01", sigma=0.8):
    rng = np.random.default_rng(seed)
    tm = calendar.daily_time_axis(start, T_years * 365 + 3)
    cal = calendar.build_calendar(tm)
    anom = rng.normal(0, sigma, (len(tm), C)).astype(np.float32)
    return tm, cal, anom, rng


@pytest.mark.parametrize("list_rows", [32, 15])
@pytest.mark.parametrize("years,C", [(12, 300), (40, 257), (3, 1024), (100, 64)])
def test_tail_extract_matches_its_definition(hot, years, C, list_rows):
    tm, cal, anom, rng = make_anomalies(years, C)
    pick = rng.random(anom.shape)
    anom[pick < 0.02] = np.nan
    anom[(pick > 0.02) & (pick < 0.025)] = np.inf
    anom[(pick > 0.025) & (pick < 0.03)] = -np.inf
    anom[(pick > 0.03) & (pick < 0.035)] = 7.5                      # beyond the table: dropped from the counts, flagged
    anom[:, 5] = np.nan                                             # land
    anom[:, 6] = 0.25                                               # all samples in one bin: ties broken by position
    bt = binning.hobday_bins()
    dcal = hot.upload_calendar(cal)
    if (cal.doy_start[1:] - cal.doy_start[:-1]).max() > 6 * list_rows:
        pytest.skip("more than 6 lists per bucket")
    tl = hot.tail_extract(torch.from_numpy(anom).to(hot.device), dcal, bt, list_rows=list_rows)
    hot.sync()
    keys, aux, lists_sorted = device_tails_to_keys(tl, C)
    ekeys, eaux = tails_reference(anom, cal, bt.edges)
    assert lists_sorted                                             # every list descending (consumers stop at the first miss)
    assert np.array_equal(aux, eaux)
    n = ekeys.shape[1]
    assert np.array_equal(keys[:, :n], ekeys) and not keys[:, n:].any()


def _thr_case(hot, anom, cal, bt, pct, wd, ws, ny, nx, opts=None, rows=None, list_rows=None):
    dcal = hot.upload_calendar(cal)
    ad = torch.from_numpy(anom).to(hot.device)
    tl = hot.tail_extract(ad, dcal, bt, list_rows=list_rows)
    hot.ctx.debug_counters(reset=True)
    with hot.ctx.options(**(opts or {})):
        t = hot.hobday_thresholds_tails(tl, ad, dcal, bt, pct / 100.0, wd, ws or 1, ny, nx, rows=rows)
        m = hot.mask_ge_doy_tails(tl, ad, t["thr_doy_major"], dcal, bt)
        hot.sync()
    counters = hot.ctx.debug_counters(reset=True)
    exp_thr, exp_stats = orc.hobday_thresholds_approx(anom, cal.doy_out, pct / 100.0, wd, ws, bt.edges, bt.centres, ny, nx)
    exp_ext = orc.mask_ge_doy(anom, exp_thr, cal.doy_out)
    thr = t["thr_doy_major"].cpu().numpy().T
    if rows is not None:
        sl = slice(rows[0] * nx, rows[1] * nx)
        assert np.array_equal(thr[sl], exp_thr[sl], equal_nan=True)
        return counters
    assert np.array_equal(thr, exp_thr, equal_nan=True), "thresholds differ from the oracle"
    assert np.array_equal(m["extreme"].cpu().numpy().astype(bool), exp_ext), "mask differs from the oracle"
    assert int(m["n_true"].item()) == int(exp_ext.sum())
    st = HotPath.decode_thr_stats(t["stats_dev"])
    assert st["n_too_low"] == exp_stats["n_too_low"] and st["n_too_high"] == exp_stats["n_too_high"]
    return counters


def test_low_quantiles_walk_deep_into_the_lists(hot):
    """40 samples per bucket (2 lists), q = 0.6: 16 samples of a bucket lie above the quantile, so second and third chunks are
    read and most of every list is inside the band; q = 0.95 on the same field touches first chunks only."""
    tm, cal, anom, rng = make_anomalies(40, 12 * 20, seed=11)
    anom[:, 7] = np.nan
    bt = binning.hobday_bins()
    _thr_case(hot, anom, cal, bt, 60.0, 11, 5, 12, 20)
    _thr_case(hot, anom, cal, bt, 95.0, 11, 5, 12, 20)
    _thr_case(hot, anom, cal, bt, 60.0, 11, 5, 12, 20, list_rows=15)   # the geometry the anomaly kernel emits: 3 lists of 15
    _thr_case(hot, anom, cal, bt, 95.0, 5, 3, 12, 20, list_rows=15)
    # 100 samples per bucket: 4 lists (the largest instance of the kernels), 128 rows is the format's limit
    tm, cal, anom, rng = make_anomalies(100, 6 * 10, seed=12)
    _thr_case(hot, anom, cal, bt, 95.0, 11, 5, 6, 10)
    _thr_case(hot, anom, cal, bt, 90.0, 5, 3, 6, 10)


@pytest.mark.parametrize("tile", [16, 32])
def test_seasonal_and_patchy_thresholds_move_the_band(hot, tile):
    """The 95th percentile swings by 2.4 K over the year (3.75 bands of 64 bins) and differs by 1.5 K between the two halves
    of the grid (more than two bands inside one tile): the band is rebuilt as the walk goes and extra passes answer the cells
    it cannot hold at once -- same bits as the oracle."""
    tm, cal, _, rng = make_anomalies(30, 40 * 36, seed=5)
    doy = cal.doy_out.astype(np.float64)
    amp = 0.5 + 0.45 * np.sin(2 * np.pi * (doy - 40) / 366.0)                   # sigma 0.05 .. 0.95 K
    field = rng.normal(0, 1, (cal.T_out, 40 * 36)).astype(np.float32) * amp[:, None].astype(np.float32)
    field = field.reshape(-1, 40, 36)
    field[:, :, 18:] *= np.float32(2.2)                                          # patchy: right half far more variable
    field[:, 20:, :] += np.float32(0.3)
    anom = np.ascontiguousarray(field.reshape(cal.T_out, -1))
    bt = binning.hobday_bins()
    c = _thr_case(hot, anom, cal, bt, 95.0, 11, 5, 40, 36, opts={"THR_TILE": tile, "THR_DD": 366})
    assert c[0] > c[3] / 366 and c[2] > 0, c       # more rebuilds than one per block; stragglers needed extra passes
    _thr_case(hot, anom, cal, bt, 90.0, 5, 3, 40, 36, opts={"THR_TILE": tile})


def test_straggler_pass_limit_is_an_error_not_a_result(hot):
    """The list threshold kernel gives up on an output after `pass_limit` straggler passes (never reached in the product:
    tests/test_tail_constants.py).  With the limit forced to 0 on a field that NEEDS extra passes, the unresolved outputs are
    counted in marex_thr_stats.n_unresolved and the host raises instead of handing NaN thresholds on."""
    from marex_amd.exceptions import ProcessingError

    tm, cal, _, rng = make_anomalies(30, 40 * 36, seed=5)
    field = rng.normal(0, 1, (cal.T_out, 40, 36)).astype(np.float32) * np.float32(0.4)
    field[:, :, 18:] *= np.float32(2.5)                                          # two bands of thresholds inside one tile
    anom = np.ascontiguousarray(field.reshape(cal.T_out, -1))
    bt = binning.hobday_bins()
    dcal = hot.upload_calendar(cal)
    ad = torch.from_numpy(anom).to(hot.device)
    tl = hot.tail_extract(ad, dcal, bt)
    with hot.ctx.options(THR_PASS_LIMIT=0, THR_TILE=32):
        t = hot.hobday_thresholds_tails(tl, ad, dcal, bt, 0.95, 11, 5, 40, 36)
        hot.sync()
    with pytest.raises(ProcessingError, match="unresolved"):
        HotPath.decode_thr_stats(t["stats_dev"])
    t = hot.hobday_thresholds_tails(tl, ad, dcal, bt, 0.95, 11, 5, 40, 36)     # the product limit: resolved, no error
    hot.sync()
    assert "n_too_low" in HotPath.decode_thr_stats(t["stats_dev"])


def test_owned_rows_and_unstructured(hot):
    tm, cal, anom, rng = make_anomalies(20, 23 * 31, seed=8)
    bt = binning.hobday_bins()
    _thr_case(hot, anom, cal, bt, 95.0, 11, 5, 23, 31, rows=(2, 19))
    _thr_case(hot, anom, cal, bt, 95.0, 11, 7, 23, 31)
    _thr_case(hot, anom[:, :700].copy(), cal, bt, 95.0, 11, None, 0, 700)            # no pooling, C not a multiple of 256
    _thr_case(hot, anom[:, :700].copy(), cal, bt, 100.0, 21, None, 0, 700)           # q = 1: quantile runs off the table end


def test_constant_and_extreme_data(hot):
    """All samples in one bin (ties), everything below the table's first edge, everything beyond the last (empty histograms)."""
    tm, cal, anom, rng = make_anomalies(18, 10 * 16, seed=2)
    anom[:, 0:40] = np.float32(0.0)            # constant -> thresholds clamp to edges[3], warning counter
    anom[:, 40:80] = np.float32(-3.0)          # everything in bin 0
    anom[:, 80:120] = np.float32(9.0)          # nothing countable: NaN thresholds, mask decided on the values
    anom[:, 120:130] = np.float32(4.985)       # the last countable bins: above the upper warning bound
    bt = binning.hobday_bins()
    c = _thr_case(hot, anom, cal, bt, 95.0, 11, 5, 10, 16)
    assert c[4] > 0, c            # buckets with more than two values beyond the table: the mask looked at the anomalies
    _thr_case(hot, anom, cal, bt, 95.0, 11, 1, 10, 16)


@pytest.mark.parametrize("W,years,start", [(15, 100, "1925-01-01"), (5, 23, "2001-03-17"), (4, 9, "2003-01-01"), (13, 40, "1980-06-01")])
def test_anomaly_kernel_emits_the_tails_of_its_own_output(hot, W, years, start):
    """marex_shifting_baseline_tails_f32: same anomalies as the plain entry point, and lists holding exactly the keys the
    extraction kernel finds in those anomalies (lists of 15 output years; mid-year starts, leap days, the 6-list limit)."""
    tm = calendar.daily_time_axis(start, years * 365 + years // 4)
    x = synth.synth_field(synth.make_tables(tm, 5, 29))
    ocean = np.flatnonzero(np.isfinite(x[0]))
    x[200:230, ocean[0]] = np.nan
    x[:, ocean[1]] += np.float32(30.0) * (np.arange(x.shape[0]) % 11 == 0)     # values beyond the table
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    xd = torch.from_numpy(x).to(hot.device)
    assert hot.shifting_tails_ok(dcal)
    a = hot.shifting_baseline_tails(xd, dcal, W, 21, bt)
    b = hot.shifting_baseline(xd, dcal, W, 21, None)
    hot.sync()
    assert np.array_equal(a["out"].cpu().numpy(), b["out"].cpu().numpy(), equal_nan=True)
    assert np.array_equal(a["invalid_count"].cpu().numpy(), b["invalid_count"].cpu().numpy())
    k1, a1, s1 = device_tails_to_keys(a["tails"], x.shape[1])
    ref = hot.tail_extract(a["out"], dcal, bt, list_rows=15)
    hot.sync()
    k2, a2, s2 = device_tails_to_keys(ref, x.shape[1])
    assert s1 and s2 and np.array_equal(a1, a2) and np.array_equal(k1, k2)


@pytest.mark.parametrize("years,pct,wd,list_rows", [(15, 95.0, 11, 15), (30, 95.0, 11, 15), (5, 90.0, 5, 15), (40, 95.0, 31, 32),
                                                     (85, 99.0, 11, 15), (22, 60.0, 11, 32), (22, 100.0, 3, 15)])
def test_per_cell_threshold_kernel_without_spatial_pooling(hot, years, pct, wd, list_rows):
    """window_spatial_hobday = 1 (unstructured meshes): one lane per cell, bisection on the bin index over the first chunks of
    the window's lists.  Short and long records, both list geometries, wide day windows, q = 0.6 (bounds from first chunks do
    not decide every probe: the exact recount runs) and q = 1."""
    tm, cal, anom, rng = make_anomalies(years, 333, seed=years)
    anom[:, 5] = np.nan
    anom[: anom.shape[0] // 2, 6] = np.nan           # half a record: fewer samples, other totals than its neighbours
    anom[:, 7] = np.float32(0.25)                     # ties: every sample in one bin
    anom[:, 8] = np.float32(7.0)                      # nothing countable
    bt = binning.hobday_bins()
    c = _thr_case(hot, anom, cal, bt, pct, wd, None, 0, 333, list_rows=list_rows, opts={"THR_CELLS": 2})  # forced: by default only
    if pct == 60.0:                                                                                    # windows of <= 16 lists
        assert c[1] > 0, c                            # probes the first chunks could not decide
    c0 = _thr_case(hot, anom, cal, bt, pct, wd, None, 0, 333, list_rows=list_rows, opts={"THR_CELLS": 0})  # the tile kernel, same bits
    assert c0[1] == 0
    if years <= 30:  # gridded field asked for ws = 1, owned rows only, several day blocks
        _thr_case(hot, anom[:, :330].copy(), cal, bt, pct, wd, 1, 11, 30, rows=(2, 9), list_rows=list_rows,
                  opts={"THR_CELLS": 2, "THR_CELLS_BLOCKS": 5})
        _thr_case(hot, anom[:, :330].copy(), cal, bt, pct, wd, 1, 11, 30, rows=(2, 9), list_rows=list_rows)  # the default choice


@pytest.mark.parametrize("per_bucket", [1, 2, 3])
def test_outliers_beyond_the_table_stay_on_the_list_path(hot, per_bucket):
    """Samples at or beyond ``max_anomaly`` have no key (the histogram drops them) but are extremes of every finite
    threshold.  One or two per bucket travel in the aux word and the mask places them without reading the anomalies (no slow
    group: the mask kernel's time does not depend on outliers); three or more fall back to the compare on the values.  Same
    bits as the oracle either way, for both list geometries and with / without pooling."""
    tm, cal, anom, rng = make_anomalies(20, 12 * 16, seed=21)
    anom[:, 3] = np.nan
    for d in range(366):  # `per_bucket` samples of every bucket of every third cell go beyond the table (+inf among them)
        rows = cal.doy_rows[cal.doy_start[d]:cal.doy_start[d + 1]]
        for c in range(0, anom.shape[1], 3):
            pick = rng.choice(rows, size=min(per_bucket, rows.size), replace=False)
            anom[pick, c] = np.float32(5.0 + rng.random()) if (c // 3) % 2 == 0 else np.float32(np.inf)
    anom[:, 3] = np.nan
    bt = binning.hobday_bins()
    for list_rows, ws, ny, nx in ((15, 5, 12, 16), (32, 5, 12, 16), (15, None, 0, 12 * 16)):
        c = _thr_case(hot, anom, cal, bt, 95.0, 11, ws, ny, nx, list_rows=list_rows)
        assert (c[4] == 0) == (per_bucket <= 2), (per_bucket, c)


@pytest.mark.parametrize("years,C", [(12, 300), (40, 257), (60, 130), (100, 64)])
def test_fixed_baseline_kernels_emit_the_tails_of_their_own_output(hot, years, C):
    """marex_fixed_baseline_tails_f32 / marex_detrend_fixed_baseline_tails_f32: the anomalies of the plain entry points, and key
    lists + aux words that are BYTE for byte what the extraction kernel makes of those anomalies (32 rows per list; buckets of
    12 / 40 rows: the 48-row register kernel, one / two lists; 60 / 100 rows: the 128-row kernel, two / four lists) -- NaN gaps,
    land, +-inf, values beyond the table, ties."""
    rng = np.random.default_rng(100 + years)
    tm = calendar.daily_time_axis("1950-03-01", years * 365 + years // 4)
    cal = calendar.build_calendar(tm)
    x = (rng.normal(0, 0.9, (len(tm), C)) + 3.0 * np.sin(2 * np.pi * cal.doy / 365.25)[:, None]
         + 0.0004 * np.arange(len(tm))[:, None]).astype(np.float32)
    pick = rng.random(x.shape)
    x[pick < 0.01] = np.nan
    x[(pick > 0.01) & (pick < 0.012)] = np.inf
    x[(pick > 0.012) & (pick < 0.014)] = -np.inf
    x[(pick > 0.014) & (pick < 0.018)] += np.float32(9.0)           # beyond the table, often several per bucket
    x[:, 5] = np.nan                                                # land
    x[:, 6] = np.float32(0.25)                                      # ties: every anomaly 0
    bt = binning.hobday_bins()
    dcal = hot.upload_calendar(cal)
    xd = torch.from_numpy(x).to(hot.device)

    def same_tails(a, out):
        ref = hot.tail_extract(out, dcal, bt)
        hot.sync()
        assert a["tails"]["list_rows"] == ref["list_rows"] == 32 and a["tails"]["max_bucket"] == ref["max_bucket"]
        assert torch.equal(a["tails"]["aux"], ref["aux"]), "aux words differ from the extraction kernel's"
        assert torch.equal(a["tails"]["tails"], ref["tails"]), "key lists differ from the extraction kernel's"

    hot.ctx.set_option("FIXED_TAILS", 2)  # whenever possible (the default stops at 48-row buckets: the 128-row kernel is slower fused)
    try:
        _fixed_tails_case(hot, xd, dcal, bt, tm, same_tails)
    finally:
        hot.ctx.set_option("FIXED_TAILS", None)
    with hot.ctx.options(FIXED_TAILS=0):                            # switched off: no lists, the extraction pass makes them later
        assert "tails" not in hot.fixed_baseline(xd, dcal, None, None, tails_bins=bt)
    assert ("tails" in hot.fixed_baseline(xd, dcal, None, None, tails_bins=bt)) == (years <= 48)   # the default


def _fixed_tails_case(hot, xd, dcal, bt, tm, same_tails):
    plain = hot.fixed_baseline(xd, dcal, None, None)
    fused = hot.fixed_baseline(xd, dcal, None, None, tails_bins=bt)
    hot.sync()
    assert "tails" in fused and np.array_equal(fused["out"].cpu().numpy(), plain["out"].cpu().numpy(), equal_nan=True)
    assert torch.equal(fused["invalid_count"], plain["invalid_count"]) and torch.equal(fused["mask"], plain["mask"])
    same_tails(fused, fused["out"])
    rp = (1955, 1958)                                               # reference period: the climatology of a few years only
    fused_rp = hot.fixed_baseline(xd, dcal, rp, None, tails_bins=bt)
    hot.sync()
    same_tails(fused_rp, fused_rp["out"])
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2], False)
    xf = torch.nan_to_num(xd, nan=0.0, posinf=5.0, neginf=-5.0)     # the fit needs numbers; land stays land
    xf[:, 5] = float("nan")
    d0 = hot.detrend_fixed_baseline(xf, model, pmodel, True, dcal)
    out0 = d0["out"].clone()
    d1 = hot.detrend_fixed_baseline(xf, model, pmodel, True, dcal, tails_bins=bt)
    hot.sync()
    assert "tails" in d1 and np.array_equal(d1["out"].cpu().numpy(), out0.cpu().numpy(), equal_nan=True)
    same_tails(d1, d1["out"])


@pytest.mark.parametrize("years,tile,dd", [(7, 16, 0), (7, 16, 9), (30, 16, 0), (30, 32, 0), (12, 32, 61)])
def test_bin_matrix_band_kernel_follows_seasonal_and_patchy_thresholds(hot, years, tile, dd):
    """k_thr_band (the bin-matrix path: short buckets with pooling, cfg2) with the band that FOLLOWS the thresholds (round 4):
    a 95th percentile that swings by 2.4 K over the year and differs by 1.5 K between the two halves of the grid -- rebuilds as
    the walk goes, straggler passes for cells the band cannot hold at once -- gives the oracle's bits; so do 5-sample buckets
    (7 years), both tile sizes, short and long day blocks, q = 1 and a low quantile."""
    tm, cal, _, rng = make_anomalies(years, 40 * 36, seed=50 + years)
    doy = cal.doy_out.astype(np.float64)
    amp = 0.5 + 0.45 * np.sin(2 * np.pi * (doy - 40) / 366.0)
    field = rng.normal(0, 1, (cal.T_out, 40 * 36)).astype(np.float32) * amp[:, None].astype(np.float32)
    field = field.reshape(-1, 40, 36)
    field[:, :, 18:] *= np.float32(2.2)
    field[:, 20:, :] += np.float32(0.3)
    field[:, 3, 5] = np.nan                     # a land cell inside
    anom = np.ascontiguousarray(field.reshape(cal.T_out, -1))
    bt = binning.hobday_bins()
    dcal = hot.upload_calendar(cal)
    ad = torch.from_numpy(anom).to(hot.device)
    binsb = hot.digitize(ad, dcal, bt)
    opts = {"THR_TILE": tile}
    if dd:
        opts["THR_DD"] = dd
    for pct, wd, ws in ((95.0, 11, 5), (100.0, 5, 3), (60.0, 11, 5)):
        with hot.ctx.options(**opts):
            t = hot.hobday_thresholds(binsb, ad, dcal, bt, pct / 100.0, wd, ws, 40, 36)
            hot.sync()
        exp_thr, exp_stats = orc.hobday_thresholds_approx(anom, cal.doy_out, pct / 100.0, wd, ws, bt.edges, bt.centres, 40, 36)
        assert np.array_equal(t["thr_doy_major"].cpu().numpy().T, exp_thr, equal_nan=True), (pct, wd, ws)
        st = HotPath.decode_thr_stats(t["stats_dev"])
        assert st["n_too_low"] == exp_stats["n_too_low"] and st["n_too_high"] == exp_stats["n_too_high"]
    with hot.ctx.options(THR_EXACT_PATH=1, **opts):  # the exact path (coarse + fine sweeps) is still there: same bits
        t = hot.hobday_thresholds(binsb, ad, dcal, bt, 0.95, 11, 5, 40, 36, rows=(3, 31))
        hot.sync()
    exp_thr, _ = orc.hobday_thresholds_approx(anom, cal.doy_out, 0.95, 11, 5, bt.edges, bt.centres, 40, 36)
    sl = slice(3 * 36, 31 * 36)
    assert np.array_equal(t["thr_doy_major"].cpu().numpy().T[sl], exp_thr[sl], equal_nan=True)
