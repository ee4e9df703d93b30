"""GPU parity: shifting_baseline + hobday_extreme (approximate) through the C ABI vs the oracle.

Integer / boolean outputs (bins, mask, extreme_events, validation counts) and -- because the arithmetic
contract fixes every rounding -- the float32 anomalies and thresholds must be BIT-identical.
"""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _same_f32(got, exp):
    """Equal as numbers (NaN == NaN, -0 == +0)."""
    got = np.asarray(got, dtype=np.float32)
    exp = np.asarray(exp, dtype=np.float32)
    return np.array_equal(got, exp, equal_nan=True)


PATHS = ("tails", "bins")  # representation of the dayofyear histograms behind the thresholds: both must give the oracle's bits


def run_case(hot, start, periods, ny, nx, W, S, wd, ws, pct=95.0, unstructured=False, seed=20240607, mutate=None, path=None,
             opts=None):
    tm = calendar.daily_time_axis(start, periods)
    tab = synth.make_tables(tm, 0 if unstructured else ny, nx if not unstructured else ny * nx, seed, unstructured=unstructured)
    x = synth.synth_field(tab)
    if mutate is not None:
        mutate(x)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    gny, gnx = (0, x.shape[1]) if unstructured else (ny, nx)
    exp = orc.preprocess_arrays(
        x, cal, ny=gny, nx=gnx, window_year_baseline=W, smooth_days_baseline=S, window_days_hobday=wd,
        window_spatial_hobday=ws, threshold_percentile=pct, edges=bt.edges, centres=bt.centres,
    )
    dcal = hot.upload_calendar(cal)
    xd = torch.from_numpy(x).to(hot.device)
    hot.hobday_path = path
    try:
        with hot.ctx.options(**(opts or {})):
            got = hot.shifting_hobday(xd, dcal, W=W, S=S, bins=bt, q=pct / 100.0, wd=wd, ws=(ws or 1), ny=gny, nx=gnx)
            hot.sync()
    finally:
        hot.hobday_path = None
    if path is not None:
        assert got["path"] == path
    return x, cal, bt, exp, got


def check_all(x, cal, bt, exp, got):
    anom = got["dat_anomaly"].cpu().numpy()
    assert _same_f32(anom, exp["dat_anomaly"]), "anomalies differ from the oracle"
    assert np.array_equal(got["mask"].cpu().numpy().astype(bool), exp["mask"])
    v = orc.validate_data_values(x)
    inv = got["invalid_count"].cpu().numpy()
    assert np.array_equal(inv, (~np.isfinite(x)).sum(axis=0))
    assert int(np.where(exp["mask"], inv, 0).max()) == v["max_invalid"]
    if got["path"] == "bins":  # bins: device rows are dayofyear-sorted
        bins_exp = orc.digitize_bins(exp["dat_anomaly"], bt.edges)[cal.doy_rows]
        bins_got = HotPath.bins_to_rows(got["_keep"][1][0], x.shape[1]).cpu().numpy().view(np.uint16)
        assert np.array_equal(bins_got, bins_exp)
    thr = got["thresholds"].cpu().numpy()
    assert _same_f32(thr, exp["thresholds"]), "thresholds differ from the oracle"
    ext = got["extreme_events"].cpu().numpy().astype(bool)
    assert np.array_equal(ext, exp["extreme_events"]), "extreme_events mask differs"
    assert int(got["n_true"].item()) == int(exp["extreme_events"].sum())
    st = HotPath.decode_thr_stats(got["stats_dev"])
    assert st["n_too_low"] == exp["stats"]["n_too_low"] and st["n_too_high"] == exp["stats"]["n_too_high"]
    if np.isfinite(exp["stats"]["max"]):
        assert np.float32(st["max"]) == np.float32(exp["stats"]["max"])
        assert np.float32(st["min"]) == np.float32(exp["stats"]["min"])


@pytest.mark.parametrize("path", PATHS)
def test_gridded_default_windows(hot, path):
    """30 yr x 12x16, W=15, S=21, wd=11, ws=5 (the reference defaults for gridded data)."""
    r = run_case(hot, "1990-01-01", 30 * 365 + 8, 12, 16, 15, 21, 11, 5, path=path)
    check_all(*r)
    ocean = r[3]["mask"]
    freq = r[3]["extreme_events"][:, ocean].mean()
    assert 0.04 < freq < 0.06  # the reference's own pin: 5 % +- 1 % (tests/conftest.py:215-231)


@pytest.mark.parametrize("path", PATHS)
def test_gridded_mid_year_start_small_windows(hot, path):
    """Series starting mid-year (partial first / last calendar years), W=5, even S, wd=5, ws=3."""
    r = run_case(hot, "2001-07-19", 12 * 365 + 40, 9, 20, 5, 10, 5, 3, pct=90.0, path=path)
    check_all(*r)


@pytest.mark.parametrize("path", PATHS)
def test_unstructured_no_pooling(hot, path):
    """(time, ncells) layout, C not a multiple of 4 or 256, no spatial pooling (detect.py:1361-1385)."""
    r = run_case(hot, "1995-01-01", 20 * 365 + 5, 1, 405, 7, 21, 11, None, unstructured=True, path=path)
    check_all(*r)


def test_generic_smoothing_widths(hot):
    for S in (1, 5, 31):
        r = run_case(hot, "2000-01-01", 10 * 365 + 3, 5, 8, 4, S, 11, 5, pct=95.0)
        check_all(*r)


@pytest.mark.parametrize("path", PATHS)
def test_weird_cells(hot, path):
    """NaN at t=0 but finite later ("land" by the t=0 rule), NaN gaps inside ocean cells, +-inf, constant cells."""

    def mutate(x):
        ocean = np.flatnonzero(np.isfinite(x[0]))
        x[:40, ocean[0]] = np.nan          # NaN start -> classified as land, finite later
        x[500:520, ocean[1]] = np.nan      # gap inside an ocean cell (validation counts it)
        x[900, ocean[2]] = np.inf
        x[:, ocean[3]] = np.float32(12.5)  # constant series -> anomaly 0 -> clamp to edges[3]
        x[:, ocean[4]] += np.float32(40.0) * (np.arange(x.shape[0]) % 7 == 0)  # spikes beyond max_anomaly

    r = run_case(hot, "1998-03-01", 14 * 365 + 4, 6, 12, 6, 21, 11, 5, mutate=mutate, path=path)
    check_all(*r)


def test_threshold_kernel_variants(hot):
    """Bin-matrix kernels: band algorithm with other day-block lengths and the sliding-histogram kernel (16/32-bit counters,
    other segment widths); tail kernels: other tiles / day-block lengths.  All give the same bits as the default."""
    variants = (
        ("bins", {"THR_DD": 1}), ("bins", {"THR_DD": 5}), ("bins", {"THR_DD": 32}),
        ("bins", {"THR_EXACT_PATH": 1}), ("bins", {"THR_TILE": 32}), ("bins", {"THR_TILE": 32, "THR_EXACT_PATH": 1}),
        ("bins", {"THR_TILE": 3216}), ("bins", {"THR_TILE": 32, "THR_DD": 7, "THR_COARSE_PD": 5}),
        ("bins", {"MASK_BINS": 1}), ("bins", {"MASK_BINS": 1, "MASK_VEC": 4}), ("bins", {"MASK_BINS": 0}),
        ("bins", {"THR_ALGO": 1}), ("bins", {"THR_ALGO": 1, "THR_U32": 1}),
        ("bins", {"THR_ALGO": 1, "THR_NW": 5}), ("bins", {"THR_ALGO": 1, "THR_NW": 64, "THR_U32": 1}),
        ("tails", {"THR_DD": 1}), ("tails", {"THR_DD": 5}), ("tails", {"THR_DD": 366}), ("tails", {"THR_TILE": 32}),
        ("tails", {"THR_TILE": 32, "THR_DD": 7}), ("tails", {"THR_TILE": 32, "THR_TALL": 0}),
        ("tails", {"SHIFT_TAILS": 0}), ("tails", {"SHIFT_TAILS": 0, "THR_TILE": 32}),
    )
    for path, opts in variants:
        r = run_case(hot, "2003-01-01", 9 * 365 + 2, 19, 37, 4, 21, 11, 5, path=path, opts=opts)
        check_all(*r)


@pytest.mark.parametrize("path", PATHS)
def test_threshold_edge_quantiles_and_windows(hot, path):
    """q = 1.0 (searchsorted runs off the end), low q, wide day / spatial windows, tiny grids."""
    for pct, wd, ws, ny, nx in ((100.0, 11, 5, 6, 10), (60.0, 31, 3, 5, 9), (99.0, 5, 7, 20, 37), (95.0, 11, 5, 3, 4)):
        r = run_case(hot, "2001-01-01", 11 * 365 + 3, ny, nx, 3, 21, wd, ws, pct=pct, path=path)
        check_all(*r)


def test_shifting_chunk_variants(hot):
    """Every dayofyear-chunk width / history placement (registers or LDS ring) of the anomaly kernel gives
    identical bits, for W inside both ring capacities (8 and 16); with and without the fused bin matrix."""
    # SHIFT_D / SHIFT_FAST=0 route everything through the general kernel k_shifting
    for ring, D in ((0, 1), (0, 4), (1, 2), (1, 4), (1, 8), (1, "fast-off")):
        opts = {"SHIFT_RING": ring}
        opts.update({"SHIFT_FAST": 0} if D == "fast-off" else {"SHIFT_D": D})
        r = run_case(hot, "2003-01-01", 9 * 365 + 2, 7, 21, 4, 21, 11, 5, opts=opts, path="bins")
        check_all(*r)
        r = run_case(hot, "1990-06-01", 20 * 365 + 5, 5, 9, 13, 21, 11, 3, opts=opts, path="tails")
        check_all(*r)


@pytest.mark.parametrize("path", PATHS)
def test_fast_anomaly_kernel_instances_and_general_fallbacks(hot, path):
    """Every instantiated W of k_shift_fast, a W without an instance (general kernel), even W (real division),
    series starting mid-chunk and ending mid-year (edge years, partial chunks)."""
    for W, start, periods in ((3, "2004-01-01", 8 * 365 + 2), (6, "2001-03-17", 11 * 365 + 100), (7, "2000-01-01", 12 * 365 + 3),
                              (10, "1996-01-01", 16 * 365 + 4), (13, "1990-06-01", 20 * 365 + 5), (8, "1999-01-01", 13 * 365 + 3),
                              (16, "1990-01-01", 21 * 365 + 5)):
        r = run_case(hot, start, periods, 5, 9, W, 21, 11, 3, path=path)
        check_all(*r)


@pytest.mark.parametrize("path", PATHS)
def test_long_buckets_pick_the_big_tile(hot, path):
    """28 output years per dayofyear bucket on a 20x40 grid: the threshold entry points switch to 32x32 tiles."""
    r = run_case(hot, "1985-01-01", 32 * 365 + 8, 20, 40, 4, 21, 11, 5, path=path)
    check_all(*r)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("ny,nx,opts", [(30, 52, {}), (31, 27, {"THR_EXACT_PATH": 1}), (61, 53, {"THR_DD": 9}), (30, 52, {"THR_TALL": 0})])
def test_tall_tile_for_bands_it_tiles_better(hot, path, ny, nx, opts):
    """34 x 30 tiles (1020 cells on 1024 threads, four spare lanes) replace 32 x 32 where they need fewer tiles -- e.g. 30
    rows: one row of tiles instead of two; same bits as the oracle, also on the exact path and with short day blocks."""
    r = run_case(hot, "2001-01-01", 8 * 365 + 2, ny, nx, 3, 21, 11, 5, path=path, opts=dict(opts, THR_TILE=32))
    check_all(*r)


@pytest.mark.parametrize("W,years,start,nx", [(15, 40, "1980-01-01", 66), (5, 23, "2001-03-17", 52), (15, 33, "1990-07-01", 64)])
def test_lean_and_fast_anomaly_kernels_agree_on_awkward_fields(hot, W, years, start, nx):
    """k_shift_lean (lean records, straight-line body for regular years, LDS-DMA staging) against k_shift_fast (SHIFT_LEAN=0) on
    calendars and fields that leave the straight-line body: mid-year starts, a partial last cell group, gaps, a late-starting
    cell, values beyond the table, +-inf.  Same anomalies, counts, lists and aux words, bit for bit."""
    tm = calendar.daily_time_axis(start, years * 365 + years // 4)
    x = synth.synth_field(synth.make_tables(tm, 3, nx))
    ocean = np.flatnonzero(np.isfinite(x[0]))
    x[400:430, ocean[0]] = np.nan                      # a gap: NaN in the history for W years
    x[: 6 * 365, ocean[1]] = np.nan                    # a cell that starts late
    x[:, ocean[2]] += np.float32(40.0) * (np.arange(x.shape[0]) % 7 == 0)   # values beyond the table, several per bucket
    x[1234, ocean[3]] = np.inf
    x[2345, ocean[4]] = -np.inf
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    xd = torch.from_numpy(x).to(hot.device)
    res = {}
    for lean in (1, 0):
        with hot.ctx.options(SHIFT_LEAN=lean):
            a = hot.shifting_baseline_tails(xd, dcal, W, 21, bt)
            hot.sync()
        res[lean] = {k: a[k].cpu().numpy() for k in ("out", "mask", "invalid_count")}
        res[lean]["tails"] = a["tails"]["tails"].cpu().numpy()
        res[lean]["aux"] = a["tails"]["aux"].cpu().numpy()
    for k in res[1]:
        assert np.array_equal(res[1][k], res[0][k], equal_nan=res[1][k].dtype.kind == "f"), k
    exp, mask = orc.shifting_baseline_anomaly(x, cal, W, 21)
    assert np.array_equal(res[1]["out"], exp, equal_nan=True) and np.array_equal(res[1]["mask"].astype(bool), mask)


@pytest.mark.parametrize("path", PATHS)
def test_cell_groups_that_are_all_nan_for_some_years(hot, path):
    """The lean anomaly kernel skips the arithmetic of a wave-year whose 64 cells are all NaN on the centre rows.  Whole groups
    of 64 consecutive cells that are NaN from the start and come alive later, that fall silent for a few years in the middle,
    that are NaN on single days only (not skipped: neighbours of the centre rows), and plain land -- same bits as the oracle."""

    def mutate(x):
        T, C = x.shape
        assert C >= 256
        x[: 7 * 365 + 100, 0:64] = np.nan            # a group that starts late (NaN history, then nanmean years)
        x[9 * 365 : 12 * 365 + 17, 64:128] = np.nan  # a group with a three-year gap in the middle
        x[:, 128:192] = np.nan                       # land
        x[4000:4003, 192:256] = np.nan               # three days only: the years around them are not all-NaN on their centre rows
        x[5000, 192:256:2] = np.inf

    r = run_case(hot, "1991-01-01", 22 * 365 + 5, 4, 80, 5, 21, 11, 5, mutate=mutate, path=path)
    check_all(*r)
    r = run_case(hot, "1986-01-01", 34 * 365 + 8, 4, 80, 15, 21, 11, 5, mutate=mutate, path=path)
    check_all(*r)


@pytest.mark.parametrize("W,years,start,nx", [(5, 12, "2001-03-17", 52), (15, 33, "1990-07-01", 64), (5, 10, "2000-01-01", 80)])
def test_lean_and_fast_anomaly_kernels_agree_on_the_bin_matrix(hot, W, years, start, nx):
    """The bin-matrix variant of k_shift_lean (short series whose thresholds come from the band kernel) against k_shift_fast
    (SHIFT_LEAN_BINS=0) on the same awkward fields, including whole cell groups of NaN: same anomalies, counts and bins."""
    tm = calendar.daily_time_axis(start, years * 365 + years // 4)
    x = synth.synth_field(synth.make_tables(tm, 4, nx))
    ocean = np.flatnonzero(np.isfinite(x[0]))
    x[400:430, ocean[0]] = np.nan
    x[: 3 * 365, ocean[1]] = np.nan
    x[:, ocean[2]] += np.float32(40.0) * (np.arange(x.shape[0]) % 7 == 0)
    x[1234, ocean[3]] = np.inf
    x[2345, ocean[4]] = -np.inf
    x[: 2 * 365 + 50, 64:128] = np.nan  # a whole cell group that starts late
    x[:, 128:192] = np.nan              # land
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    xd = torch.from_numpy(x).to(hot.device)
    res = {}
    for lean in (1, 0):
        with hot.ctx.options(SHIFT_LEAN_BINS=lean):
            a = hot.shifting_baseline(xd, dcal, W, 21, bt)
            hot.sync()
        res[lean] = {k: a[k].cpu().numpy().copy() for k in ("out", "mask", "invalid_count", "bins")}
    for k in res[1]:
        assert np.array_equal(res[1][k], res[0][k], equal_nan=res[1][k].dtype.kind == "f"), k
    exp, mask = orc.shifting_baseline_anomaly(x, cal, W, 21)
    assert np.array_equal(res[1]["out"], exp, equal_nan=True) and np.array_equal(res[1]["mask"].astype(bool), mask)
    bins_exp = orc.digitize_bins(exp, bt.edges)[cal.doy_rows]
    bins_got = HotPath.bins_to_rows(torch.from_numpy(res[1]["bins"]).to(hot.device), x.shape[1]).cpu().numpy().view(np.uint16)
    assert np.array_equal(bins_got, bins_exp)


@pytest.mark.parametrize("rows,cols", [(366, 1), (366, 33), (366, 1000), (366, 4096), (12, 77)])
def test_threshold_transpose(hot, rows, cols):
    """[366, C] -> [C, 366] and any other shape (32 x 32 tiles through LDS)."""
    a = torch.arange(rows * cols, dtype=torch.float32, device=hot.device).reshape(rows, cols) * 0.5
    t = hot.transpose(a)
    hot.sync()
    assert t.shape == (cols, rows) and torch.equal(t, a.T.contiguous())
