"""GPU: latitude-band shards with overlap rows, processed one after the other on one device, stitch to
exactly the unsharded result (SURVEY.md 8e "testability"); plus size-independent properties on a larger field."""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar, synth
from marex_amd.dist import plan_shards, stitch_cells

pytestmark = pytest.mark.gpu


def _run(hot, tm, cal, dcal, bt, ny_global, nx, shard=None, W=5):
    if shard is None:
        tab = synth.make_tables(tm, ny_global, nx)
        x = hot.synth_field(tab)
        r = hot.shifting_hobday(x, dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=ny_global, nx=nx)
    else:
        tab = synth.make_tables(tm, shard.ny_in, nx, lat_range=(shard.in0, shard.in1, ny_global))
        x = hot.synth_field(tab, cell_base=shard.cell_base)
        own = (shard.own0 - shard.in0, shard.own1 - shard.in0)
        r = hot.shifting_hobday(x, dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=shard.ny_in, nx=nx, own_rows=own)
    hot.sync()
    return {k: r[k].cpu().numpy() for k in ("dat_anomaly", "extreme_events", "thr_doy_major", "mask")}, int(r["n_true"].item())


@pytest.mark.parametrize("world", [2, 3, 8])
def test_band_shards_stitch_bit_identical(hot, world):
    ny, nx, W = 40, 48, 5
    tm = calendar.daily_time_axis("2010-01-01", 10 * 365 + 3)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    full, n_full = _run(hot, tm, cal, dcal, bt, ny, nx)
    shards = plan_shards(ny, nx, world, 2)
    parts, n_sum = [], 0
    for sh in shards:
        p, n = _run(hot, tm, cal, dcal, bt, ny, nx, shard=sh)
        parts.append(p)
        n_sum += n
    for key in ("dat_anomaly", "extreme_events", "thr_doy_major", "mask"):
        got = stitch_cells([p[key] for p in parts], shards)
        assert np.array_equal(got, full[key], equal_nan=True), key
    assert n_sum == n_full == int(full["extreme_events"].sum())  # the kernel counts owned cells only


def test_full_width_properties(hot):
    """1440-wide rows (the benchmark grid's width) x 48 rows: determinism, land handling, frequency, counters."""
    ny, nx, W = 48, 1440, 5
    tm = calendar.daily_time_axis("2015-01-01", 3652)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    a, na = _run(hot, tm, cal, dcal, bt, ny, nx)
    b, nb_ = _run(hot, tm, cal, dcal, bt, ny, nx)
    for key in a:
        assert np.array_equal(a[key], b[key], equal_nan=True)  # idempotent / deterministic
    ocean = a["mask"].astype(bool)
    assert 0.5 < ocean.mean() < 0.9
    assert np.isnan(a["dat_anomaly"][:, ~ocean]).all() and not a["extreme_events"][:, ~ocean].any()
    assert np.isnan(a["thr_doy_major"][:, ~ocean]).all() and (a["thr_doy_major"][:, ocean] >= bt.lower_bound).all()
    freq = a["extreme_events"][:, ocean].mean()
    assert abs(freq - 0.05) < 0.01
    assert na == nb_ == int(a["extreme_events"].sum())
    # mask == anomaly >= threshold[doy] recomputed on the host from the device outputs
    exp = a["dat_anomaly"] >= a["thr_doy_major"][cal.doy_out.astype(np.int64) - 1]
    assert np.array_equal(a["extreme_events"].astype(bool), exp)


@pytest.mark.parametrize("nstream", [2, 3])
def test_engine_set_round_robin_equals_one_engine(hot, nstream):
    """The product schedule of a rank with several bands (marex_amd.dist.EngineSet: bands round-robin over engines that own a
    HIP stream and a workspace each) gives the numbers of one engine working through the bands in order."""
    from marex_amd.dist import EngineSet, shard_step

    ny, nx, W, world = 40, 48, 5, 5
    tm = calendar.daily_time_axis("2010-01-01", 10 * 365 + 3)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    shards = plan_shards(ny, nx, world, 2)
    xs = [hot.synth_field(synth.make_tables(tm, sh.ny_in, nx, lat_range=(sh.in0, sh.in1, ny)), cell_base=sh.cell_base) for sh in shards]
    kw = dict(W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, nx=nx)
    r1, local1, mx1 = shard_step(hot, shards, xs, dcal, workspace={}, **kw)
    hot.sync()
    es = EngineSet(0, nstream)
    for _ in range(2):  # the second pass reuses the engines' workspaces
        r2, local2, mx2 = shard_step(es, shards, xs, cal, **kw)
    torch.cuda.synchronize()
    assert local1.tolist() == local2.tolist() and mx1.tolist() == mx2.tolist()
    assert int(local1[3]) > 0
    own = shards[-1].own_cell_slice()  # the last band, whichever engine ran it; thresholds / extremes exist for owned cells only
    for key in ("dat_anomaly", "extreme_events", "thr_doy_major", "mask"):
        a, b = r1[key].cpu().numpy(), r2[key].cpu().numpy()
        if key in ("extreme_events", "thr_doy_major"):
            a, b = a[..., own], b[..., own]
        assert np.array_equal(a, b, equal_nan=True), key
    tot, n = es.timing_get("shifting")
    assert n == 0  # timing was never enabled
