"""GPU: the tiling the headline number is measured on (bench.py at N = 1: 6 latitude bands of 120 + 4 rows of the 100-yr field,
round-robin over two engines / HIP streams).  Bands 2 and 3 of ``plan_shards(720, 1440, 6, 2)`` at FULL size go through
``EngineSet(0, 2)`` + ``shard_step`` exactly as in the bench: four rows of 30-row threshold tiles per band (34 x 30-cell tiles,
row 0 of the tiles at the band's first owned row), owned rows only.  Checked: size-independent properties of both bands on the device, the step's summed counters, and bit parity with the
oracle on 5 x 24 cut-outs whose middle rows sit on either side of a tile-row boundary (and whose columns straddle a tile-column
boundary)."""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar, synth
from marex_amd.calendar import N_DOY
from marex_amd.dist import SUMMARY_KEYS, EngineSet, plan_shards, shard_step
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def _views(wsp, T_out, C):
    return {
        "anom": wsp["anom"][: T_out * C].view(T_out, C),
        "ext": wsp["extreme"][: T_out * C].view(T_out, C),
        "thr": wsp["thr_doy_major"][: N_DOY * C].view(N_DOY, C),
        "mask": wsp["mask"][:C].bool(),
        "n_true": int(wsp["n_true"][0].item()),
    }


def test_two_neighbouring_bands_of_the_six_band_tiling_on_two_streams(hot):
    import gc

    gc.collect()
    torch.cuda.empty_cache()  # earlier tests' cached blocks go back: this one needs half the card
    free = torch.cuda.mem_get_info(0)[0]
    if free < 170e9:
        pytest.skip(f"needs ~160 GB of free HBM, {free / 1e9:.0f} GB available")
    ny_g, nx, T, W, halo = 720, 1440, 36500, 15, 2
    tm = calendar.daily_time_axis("1925-01-01", T)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    all_shards = plan_shards(ny_g, nx, 6, halo)
    shards = [all_shards[2], all_shards[3]]
    assert [(s.in0, s.in1, s.own0, s.own1) for s in shards] == [(238, 362, 240, 360), (358, 482, 360, 480)]
    es = EngineSet(0, 2)
    xs = [es.engines[0].synth_field(synth.make_tables(tm, s.ny_in, nx, lat_range=(s.in0, s.in1, s.ny_global)), cell_base=s.cell_base) for s in shards]
    torch.cuda.synchronize()
    r, local, mx = shard_step(es, shards, xs, cal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, nx=nx)
    torch.cuda.synchronize()
    assert r["path"] == "tails"
    summ = dict(zip(SUMMARY_KEYS, [int(v) for v in local.tolist()]))
    assert summ["thr_unresolved"] == 0 and summ["invalid_total"] == 0
    doy_idx = torch.from_numpy(cal.doy_out.astype(np.int64) - 1).to(hot.device)
    n_sum, ocean_sum = 0, 0
    for k, sh in enumerate(shards):
        C = sh.cells_in
        v = _views(es.workspaces[k], cal.T_out, C)
        own = sh.own_cell_slice()
        anom, ext, thr = v["anom"][:, own], v["ext"][:, own], v["thr"][:, own]
        n_true = 0
        for lo in range(0, cal.T_out, 2048):  # (1) the mask is anomaly >= threshold[dayofyear] on every owned cell-day
            hi = min(lo + 2048, cal.T_out)
            exp = anom[lo:hi] >= thr[doy_idx[lo:hi]]
            assert torch.equal(ext[lo:hi].bool(), exp), (k, lo)
            n_true += int(exp.sum().item())
        assert n_true == v["n_true"]
        ocean = v["mask"][own]
        assert torch.isnan(thr[:, ~ocean]).all() and bool(torch.isfinite(thr[:, ocean]).all())
        freq = n_true / (cal.T_out * int(ocean.sum().item()))
        assert abs(freq - 0.05) < 0.004, freq  # (2) the reference's frequency pin
        n_sum += n_true
        ocean_sum += int(ocean.sum().item())
    assert summ["n_extreme"] == n_sum and summ["n_ocean"] == ocean_sum  # (3) the step's counters are the bands' sums
    # (4) oracle parity around the first tile-row boundary of band 2: tile rows start at the first owned row (in-band row 2), so
    # in-band rows 31 | 32 belong to different tiles; columns 768..791 straddle the tile-column boundary at 780 = 30 x 26
    sh, v = shards[0], _views(es.workspaces[0], cal.T_out, shards[0].cells_in)
    for j0 in (29, 30):  # middle rows 31 and 32
        j1, i0, i1 = j0 + 5, 768, 792
        cells = (np.arange(j0, j1)[:, None] * nx + np.arange(i0, i1)[None, :]).reshape(-1)
        ct = torch.from_numpy(cells).to(hot.device)
        exp = orc.preprocess_arrays(xs[0][:, ct].cpu().numpy(), cal, ny=5, nx=i1 - i0, window_year_baseline=W,
                                    smooth_days_baseline=21, window_days_hobday=11, window_spatial_hobday=5,
                                    threshold_percentile=95.0, edges=bt.edges, centres=bt.centres)
        assert np.array_equal(v["anom"][:, ct].cpu().numpy(), exp["dat_anomaly"], equal_nan=True)
        w = i1 - i0
        inner = np.arange(2 * w + 2, 3 * w - 2)  # middle row, two columns in from the cut-out's edges (the oracle wraps there)
        it = ct[torch.from_numpy(inner).to(hot.device)]
        assert np.array_equal(v["thr"][:, it].cpu().numpy().T, exp["thresholds"][inner], equal_nan=True)
        assert np.array_equal(v["ext"][:, it].cpu().numpy().astype(bool), exp["extreme_events"][:, inner])
    del xs, es, v
    gc.collect()
    torch.cuda.empty_cache()
