"""GPU: the BASELINE.json sizes themselves.

* cfg2 at FULL size (10-yr daily x 720x1440, 15 GB resident): size-independent properties on the whole field computed on
  the device, plus bit parity with the oracle on a 5 x 48 block cut out of the middle of the grid.
* the 100-yr time axis of cfg3 (36 500 days, W = 15, 85-sample dayofyear buckets) on a 20x24 grid: full bit parity.
"""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar, synth
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def test_cfg2_full_size_properties_and_row_parity(hot):
    ny, nx, T, W = 720, 1440, 3652, 5
    tm = calendar.daily_time_axis("2015-01-01", T)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    x = hot.synth_field(synth.make_tables(tm, ny, nx))
    r = hot.shifting_hobday(x, dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=ny, nx=nx)
    hot.sync()
    anom, ext, thr, mask = r["dat_anomaly"], r["extreme_events"], r["thr_doy_major"], r["mask"].bool()
    assert anom.shape == (cal.T_out, ny * nx) and thr.shape == (366, ny * nx)
    # (1) the mask is exactly anomaly >= threshold[dayofyear] over all 1.9e9 cell-days, and the kernel's count agrees
    doy_idx = torch.from_numpy(cal.doy_out.astype(np.int64) - 1).to(hot.device)
    n_true = 0
    for lo in range(0, cal.T_out, 256):  # row blocks keep the gathered threshold rows small
        hi = min(lo + 256, cal.T_out)
        exp = anom[lo:hi] >= thr[doy_idx[lo:hi]]
        assert torch.equal(ext[lo:hi].bool(), exp)
        n_true += int(exp.sum().item())
    assert n_true == int(r["n_true"].item())
    # (2) land / ocean handling and the reference's frequency pin (5 % +- 1 %, here tighter on 1.4e9 ocean cell-days)
    assert torch.isnan(anom[:, ~mask]).all() and torch.isnan(thr[:, ~mask]).all()
    assert bool(torch.isfinite(anom[:, mask]).all())
    freq = n_true / (cal.T_out * int(mask.sum().item()))
    assert abs(freq - 0.05) < 0.004, freq
    assert int(r["invalid_count"][mask].sum().item()) == 0
    # (3) bit parity with the oracle on a 5 x 48 block cut out of the middle of the field (rows 358..362, columns
    #     700..747): anomalies for every cell of the block; thresholds / extremes for the interior of its middle row,
    #     whose 5x5 pooling neighbourhood lies inside the block (the oracle wraps the block's own edges instead)
    j0, j1, i0, i1 = 358, 363, 700, 748
    cells = (np.arange(j0, j1)[:, None] * nx + np.arange(i0, i1)[None, :]).reshape(-1)
    ct = torch.from_numpy(cells).to(hot.device)
    xb = x[:, ct].cpu().numpy()
    exp = orc.preprocess_arrays(xb, cal, ny=j1 - j0, nx=i1 - i0, window_year_baseline=W, smooth_days_baseline=21,
                                window_days_hobday=11, window_spatial_hobday=5, threshold_percentile=95.0,
                                edges=bt.edges, centres=bt.centres)
    assert np.array_equal(anom[:, ct].cpu().numpy(), exp["dat_anomaly"], equal_nan=True)
    w = i1 - i0
    inner = np.arange(2 * w + 2, 3 * w - 2)  # middle row, two columns in from the block's edges
    it = ct[torch.from_numpy(inner).to(hot.device)]
    assert np.array_equal(thr[:, it].cpu().numpy().T, exp["thresholds"][inner], equal_nan=True)
    assert np.array_equal(ext[:, it].cpu().numpy().astype(bool), exp["extreme_events"][:, inner])


def test_hundred_year_axis_small_grid_bit_exact(hot):
    ny, nx, T, W = 20, 24, 36500, 15
    tm = calendar.daily_time_axis("1925-01-01", T)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    x = synth.synth_field(synth.make_tables(tm, ny, nx))
    exp = orc.preprocess_arrays(x, cal, ny=ny, nx=nx, window_year_baseline=W, smooth_days_baseline=21,
                                window_days_hobday=11, window_spatial_hobday=5, threshold_percentile=95.0,
                                edges=bt.edges, centres=bt.centres)
    r = hot.shifting_hobday(torch.from_numpy(x).to(hot.device), hot.upload_calendar(cal), W=W, S=21, bins=bt, q=0.95,
                            wd=11, ws=5, ny=ny, nx=nx)
    hot.sync()
    assert cal.T_out == 31022
    assert np.array_equal(r["dat_anomaly"].cpu().numpy(), exp["dat_anomaly"], equal_nan=True)
    assert np.array_equal(r["thresholds"].cpu().numpy(), exp["thresholds"], equal_nan=True)
    assert np.array_equal(r["extreme_events"].cpu().numpy().astype(bool), exp["extreme_events"])
    ocean = exp["mask"]
    assert abs(exp["extreme_events"][:, ocean].mean() - 0.05) < 0.01


def test_cfg3_band_full_size_properties_and_block_parity(hot):
    """One of the 8 latitude bands of cfg3 at full size (100-yr daily x 94x1440 incl. overlap rows, 19.8 GB resident)."""
    ny, nx, T, W = 94, 1440, 36500, 15
    tm = calendar.daily_time_axis("1925-01-01", T)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    x = hot.synth_field(synth.make_tables(tm, ny, nx, lat_range=(268, 362, 720)), cell_base=268 * nx)
    r = hot.shifting_hobday(x, dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=ny, nx=nx)
    hot.sync()
    anom, ext, thr, mask = r["dat_anomaly"], r["extreme_events"], r["thr_doy_major"], r["mask"].bool()
    doy_idx = torch.from_numpy(cal.doy_out.astype(np.int64) - 1).to(hot.device)
    n_true = 0
    for lo in range(0, cal.T_out, 2048):
        hi = min(lo + 2048, cal.T_out)
        exp = anom[lo:hi] >= thr[doy_idx[lo:hi]]
        assert torch.equal(ext[lo:hi].bool(), exp)
        n_true += int(exp.sum().item())
    assert n_true == int(r["n_true"].item())
    freq = n_true / (cal.T_out * int(mask.sum().item()))
    assert abs(freq - 0.05) < 0.004, freq
    j0, j1, i0, i1 = 40, 45, 900, 924
    cells = (np.arange(j0, j1)[:, None] * nx + np.arange(i0, i1)[None, :]).reshape(-1)
    ct = torch.from_numpy(cells).to(hot.device)
    exp = orc.preprocess_arrays(x[:, ct].cpu().numpy(), cal, ny=j1 - j0, nx=i1 - i0, window_year_baseline=W,
                                smooth_days_baseline=21, window_days_hobday=11, window_spatial_hobday=5,
                                threshold_percentile=95.0, edges=bt.edges, centres=bt.centres)
    assert np.array_equal(anom[:, ct].cpu().numpy(), exp["dat_anomaly"], equal_nan=True)
    w = i1 - i0
    inner = np.arange(2 * w + 2, 3 * w - 2)
    it = ct[torch.from_numpy(inner).to(hot.device)]
    assert np.array_equal(thr[:, it].cpu().numpy().T, exp["thresholds"][inner], equal_nan=True)
    assert np.array_equal(ext[:, it].cpu().numpy().astype(bool), exp["extreme_events"][:, inner])
