"""GPU: the BASELINE.json configurations that had no test of their own at their own sizes.

* configs[4] (cfg5): ``detrend_fixed_baseline`` (orders 1, 2) + ``hobday_extreme`` p90 on the 100-yr axis (T = 36 500) --
  bit parity with the oracle on a 20x24 grid AND the fp32-vs-fp64 sweep the configuration is named for: the float32
  residual of the device against an order-independent float64 least-squares residual (detect.py:2143-2224, 2400-2462),
  tolerance ``1e-5 * max|x|`` (BASELINE.json: "fp32 anomalies within 1e-5 relative"), the maximum is printed.
* one cfg5 band at full size (100-yr daily x 94x1440): size-independent properties + parity on a cut-out.
* one configs[3] (cfg4) share at full size (30-yr daily x 500 000 unstructured cells, no pooling): mask == anomaly >=
  threshold[dayofyear] on the device, the kernel's count, the reference's frequency pin, oracle parity on a 400-cell cut-out.
"""
import numpy as np
import pytest
import torch

from marex_amd import binning, calendar, synth
from marex_amd.dist import plan_shards, shard_step
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def _lsq_residual64(x, model, pmodel, fzm=True):
    """Order-free reference: float64 least squares, residual, optional zero mean (detect.py:2206-2224 in float64)."""
    x64 = x.astype(np.float64)
    res = x64 - model.T @ (pmodel.T @ x64)
    return res - res.mean(axis=0) if fzm else res


@pytest.mark.parametrize("orders", [(1,), (1, 2)])
def test_cfg5_hundred_year_axis_p90_bit_parity_and_fp64_sweep(hot, orders):
    ny, nx, T = 20, 24, 36500
    tm = calendar.daily_time_axis("1925-01-01", T)
    cal = calendar.build_calendar(tm)
    bt = binning.hobday_bins()
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), list(orders), False)
    x = synth.synth_field(synth.make_tables(tm, ny, nx))
    exp = orc.preprocess_arrays(x, cal, ny=ny, nx=nx, method_anomaly="detrend_fixed_baseline", window_days_hobday=11,
                                window_spatial_hobday=5, threshold_percentile=90.0, edges=bt.edges, centres=bt.centres,
                                model=model, pmodel=pmodel, force_zero_mean=True)
    dcal = hot.upload_calendar(cal)
    xd = torch.from_numpy(x).to(hot.device)
    sh = plan_shards(ny, nx, 1, 2)[0]
    r, local, _ = shard_step(hot, [sh], [xd], dcal, bins=bt, q=0.90, wd=11, ws=5, nx=nx, workspace={}, detrend=(model, pmodel))
    hot.sync()
    assert cal.T_out == T
    assert np.array_equal(r["dat_anomaly"].cpu().numpy(), exp["dat_anomaly"], equal_nan=True)
    assert np.array_equal(r["thr_doy_major"].cpu().numpy().T, exp["thresholds"], equal_nan=True)
    assert np.array_equal(r["extreme_events"].cpu().numpy().astype(bool), exp["extreme_events"])
    ocean = exp["mask"]
    freq = exp["extreme_events"][:, ocean].mean()
    assert abs(freq - 0.10) < 0.01 and int(local[3]) == int(exp["extreme_events"].sum())
    # the sweep: float32 device residual of the trend fit (the stage that differs between precisions) vs float64 least squares
    d = hot.detrend(xd, model, pmodel, True)
    hot.sync()
    res32 = d["out"].cpu().numpy()[:, ocean]
    res64 = _lsq_residual64(x[:, ocean], model, pmodel)
    scale = float(np.abs(x[:, ocean]).max())
    err = float(np.abs(res32 - res64).max())
    print(f"\n[cfg5 sweep, orders {orders}, T={T}] max|fp32 - fp64| = {err:.3e} = {err / scale:.3e} of max|x| = {scale:.2f}")
    assert err <= 1e-5 * scale
    assert float(np.abs(res32.mean(axis=0)).max()) < 1e-5
    # and through the one-chain kernel: anomaly = residual - daily climatology, against the same in float64
    clim64 = np.zeros((366, res64.shape[1]))
    for dd in range(366):
        rows = cal.doy == dd + 1
        if rows.any():
            clim64[dd] = res64[rows].mean(axis=0)
    anom64 = res64 - clim64[cal.doy.astype(np.int64) - 1]
    err_a = float(np.abs(r["dat_anomaly"].cpu().numpy()[:, ocean] - anom64).max())
    print(f"[cfg5 sweep, orders {orders}] anomaly after the climatology: max|fp32 - fp64| = {err_a:.3e} = {err_a / scale:.3e} of max|x|")
    assert err_a <= 1e-5 * scale


def test_cfg5_band_full_size_properties_and_block_parity(hot):
    """One of the 8 latitude bands of cfg5 at full size (100-yr daily x 94x1440, detrend orders (1, 2), p90)."""
    ny, nx, T = 94, 1440, 36500
    tm = calendar.daily_time_axis("1925-01-01", T)
    cal = calendar.build_calendar(tm)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2], False)
    x = hot.synth_field(synth.make_tables(tm, ny, nx, lat_range=(268, 362, 720)), cell_base=268 * nx)
    sh = plan_shards(720, nx, 8, 2)[3]
    assert (sh.in0, sh.in1) == (268, 362)
    wsp = {}
    r, local, mx = shard_step(hot, [sh], [x], dcal, bins=bt, q=0.90, wd=11, ws=5, nx=nx, workspace=wsp, detrend=(model, pmodel))
    hot.sync()
    own = sh.own_cell_slice()
    anom, ext, thr, mask = r["dat_anomaly"], r["extreme_events"], r["thr_doy_major"], r["mask"].bool()
    doy_idx = torch.from_numpy(cal.doy_out.astype(np.int64) - 1).to(hot.device)
    n_true = 0
    for lo in range(0, cal.T_out, 2048):
        hi = min(lo + 2048, cal.T_out)
        e = anom[lo:hi, own] >= thr[doy_idx[lo:hi]][:, own]
        assert torch.equal(ext[lo:hi, own].bool(), e)
        n_true += int(e.sum().item())
    assert n_true == int(r["n_true"].item()) == int(local[3])
    n_ocean = int(mask[own].sum().item())
    assert n_ocean == int(local[0]) and int(mx) == 0
    freq = n_true / (cal.T_out * n_ocean)
    assert abs(freq - 0.10) < 0.005, freq
    assert bool(torch.isfinite(anom[:, mask]).all()) and bool(torch.isnan(anom[:, ~mask]).all())
    assert float(anom[:, mask].mean(dim=0).abs().max()) < 1e-4  # force_zero_mean + a climatology of its own
    j0, j1, i0, i1 = 40, 45, 900, 924
    cells = (np.arange(j0, j1)[:, None] * nx + np.arange(i0, i1)[None, :]).reshape(-1)
    ct = torch.from_numpy(cells).to(hot.device)
    exp = orc.preprocess_arrays(x[:, ct].cpu().numpy(), cal, ny=j1 - j0, nx=i1 - i0, method_anomaly="detrend_fixed_baseline",
                                window_days_hobday=11, window_spatial_hobday=5, threshold_percentile=90.0, edges=bt.edges,
                                centres=bt.centres, model=model, pmodel=pmodel, force_zero_mean=True)
    assert np.array_equal(anom[:, ct].cpu().numpy(), exp["dat_anomaly"], equal_nan=True)
    w = i1 - i0
    inner = np.arange(2 * w + 2, 3 * w - 2)
    it = ct[torch.from_numpy(inner).to(hot.device)]
    assert np.array_equal(thr[:, it].cpu().numpy().T, exp["thresholds"][inner], equal_nan=True)
    assert np.array_equal(ext[:, it].cpu().numpy().astype(bool), exp["extreme_events"][:, inner])


def test_cfg4_share_full_size_properties_and_cell_parity(hot):
    """One GPU's share of configs[3]: 30-yr daily x 500 000 cells of the 2e6-cell unstructured mesh, no spatial pooling."""
    C, Ctot, T, W = 500_000, 2_000_000, 10957, 15
    tm = calendar.daily_time_axis("1995-01-01", T)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    sh = plan_shards(0, Ctot, 4, 0)[1]
    assert sh.cells_in == C and sh.cell_base == C
    x = hot.synth_field(synth.make_tables(tm, 0, C, unstructured=True), cell_base=sh.cell_base)
    r, local, mx = shard_step(hot, [sh], [x], dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=1, nx=Ctot, workspace={})
    hot.sync()
    assert r["path"] == "tails" and cal.T_out == 5478
    anom, ext, thr, mask = r["dat_anomaly"], r["extreme_events"], r["thr_doy_major"], r["mask"].bool()
    doy_idx = torch.from_numpy(cal.doy_out.astype(np.int64) - 1).to(hot.device)
    n_true = 0
    for lo in range(0, cal.T_out, 512):
        hi = min(lo + 512, cal.T_out)
        e = anom[lo:hi] >= thr[doy_idx[lo:hi]]
        assert torch.equal(ext[lo:hi].bool(), e)
        n_true += int(e.sum().item())
    assert n_true == int(r["n_true"].item()) == int(local[3])
    n_ocean = int(mask.sum().item())
    assert n_ocean == int(local[0]) and int(mx) == 0 and 0.6 < n_ocean / C < 0.8
    freq = n_true / (cal.T_out * n_ocean)
    assert abs(freq - 0.05) < 0.004, freq
    assert bool(torch.isnan(anom[:, ~mask]).all()) and bool(torch.isnan(thr[:, ~mask]).all())
    assert bool((thr[:, mask] >= float(bt.lower_bound)).all())
    # oracle parity on 400 cells cut out of the share (cells are independent without pooling: any subset will do)
    cells = np.concatenate([np.arange(0, 100), np.arange(249_950, 250_150), np.arange(C - 100, C)])
    ct = torch.from_numpy(cells).to(hot.device)
    exp = orc.preprocess_arrays(x[:, ct].cpu().numpy(), cal, ny=0, nx=len(cells), window_year_baseline=W, smooth_days_baseline=21,
                                window_days_hobday=11, window_spatial_hobday=None, threshold_percentile=95.0, edges=bt.edges,
                                centres=bt.centres)
    assert np.array_equal(anom[:, ct].cpu().numpy(), exp["dat_anomaly"], equal_nan=True)
    assert np.array_equal(thr[:, ct].cpu().numpy().T, exp["thresholds"], equal_nan=True)
    assert np.array_equal(ext[:, ct].cpu().numpy().astype(bool), exp["extreme_events"])
