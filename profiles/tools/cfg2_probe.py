"""Band statistics of the tile threshold kernel on the cfg2 shape (10-year record, 5 samples per bucket, 5x5 pooling)."""
import sys
sys.path.insert(0, ".")
from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath
hot = HotPath(0)
ny, nx, T = 180, 1440, 3652
tm = calendar.daily_time_axis("2010-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=5)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
x = hot.synth_field(synth.make_tables(tm, ny, nx, lat_range=(270, 450, 720)), cell_base=270 * nx)
for path in ("tails", "bins"):
    hot.hobday_path = path
    wsp = {}
    for it in range(3):
        if it == 1:
            hot.sync(); hot.ctx.timing_enable(True); hot.ctx.timing_reset(); hot.ctx.debug_counters(reset=True)
        r = hot.shifting_hobday(x, dcal, W=5, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=ny, nx=nx, workspace=wsp)
    hot.sync()
    ms, n = hot.ctx.timing_get("thresholds")
    c = hot.ctx.debug_counters(reset=True)
    thr = r["thr_doy_major"]
    import torch
    ok = torch.isfinite(thr[0])
    t = thr[:, ok]
    print(path, f"thresholds {ms / n:.2f} ms; rebuilds {c[0] // 2}, passes {c[2] // 2}, tile-days {c[3] // 2};",
          f"threshold spread over the field on day 0: p1 {t[0].quantile(0.01).item():.3f} p99 {t[0].quantile(0.99).item():.3f}")
    hot.ctx.timing_enable(False)
