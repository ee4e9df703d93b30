"""Measurement (not a test): how the threshold kernel's speculative band copes with thresholds that drift with the season.
The synthetic benchmark field has stationary noise; here extra noise whose amplitude follows the day of the year is added, so
the p95 threshold swings by roughly +-20 % over a year.  Prints the kernel time per day-block length (MAREX_THR_DD)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marex_amd import binning, calendar, synth  # noqa: E402
from marex_amd.engine import HotPath  # noqa: E402

ny, nx, T = 94, 1440, 36500
hot = HotPath(0)
tm = calendar.daily_time_axis("1925-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=15)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
x = hot.synth_field(synth.make_tables(tm, ny, nx, lat_range=(268, 362, 720)), cell_base=268 * nx)
amp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
if amp > 0:
    g = torch.from_numpy((0.5 * (1 + np.sin(2 * np.pi * cal.doy / 365.25))).astype(np.float32)).to(hot.device)
    gen = torch.Generator(device=hot.device).manual_seed(1)
    for t0 in range(0, T, 2000):  # in slabs: no second field-sized temporary
        t1 = min(T, t0 + 2000)
        x[t0:t1] += amp * g[t0:t1, None] * torch.randn((t1 - t0, ny * nx), device=hot.device, generator=gen)
ws = {}
for dd in os.environ.get("DDS", "0,32,48,61,0").split(","):
    hot.ctx.set_option("THR_DD", int(dd))  # options are read from the context, not from the environment at launch
    for k in range(3):
        if k == 1:
            hot.sync()
            hot.ctx.timing_enable(True)
            hot.ctx.timing_reset()
        r = hot.shifting_hobday(x, dcal, W=15, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=ny, nx=nx, own_rows=(2, 92), workspace=ws)
    hot.sync()
    ms, n = hot.ctx.timing_get("thresholds")
    thr = r["thresholds"][2 * nx: 92 * nx]
    ocean = torch.isfinite(thr[:, 0])
    swing = (thr[ocean].max(dim=1).values - thr[ocean].min(dim=1).values).median().item()
    print(f"amp={amp} MAREX_THR_DD={dd or 'auto'}: thresholds kernel {ms / max(n, 1):.2f} ms, median seasonal swing of the "
          f"threshold {swing:.2f} K, n_extreme {int(r['n_true'].item())}", flush=True)
    hot.ctx.timing_enable(False)
