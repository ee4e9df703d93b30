"""One-off measurement (not a pytest file): shifting_baseline + hobday_extreme with method_percentile="exact" at the size
of cfg2 (10-yr daily x 720x1440, W = 5): np.nanpercentile per (cell, dayofyear window), no pooling."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath

ny, nx, T, W = 720, 1440, 3652, 5
hot = HotPath(0)
tm = calendar.daily_time_axis("2015-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=W)
dcal = hot.upload_calendar(cal)
x = hot.synth_field(synth.make_tables(tm, ny, nx))
ws = {}
r = hot.shifting_baseline(x, dcal, W, 21, None, wsp=ws)
cal_out = calendar.build_calendar(tm[cal.kept])
dco = hot.upload_calendar(cal_out)


def step():
    thr = hot.hobday_thresholds_exact(r["out"], dco, 95.0, 11, wsp=ws)
    return hot.mask_ge_doy(r["out"], thr, dco, wsp=ws)


for _ in range(2):
    m = step()
hot.sync()
hot.ctx.timing_enable(True)
hot.ctx.timing_reset()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    m = step()
hot.sync()
dt = (time.perf_counter() - t0) / K
kern = {k: hot.ctx.timing_get(k) for k in ("exact", "mask")}
print({"ms_thresholds_plus_mask": round(dt * 1e3, 2), "kernel_ms": {k: round(v[0] / max(v[1], 1), 2) for k, v in kern.items()},
       "n_extreme": int(m["n_true"].item())})
