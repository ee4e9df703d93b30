"""One-off measurement (not a pytest file): tracker pre-processing (fill_holes R=8, fill_time_gaps T_fill=2) on the
extreme mask of cfg2 (1826 x 720 x 1440), produced on the device by the hot path itself."""
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
bt = binning.hobday_bins()
x = hot.synth_field(synth.make_tables(tm, ny, nx))
r = hot.shifting_hobday(x, dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=ny, nx=nx)
ext, mask = r["extreme_events"], r["mask"]
del x
ws = {}


def step():
    a = hot.fill_holes(ext, mask, ny, nx, 8, False, wsp=ws)
    g = hot.fill_time_gaps(a, mask, ny, nx, 8, 2, False, wsp=ws)
    if os.environ.get("MAREX_MEASURE_FILTER", "1") == "1":
        return hot.filter_small_objects(g, ny, nx, 0.5, None, False, wsp=ws)["filtered"]
    return g


for _ in range(2):
    b = step()
hot.sync()
hot.ctx.timing_enable(True)
hot.ctx.timing_reset()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    b = step()
hot.sync()
dt = (time.perf_counter() - t0) / K
k = hot.ctx.timing_get("morph")
n = ext.numel()
print({"ms": round(dt * 1e3, 2), "launch_groups": k[1] // K, "Gcells_per_s": round(n / dt / 1e9, 1),
       "in_true_frac": round(float(ext.float().mean().item()), 4), "out_true_frac": round(float(b.float().mean().item()), 4)})
