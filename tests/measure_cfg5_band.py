"""One-off measurement (not a pytest file): BASELINE.json configs[4] on one of 8 latitude bands -- 100-yr daily x 94x1440
(overlap rows included), detrend_fixed_baseline (orders 1,2 + harmonics) + hobday_extreme p90."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath

ny, nx, T = 94, 1440, 36500
hot = HotPath(0)
tm = calendar.daily_time_axis("1925-01-01", T)
cal = calendar.build_calendar(tm)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2], True)
x = hot.synth_field(synth.make_tables(tm, ny, nx, lat_range=(268, 362, 720)), cell_base=268 * nx)
ws = {}


def step():
    d = hot.detrend(x, model, pmodel, True, None, count_invalid=True, wsp=ws)
    r = hot.fixed_baseline(d["out"], dcal, None, bt, count_invalid=False, wsp=ws)
    t = hot.hobday_thresholds(r["bins"], r["out"], dcal, bt, 0.90, 11, 5, ny, nx, rows=(2, 92), wsp=ws)
    m = hot.mask_ge_doy(r["out"], t["thr_doy_major"], dcal, cells=(2 * nx, 92 * nx), wsp=ws, binned=(r["bins"], bt))
    return m


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
kern = {k: hot.ctx.timing_get(k) for k in ("detrend", "fixed", "thresholds", "mask")}
C = ny * nx
print({"ms_per_band": round(dt * 1e3, 2), "Mcells_ts_per_s_8bands_1gpu": round(T * 720 * 1440 / 1e6 / (8 * dt)),
       "roofline_frac": round(90 * 1440 * (4 * T + 5 * T + 1465) / dt / 8e12, 4),
       "kernel_ms": {k: round(v[0] / max(v[1], 1), 2) for k, v in kern.items()}, "n_extreme": int(m["n_true"].item())})
