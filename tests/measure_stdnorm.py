"""One-off measurement (not a pytest file): detrend_harmonic (orders 1,2 + harmonics) with std_normalise at the size of cfg2."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath

ny, nx, T = 720, 1440, 3652
hot = HotPath(0)
tm = calendar.daily_time_axis("2015-01-01", T)
cal = calendar.build_calendar(tm)
dcal = hot.upload_calendar(cal)
model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2], True)
x = hot.synth_field(synth.make_tables(tm, ny, nx))
ws = {}


def step():
    d = hot.detrend(x, model, pmodel, True, None, count_invalid=True, wsp=ws)
    return hot.std_normalise(d["out"], dcal, wsp=ws)


for _ in range(2):
    r = step()
hot.sync()
hot.ctx.timing_enable(True)
hot.ctx.timing_reset()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    r = step()
hot.sync()
dt = (time.perf_counter() - t0) / K
kern = {k: hot.ctx.timing_get(k) for k in ("detrend", "stdnorm")}
print({"ms": round(dt * 1e3, 2), "kernel_ms_per_launch": {k: round(v[0] / max(v[1], 1), 2) for k, v in kern.items()},
       "launches": {k: v[1] // K for k, v in kern.items()}})
