"""One-off measurement (not a pytest file): fixed_baseline + global_extreme (approximate and exact) at the size of cfg2
(10-yr daily x 720x1440) -- the method pair of BASELINE.json configs[0], scaled up."""
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
bt = binning.hobday_bins()
x = hot.synth_field(synth.make_tables(tm, ny, nx))
ws = {}
for mp in ("approximate", "exact"):
    def step():
        r = hot.fixed_baseline(x, dcal, None, None, count_invalid=True, wsp=ws)
        g = hot.global_threshold(r["out"], 95.0, mp, bt)
        return hot.mask_ge_const(r["out"], g["thr_f64"], wsp=ws)
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
    kern = {k: hot.ctx.timing_get(k) for k in ("fixed", "global", "mask")}
    print(mp, {"ms_per_pass": round(dt * 1e3, 2), "Mcells_ts_per_s": round(T * ny * nx / 1e6 / dt),
               "kernel_ms": {k: round(v[0] / max(v[1], 1), 2) for k, v in kern.items()}, "n_extreme": int(m["n_true"].item())})
