"""One-off measurement (not a pytest file): BASELINE.json configs[3] on ONE GPU's share -- 30-yr daily x 500 000 cells of an
unstructured mesh (1/4 of the 2e6-cell ICON-O grid), shifting_baseline(W=15, S=21) + hobday_extreme p95, no pooling."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath

C = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
T, W = 10957, 15
hot = HotPath(0)
tm = calendar.daily_time_axis("1995-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=W)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
x = hot.synth_field(synth.make_tables(tm, 0, C, unstructured=True))
ws = {}
for _ in range(2):
    r = hot.shifting_hobday(x, dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=1, ny=0, nx=C, workspace=ws)
hot.sync()
hot.ctx.timing_enable(True)
hot.ctx.timing_reset()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    r = hot.shifting_hobday(x, dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=1, ny=0, nx=C, workspace=ws)
hot.sync()
dt = (time.perf_counter() - t0) / K
kern = {k: hot.ctx.timing_get(k) for k in ("shifting", "thresholds", "mask", "transpose")}
alg = C * (4 * T + 5 * cal.T_out + 1465)
print({"ms_per_pass": round(dt * 1e3, 2), "Mcells_ts_per_s": round(T * C / 1e6 / dt), "roofline_frac": round(alg / dt / 8e12, 4),
       "kernel_ms": {k: round(v[0] / v[1], 2) for k, v in kern.items()}, "n_extreme": int(r["n_true"].item())})
