"""Small driver for rocprofv3 runs: one pass of the hot path on a reduced grid (not a pytest file)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath

ny, nx = int(sys.argv[1]) if len(sys.argv) > 1 else 180, int(sys.argv[2]) if len(sys.argv) > 2 else 1440
T, W = 3652, 5
hot = HotPath(0)
tm = calendar.daily_time_axis("2015-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=W)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
tab = synth.make_tables(tm, ny, nx)
x = hot.synth_field(tab)
for _ in range(2):
    r = hot.shifting_hobday(x, dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=ny, nx=nx)
hot.sync()
print("done", int(r["n_true"].item()))
