import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath
hot = HotPath(0)
for (start, T, W, ny, nx) in (("1925-01-01", 36500, 15, 8, 64), ("2015-01-01", 3652, 5, 8, 64)):
    tm = calendar.daily_time_axis(start, T)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    x = hot.synth_field(synth.make_tables(tm, ny, nx))
    hot.ctx.debug_counters(reset=True)
    a = hot.shifting_baseline_tails(x, dcal, W, 21, bt)
    hot.sync()
    print(start, T, W, "debug counters", hot.ctx.debug_counters(reset=True))
