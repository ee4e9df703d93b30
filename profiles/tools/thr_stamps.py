import sys, os
sys.path.insert(0, ".")
import numpy as np, torch
from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath
hot = HotPath(0)
ny, nx, T = 94, 1440, 36500
tm = calendar.daily_time_axis("1925-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=15)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
tab = synth.make_tables(tm, ny, nx, 20240607, lat_range=(88, 182, 720))
x = hot.synth_field(tab, cell_base=88*1440)
wsp = {}
for tile in (32,):
    with hot.ctx.options(THR_TILE=tile):
        for it in range(2):
            hot.ctx.debug_counters(reset=True)
            r = hot.shifting_hobday(x, dcal, W=15, S=21, bins=bt, q=0.95, wd=11, ws=5, ny=ny, nx=nx, own_rows=(2, 92), workspace=wsp)
            hot.sync()
            c = hot.ctx.debug_counters(reset=True)
    nblk = c[3] / 1.0
    print(f"tile {tile}: rebuilds {c[0]}, slow {c[1]}, passes {c[2]}, tile-days {c[3]}, mask-slow {c[4]}; per tile-day cycles: P1 {c[5]/c[3]:.0f}  P2 {c[6]/c[3]:.0f}  barrier-wait {c[7]/c[3]:.0f} (memtime ticks, 100 MHz => x24 for 2.4 GHz cycles?)")
