"""Phase timers of the anomaly kernel (-DSHIFT_STAMPS build: MAREX_LIB_PATH=marex_amd/csrc/alt/libmarex_hip_sstamps.so).
Per wave and calendar year, in s_memtime ticks (100 MHz): top = plan loads + row prefetch issue + staged rows into registers
+ LDS wait + barrier; mid = smoothing, climatology, anomaly stores, keys; end = wait for the prefetched rows + stage write +
barrier."""
import sys
sys.path.insert(0, ".")
from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath
hot = HotPath(0)
ny, nx, T = 94, 1440, 36500
tm = calendar.daily_time_axis("1925-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=15)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
tab = synth.make_tables(tm, ny, nx, 20240607, lat_range=(88, 182, 720))
x = hot.synth_field(tab, cell_base=88 * 1440)
wsp = {}
for it in range(2):
    hot.ctx.debug_counters(reset=True)
    tl = hot.shifting_baseline_tails(x, dcal, 15, 21, bt, wsp=wsp)
    hot.sync()
    c = hot.ctx.debug_counters(reset=True)
waves, years = c[4], 100
print(f"waves {waves}; ticks per wave-year: top {c[5] / waves / years:.1f}  mid {c[6] / waves / years:.1f}  end {c[7] / waves / years:.1f}"
      f"  (x24 = 2.4 GHz cycles: {24 * c[5] / waves / years:.0f} / {24 * c[6] / waves / years:.0f} / {24 * c[7] / waves / years:.0f})")
