"""Which years take the straight-line body of k_shift_lean (debug counters 6 / 7 = wave-years lean / general) and what the
anomaly stage costs with either kernel, for a few (series length, W, cells) shapes.  Run on the GPU box."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath
hot = HotPath(0)
bt = binning.hobday_bins()
for (start, T, W, ny, nx) in (("1925-01-01", 36500, 15, 8, 1440), ("2015-01-01", 3652, 5, 94, 1440), ("2015-01-01", 3652, 5, 720, 1440),
                              ("1995-01-01", 10957, 5, 94, 1440), ("1995-01-01", 10957, 15, 94, 1440)):
    tm = calendar.daily_time_axis(start, T)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    x = hot.synth_field(synth.make_tables(tm, ny, nx))
    wsp = {}
    for lean in (1, 0):
        with hot.ctx.options(SHIFT_LEAN=lean):
            hot.ctx.debug_counters(reset=True)
            hot.shifting_baseline_tails(x, dcal, W, 21, bt, wsp=wsp)
            hot.sync()
            cnt = hot.ctx.debug_counters(reset=True)
            t0 = time.perf_counter()
            for _ in range(3):
                hot.shifting_baseline_tails(x, dcal, W, 21, bt, wsp=wsp)
            hot.sync()
            ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"T={T} W={W} C={ny*nx} lean={lean}: {ms:.2f} ms, wave-years lean/general {cnt[6]}/{cnt[7]}", flush=True)
    del x, wsp
    torch.cuda.empty_cache()
