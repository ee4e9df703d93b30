"""Per-cell threshold kernel on the cfg4 shape (0.5 M unstructured cells, 30 years): kernel ms and the exact-recount counter."""
import sys
sys.path.insert(0, ".")
import torch
from marex_amd import binning, calendar, synth
from marex_amd.engine import HotPath
hot = HotPath(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10957
tm = calendar.daily_time_axis("1990-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=15)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
x = hot.synth_field(synth.make_tables(tm, 0, C, unstructured=True))
wsp = {}
a = hot.shifting_baseline_tails(x, dcal, 15, 21, bt, wsp=wsp)
tl = a["tails"]
print("max_bucket", tl["max_bucket"], "list_rows", tl["list_rows"], "lists", tuple(tl["tails"].shape))
for opts in [{}, {"THR_CELLS_CACHE": 0}, {"THR_CELLS": 0}]:
    with hot.ctx.options(**opts):
        for it in range(3):
            if it == 1:
                hot.sync(); hot.ctx.timing_enable(True); hot.ctx.timing_reset(); hot.ctx.debug_counters(reset=True)
            t = hot.hobday_thresholds_tails(tl, a["out"], dcal, bt, 0.95, 11, 1, 0, C, wsp=wsp)
        hot.sync()
        ms, n = hot.ctx.timing_get("thresholds")
        print(opts, f"{ms / n:.2f} ms", "dbg", hot.ctx.debug_counters(reset=True)[:4])
        hot.ctx.timing_enable(False)
