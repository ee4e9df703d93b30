"""MI355X: do the fixed-baseline kernels pay for emitting the key lists themselves?  (round 4)
    python tests/measure_fused_tails.py
40-yr and 100-yr daily fields, fixed_baseline with / without `tails_bins`, + the extraction pass the fused form replaces."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from marex_amd import binning, calendar  # noqa: E402
from marex_amd.engine import HotPath  # noqa: E402

hot = HotPath(0)
bt = binning.hobday_bins()
out = {}
for years, C in ((40, 200 * 1440), (100, 94 * 1440)):
    tm = calendar.daily_time_axis("1925-01-01", years * 365 + years // 4)
    cal = calendar.build_calendar(tm)
    dcal = hot.upload_calendar(cal)
    x = torch.randn((len(tm), C), device=hot.device, dtype=torch.float32)
    wsp = {}
    res = {}
    for name, kw in (("plain", {}), ("fused", {"tails_bins": bt})):
        hot.ctx.timing_enable(True)
        for k in range(4):
            if k == 1:
                hot.sync()
                hot.ctx.timing_reset()
            a = hot.fixed_baseline(x, dcal, None, None, wsp=wsp, **kw)
            if name == "plain":
                hot.tail_extract(a["out"], dcal, bt, wsp=wsp)
        hot.sync()
        res[name] = {k: hot.ctx.timing_get(k)[0] / max(hot.ctx.timing_get(k)[1], 1) for k in ("fixed", "tails")}
    out[f"{years}yr x {C} cells"] = res
    del x, wsp
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
