import sys; sys.path.insert(0,'.')
import numpy as np, torch, collections
from marex_amd import binning, calendar, synth
from marex_amd.detect import get_engine
hot = get_engine(0)
start, periods, ny, nx, W = "2001-01-01", 11*365+3, 20, 37, 3
tm = calendar.daily_time_axis(start, periods)
x = synth.synth_field(synth.make_tables(tm, ny, nx))
cal = calendar.build_calendar(tm, window_year_baseline=W)
dcal = hot.upload_calendar(cal)
bt = binning.hobday_bins()
xd = torch.from_numpy(x).to(hot.device)
for trial in range(2):
    a = hot.shifting_baseline_tails(xd, dcal, W, 21, bt)
    hot.sync()
    tt = a["tails"]["tails"].cpu().numpy().view(np.uint16).copy()
    ref = hot.tail_extract(a["out"], dcal, bt); hot.sync()
    tr = ref["tails"].cpu().numpy().view(np.uint16)
    diff = (tt != tr).any(axis=(1, 2, 4))   # [366, C]
    dd, cc = np.nonzero(diff)
    print("trial", trial, "bad buckets", len(dd), "of", diff.size)
    print(" by d%4", collections.Counter((dd % 4).tolist()))
    print(" by c%64 (top)", collections.Counter((cc % 64).tolist()).most_common(8))
    print(" by c//64", collections.Counter((cc // 64).tolist()).most_common(12))
    print(" by chunk d//4 (top)", collections.Counter((dd // 4).tolist()).most_common(8))
    for d, c in list(zip(dd, cc))[:3]:
        print("  d", d, "c", c, "shift", tt[d, 0, :, c, :].reshape(-1).tolist(), "extract", tr[d, 0, :, c, :].reshape(-1).tolist())
