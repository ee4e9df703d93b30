import sys; sys.path.insert(0,'.')
import numpy as np, torch
from marex_amd import binning, calendar, synth
from marex_amd.detect import get_engine
hot = get_engine(0)
def keys_of(tl, C):
    t = tl["tails"].cpu().numpy().view(np.uint16)
    nper = t.shape[1]
    per_list = np.ascontiguousarray(t.transpose(0, 1, 2, 4, 3)).reshape(366, nper, 16, C)
    srt = bool((np.diff(per_list.astype(np.int32), axis=2) <= 0).all())
    allk = per_list.reshape(366, nper * 16, C)
    return -np.sort(-allk.astype(np.int32), axis=1), tl["aux"].cpu().numpy().view(np.uint16), srt
cases = [("2001-01-01", 11*365+3, 6, 10, 3, 100.0, 11, 5), ("2001-01-01", 11*365+3, 5, 9, 3, 60.0, 31, 3), ("2001-01-01", 11*365+3, 20, 37, 3, 99.0, 5, 7),
         ("2001-01-01", 11*365+3, 3, 4, 3, 95.0, 11, 5), ("1925-01-01", 36500, 20, 64, 15, 95.0, 11, 5)]
for (start, periods, ny, nx, W, pct, wd, ws) in cases:
    tm = calendar.daily_time_axis(start, periods)
    x = synth.synth_field(synth.make_tables(tm, ny, nx))
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    dcal = hot.upload_calendar(cal)
    bt = binning.hobday_bins()
    xd = torch.from_numpy(x).to(hot.device)
    a = hot.shifting_baseline_tails(xd, dcal, W, 21, bt)
    hot.sync()
    k1, a1, s1 = keys_of(a["tails"], x.shape[1])
    ref = hot.tail_extract(a["out"], dcal, bt)
    hot.sync()
    k2, a2, s2 = keys_of(ref, x.shape[1])
    print(start, ny, nx, "aux equal", np.array_equal(a1, a2), "keys equal", np.array_equal(k1, k2), "sorted", s1, s2)
    t1 = hot.hobday_thresholds_tails(a["tails"], a["out"], dcal, bt, pct/100, wd, ws, ny, nx)["thr_doy_major"].cpu().numpy()
    t2 = hot.hobday_thresholds_tails(ref, a["out"], dcal, bt, pct/100, wd, ws, ny, nx)["thr_doy_major"].cpu().numpy()
    bad = np.argwhere(~((t1 == t2) | (np.isnan(t1) & np.isnan(t2))))
    print("   thr equal", len(bad) == 0, "n bad", len(bad), bad[:8].tolist())
    if len(bad):
        d, c = bad[0]
        print("   ", t1[d, c], t2[d, c], "aux", a1[d, c], a2[d, c])
        tt = a["tails"]["tails"].cpu().numpy().view(np.uint16); tr = ref["tails"].cpu().numpy().view(np.uint16)
        print("   lists shift  :", tt[d, :, :, c, :].reshape(-1).tolist())
        print("   lists extract:", tr[d, :, :, c, :].reshape(-1).tolist())
