"""Experiment: the 8 bands of cfg3 on ONE GPU, alternating between two engines with their own HIP streams (kernels of
neighbouring bands overlap: an HBM-heavy anomaly kernel next to the issue-bound threshold / mask kernels) vs one stream."""
import sys, time
sys.path.insert(0, ".")
import torch
from marex_amd import binning, calendar, synth
from marex_amd.dist import plan_shards, shard_step
from marex_amd.engine import HotPath
T, ny, nx, W = 36500, 720, 1440, 15
tm = calendar.daily_time_axis("1925-01-01", T)
cal = calendar.build_calendar(tm, window_year_baseline=W)
bt = binning.hobday_bins()
shards = plan_shards(ny, nx, 8, 2)
base = HotPath(0)
xs = []
for sh in shards:
    tab = synth.make_tables(tm, sh.ny_in, nx, 20240607, lat_range=(sh.in0, sh.in1, sh.ny_global))
    xs.append(base.synth_field(tab, cell_base=sh.cell_base))
torch.cuda.synchronize()
kw = dict(W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, nx=nx)
for nstream in (1, 2, 3):
    engines = [HotPath(0, own_stream=True) for _ in range(nstream)]
    dcals = [e.upload_calendar(cal) for e in engines]
    wsps = [{} for _ in engines]
    def step():
        outs = []
        for i, (sh, x) in enumerate(zip(shards, xs)):
            e = engines[i % nstream]
            with torch.cuda.stream(e.stream):
                outs.append(shard_step(e, [sh], [x], dcals[i % nstream], workspace=wsps[i % nstream], **kw)[1])
        return outs
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        outs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    tot = sum(o.cpu() for o in outs)
    print(f"{nstream} stream(s): {dt * 1e3:.1f} ms per step, n_extreme {int(tot[3])}", flush=True)
    del engines, dcals, wsps
    torch.cuda.empty_cache()
