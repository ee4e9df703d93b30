"""Measurement (not a test): the headline configuration (100 yr x 720x1440) through the PUBLIC API -- the field resident in
HBM as a labelled device array (151 GB), marex_amd.preprocess_data cutting it into latitude bands that fit the HBM left
over, the Dataset (128 GB of anomalies + 32 GB of events) assembled on the host.  Cross-check against bench.py, which
drives the same kernels through the fused engine call: same synthetic field => same number of extreme events."""
import logging
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import marex_amd  # noqa: E402
from marex_amd import calendar, detect, synth  # noqa: E402
from marex_amd.zarr_io import DeviceDataArray  # noqa: E402

ny, nx, T = 720, 1440, 36500
eng = detect.get_engine(0)
tm = calendar.daily_time_axis("1925-01-01", T)
x = torch.empty((T, ny * nx), dtype=torch.float32, device=eng.device)
t0 = time.time()
for r0 in range(0, ny, 60):
    tab = synth.make_tables(tm, 60, nx, lat_range=(r0, r0 + 60, ny))
    x[:, r0 * nx:(r0 + 60) * nx] = eng.synth_field(tab, cell_base=r0 * nx)
eng.sync()
torch.cuda.empty_cache()
print(f"device field {x.numel() * 4 / 1e9:.1f} GB generated in {time.time() - t0:.1f} s; "
      f"HBM free {torch.cuda.mem_get_info(eng.device)[0] / 1e9:.0f} GB", flush=True)
da = DeviceDataArray(x.view(T, ny, nx), ("time", "lat", "lon"),
                     {"time": tm, "lat": np.linspace(-89.875, 89.875, ny), "lon": np.arange(nx) * 0.25}, name="sst")
logging.basicConfig(level=logging.INFO)
logging.getLogger("marex_amd").setLevel(logging.INFO)
t0 = time.time()
ds = marex_amd.preprocess_data(da, method_anomaly="shifting_baseline", method_extreme="hobday_extreme")
dt = time.time() - t0
n = int(np.count_nonzero(ds.extreme_events.values))
print(f"preprocess_data: {dt:.1f} s wall = {T * ny * nx / 1e6 / dt:.0f} Mcells*ts/s incl. {(ds.dat_anomaly.values.nbytes + n * 0 + ds.extreme_events.values.nbytes) / 1e9:.0f} GB "
      f"of results brought to the host; n_ocean={int(ds.mask.values.sum())} n_extreme={n} "
      f"(bench.py on the same field: 757773 / 1191420079)", flush=True)
print("dims", ds.dat_anomaly.dims, ds.dat_anomaly.shape, ds.thresholds.shape, ds.extreme_events.dtype)
