"""Measurement (not a test): preprocess_data on a HOST array larger than the HBM left free -- the field is cut into
latitude bands automatically (marex_amd.detect.plan_blocks), each band uploaded, processed and brought back.
Usage: python tests/measure_out_of_core.py [ny] [hold_GB]   (hold_GB of HBM are blocked first to force several bands)"""
import logging
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import marex_amd  # noqa: E402
from marex_amd import calendar, detect, synth  # noqa: E402
from marex_amd.xr_compat import DataArray  # noqa: E402

ny = int(sys.argv[1]) if len(sys.argv) > 1 else 300
hold_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
nx, T = 1440, 36500
eng = detect.get_engine(0)
tm = calendar.daily_time_axis("1925-01-01", T)
x = np.empty((T, ny, nx), dtype=np.float32)
t0 = time.time()
for r0 in range(0, ny, 50):  # the synthetic field is generated on the device band by band and parked in host memory
    r1 = min(ny, r0 + 50)
    tab = synth.make_tables(tm, r1 - r0, nx, lat_range=(r0, r1, ny))
    x[:, r0:r1, :] = eng.synth_field(tab, cell_base=r0 * nx).cpu().numpy().reshape(T, r1 - r0, nx)
print(f"host field {x.nbytes / 1e9:.1f} GB generated in {time.time() - t0:.1f} s", flush=True)
torch.cuda.empty_cache()
hold = torch.empty(int(hold_gb * 1e9), dtype=torch.uint8, device=eng.device)
da = DataArray(x, dims=("time", "lat", "lon"), coords={"time": tm, "lat": np.linspace(-89.875, 89.875, ny), "lon": np.arange(nx) * 0.25})
logging.getLogger("marex_amd").setLevel(logging.INFO)
logging.basicConfig(level=logging.INFO)


def run(blocks):
    if blocks:
        os.environ["MAREX_BLOCKS"] = str(blocks)
    else:
        os.environ.pop("MAREX_BLOCKS", None)
    t = time.time()
    ds = marex_amd.preprocess_data(da, method_anomaly="shifting_baseline", method_extreme="hobday_extreme")
    dt = time.time() - t
    n = int(np.count_nonzero(ds.extreme_events.values))
    thr = ds.thresholds.values.copy()
    sig = float(np.nansum(ds.dat_anomaly.values[::97].astype(np.float64)))
    print(f"blocks={blocks or 'auto'}: {dt:.1f} s, {T * ny * nx / 1e6 / dt:.0f} Mcells*ts/s end to end from host memory, "
          f"n_extreme={n}, anomaly checksum={sig:.6f}", flush=True)
    return n, thr, sig


a = run(0)
b = run(7)
print("identical:", a[0] == b[0] and np.array_equal(a[1], b[1], equal_nan=True) and a[2] == b[2])
