"""One-off measurement (not a pytest file): device-side decoding of Blosc-LZ4 chunks.  The 183 SST chunks of the fixture
(96 000 bytes each: two stored byte planes, two LZ4 planes) are decoded REP times over into one big array."""
import os
import struct
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from marex_amd.engine import HotPath

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures", "sst_gridded.zarr", "to")
hot = HotPath(0)
blobs, pos = [], 0
src, cs, dst, raw_ = [], [], [], []
boff, be0, bne = [], [], []
planes = 0
for ci in range(183):
    raw = open(os.path.join(p, f"{ci}.0.0"), "rb").read()
    _, _, flags, ts, nbytes, bs, cb = struct.unpack("<BBBBIII", raw[:16])
    q = struct.unpack("<i", raw[16:20])[0]
    for s in range(4):
        (c,) = struct.unpack("<i", raw[q:q + 4])
        q += 4
        src.append(pos + q); cs.append(c); dst.append(planes + s * 24000); raw_.append(24000)
        q += c
    boff.append(planes); be0.append(ci * 24000); bne.append(24000)
    planes += nbytes
    blobs.append(raw); pos += len(raw)
n1 = len(src)
src = np.tile(np.array(src, np.int64), REP)
cs = np.tile(np.array(cs, np.int32), REP)
raw_ = np.tile(np.array(raw_, np.int32), REP)
dst = (np.array(dst, np.int64)[None, :] + (np.arange(REP, dtype=np.int64) * planes)[:, None]).reshape(-1)
boff = (np.array(boff, np.int64)[None, :] + (np.arange(REP, dtype=np.int64) * planes)[:, None]).reshape(-1)
be0 = (np.array(be0, np.int64)[None, :] + (np.arange(REP, dtype=np.int64) * 183 * 24000)[:, None]).reshape(-1)
bne = np.tile(np.array(bne, np.int32), REP)
dev = hot.device
comp = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
T = lambda a: torch.from_numpy(a).to(dev)
t_src, t_cs, t_dst, t_raw, t_off, t_e0, t_ne = T(src), T(cs), T(dst), T(raw_), T(boff), T(be0), T(bne)
pl = torch.empty(planes * REP, dtype=torch.uint8, device=dev)
out = torch.empty(planes * REP, dtype=torch.uint8, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
hot._bind_stream()


def step():
    hot.ctx.check(hot.lib.marex_lz4_decode_streams(hot.ctx.handle, comp.data_ptr(), t_src.data_ptr(), t_cs.data_ptr(), t_dst.data_ptr(),
                                                   t_raw.data_ptr(), len(src), 24000, pl.data_ptr(), status.data_ptr()), "decode")
    hot.ctx.check(hot.lib.marex_unshuffle_place(hot.ctx.handle, pl.data_ptr(), t_off.data_ptr(), t_e0.data_ptr(), t_ne.data_ptr(),
                                                t_ne.data_ptr(), len(boff), 24000, 4, 1, out.data_ptr()), "place")


step(); hot.sync()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    step()
hot.sync()
dt = (time.perf_counter() - t0) / K
print({"raw_GB": round(planes * REP / 1e9, 3), "compressed_GB": round(pos * REP / 1e9, 3), "ms": round(dt * 1e3, 2),
       "decoded_GB_per_s": round(planes * REP / dt / 1e9, 1), "status": int(status.item())})
