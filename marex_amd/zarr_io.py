"""Minimal Zarr v2 reader for the stores the reference writes and tests with (first step): directory
store, C order, ``.`` chunk keys, Blosc-1 / LZ4 / byte-shuffle chunks decoded by ``marex_blosc_decompress_h`` (host side of
the C ABI).  Enough to run the hot path on ``tests/data/*.zarr`` of the reference; zstd-compressed coordinate arrays and
writing are not covered."""
from __future__ import annotations

import ctypes as C
import itertools
import json
import os
from typing import Dict

import numpy as np

from . import _lib
from .exceptions import DataValidationError, DependencyError


def _decompress(raw: bytes, nbytes: int) -> bytes:
    lib = _lib.load()
    out = C.create_string_buffer(nbytes)
    n = C.c_int64(0)
    rc = lib.marex_blosc_decompress_h(raw, len(raw), out, nbytes, C.byref(n))
    if rc == -6:
        raise DependencyError("unsupported Blosc codec or filter (only LZ4 / memcpy with byte shuffle are decoded)")
    if rc != 0 or n.value != nbytes:
        raise DataValidationError("malformed Blosc chunk", details=f"code {rc}, decoded {n.value} of {nbytes} bytes")
    return out.raw


def read_array(path: str) -> np.ndarray:
    """One Zarr v2 array directory -> NumPy array."""
    meta = json.load(open(os.path.join(path, ".zarray")))
    if meta.get("zarr_format") != 2 or meta.get("order", "C") != "C" or meta.get("filters"):
        raise DependencyError("only Zarr v2, C order, no filters", details=str({k: meta.get(k) for k in ("zarr_format", "order", "filters")}))
    comp = meta.get("compressor")
    if comp is not None and comp.get("id") != "blosc":
        raise DependencyError(f"unsupported compressor {comp.get('id')!r}")
    shape, chunks, dtype = tuple(meta["shape"]), tuple(meta["chunks"]), np.dtype(meta["dtype"])
    sep = meta.get("dimension_separator", ".")
    fill = meta.get("fill_value")
    out = np.empty(shape, dtype=dtype)
    if fill is not None:
        out[...] = np.nan if fill == "NaN" else fill
    csize = int(np.prod(chunks)) * dtype.itemsize
    grid = [range((s + c - 1) // c) for s, c in zip(shape, chunks)]
    for idx in itertools.product(*grid):
        f = os.path.join(path, sep.join(str(i) for i in idx) if idx else "0")
        if not os.path.exists(f):
            continue
        raw = open(f, "rb").read()
        buf = raw if comp is None else _decompress(raw, csize)
        block = np.frombuffer(buf, dtype=dtype, count=int(np.prod(chunks))).reshape(chunks)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
    return out


def array_attrs(path: str) -> Dict:
    f = os.path.join(path, ".zattrs")
    return json.load(open(f)) if os.path.exists(f) else {}


def decode_cf_time(values: np.ndarray, attrs: Dict) -> np.ndarray:
    """``units: "<unit> since <date>"`` (CF) -> datetime64[D/s]; proleptic Gregorian / standard calendars only."""
    units = attrs.get("units", "")
    if " since " not in units:
        raise DataValidationError("time variable without CF units", details=units)
    unit, origin = units.split(" since ")
    step = {"days": "D", "hours": "h", "minutes": "m", "seconds": "s"}[unit.strip().lower()]
    t0 = np.datetime64(origin.strip().replace(" ", "T"))
    return (t0 + values.astype(np.int64).astype(f"timedelta64[{step}]")).astype("datetime64[D]" if step == "D" else "datetime64[s]")


def read_array_to_device(path: str, eng, lead: int | None = None):
    """Read a Zarr v2 array whose chunks span every dimension but the first (``chunks = (ct, *shape[1:])``, the layout of
    the reference's time-chunked stores) straight into HBM: the COMPRESSED chunk bytes are uploaded, the LZ4 streams are
    decoded one wave each (``marex_lz4_decode_streams``) and the byte shuffle is undone while the elements are placed in
    the destination (``marex_unshuffle_place``).  ``lead`` limits the read to the first ``lead`` steps of dimension 0.
    Returns a torch tensor of the array's dtype and shape ``(lead, *shape[1:])`` on ``eng.device``."""
    import struct

    import torch

    meta = json.load(open(os.path.join(path, ".zarray")))
    comp = meta.get("compressor") or {}
    if meta.get("zarr_format") != 2 or meta.get("order", "C") != "C" or meta.get("filters") or comp.get("id") != "blosc":
        raise DependencyError("device read: Zarr v2, C order, Blosc chunks, no filters")
    shape, chunks, dtype = tuple(meta["shape"]), tuple(meta["chunks"]), np.dtype(meta["dtype"])
    if chunks[1:] != shape[1:]:
        raise DependencyError("device read: chunks must span every dimension but the first", details=f"shape {shape}, chunks {chunks}")
    sep = meta.get("dimension_separator", ".")
    T = shape[0] if lead is None else min(int(lead), shape[0])
    per_step = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    ts = dtype.itemsize
    blobs, pos = [], 0
    s_src, s_cs, s_dst, s_raw = [], [], [], []
    b_off, b_e0, b_ne, b_valid = [], [], [], []
    planes_size, shuffled = 0, None
    for ci in range((T + chunks[0] - 1) // chunks[0]):
        f = os.path.join(path, sep.join([str(ci)] + ["0"] * (len(shape) - 1)))
        raw = open(f, "rb").read()
        _, _, flags, typesize, nbytes, blocksize, cbytes = struct.unpack("<BBBBIII", raw[:16])
        if flags & 0x2 or flags & 0x4 or (flags >> 5) != 1 or typesize != ts or cbytes != len(raw):
            raise DependencyError("device read: LZ4 Blosc frames with byte shuffle only", details=f"{f}: flags {flags:#x}, typesize {typesize}")
        sh = bool(flags & 0x1) and ts > 1
        if shuffled is None:
            shuffled = sh
        elif shuffled != sh:
            raise DependencyError("device read: mixed shuffle settings")
        nblocks = (nbytes + blocksize - 1) // blocksize
        bstarts = struct.unpack(f"<{nblocks}i", raw[16:16 + 4 * nblocks])
        elem_first = ci * chunks[0] * per_step
        elem_valid_end = T * per_step
        for j in range(nblocks):
            bsize = nbytes - j * blocksize if j == nblocks - 1 else blocksize
            leftover = bsize != blocksize
            nsplits = ts if (not (flags & 0x10) and not leftover and ts <= 16 and bsize % ts == 0) else 1
            neblock = bsize // nsplits
            p = bstarts[j]
            for s in range(nsplits):
                (cb,) = struct.unpack("<i", raw[p:p + 4])
                p += 4
                s_src.append(pos + p)
                s_cs.append(cb)
                s_dst.append(planes_size + s * neblock)
                s_raw.append(neblock)
                p += cb
            ne = bsize // ts
            e0 = elem_first + j * (blocksize // ts)
            b_off.append(planes_size)
            b_e0.append(e0)
            b_ne.append(ne)
            b_valid.append(int(max(0, min(ne, elem_valid_end - e0))))
            planes_size += bsize
        blobs.append(raw)
        pos += len(raw)
    dev = eng.device
    comp_d = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
    tab = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)  # noqa: E731
    t_src, t_cs, t_dst, t_raw = tab(s_src, np.int64), tab(s_cs, np.int32), tab(s_dst, np.int64), tab(s_raw, np.int32)
    t_off, t_e0, t_ne, t_valid = tab(b_off, np.int64), tab(b_e0, np.int64), tab(b_ne, np.int32), tab(b_valid, np.int32)
    planes = torch.empty(planes_size, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.empty(T * per_step * ts, dtype=torch.uint8, device=dev)
    eng._bind_stream()
    rc = eng.lib.marex_lz4_decode_streams(eng.ctx.handle, comp_d.data_ptr(), t_src.data_ptr(), t_cs.data_ptr(), t_dst.data_ptr(),
                                          t_raw.data_ptr(), len(s_src), int(max(s_raw)), planes.data_ptr(), status.data_ptr())
    eng.ctx.check(rc, "marex_lz4_decode_streams")
    rc = eng.lib.marex_unshuffle_place(eng.ctx.handle, planes.data_ptr(), t_off.data_ptr(), t_e0.data_ptr(), t_ne.data_ptr(),
                                       t_valid.data_ptr(), len(b_off), int(max(b_ne)), ts, int(bool(shuffled)), out.data_ptr())
    eng.ctx.check(rc, "marex_unshuffle_place")
    eng.sync()
    if int(status.item()) != 0:
        raise DataValidationError("malformed LZ4 stream in a chunk", details=f"{int(status.item())} streams failed")
    tdt = {"float32": torch.float32, "float64": torch.float64, "int32": torch.int32, "int64": torch.int64, "int8": torch.int8,
           "uint8": torch.uint8, "int16": torch.int16}[dtype.name]
    return out.view(tdt).reshape((T,) + shape[1:])


class DeviceDataArray:
    """A labelled array whose data lives in HBM: ``dims`` / ``coords`` / ``attrs`` like a DataArray, the data itself a
    torch tensor in ``device_tensor``.  ``marex_amd.preprocess_data`` & co. take it without a host copy or an upload."""

    def __init__(self, device_tensor, dims, coords, name=None, attrs=None):
        self.device_tensor = device_tensor
        self.dims = tuple(dims)
        self.coords = {k: (v if hasattr(v, "values") else _Coord(np.asarray(v))) for k, v in coords.items()}
        self.name = name
        self.attrs = dict(attrs or {})
        self.shape = tuple(device_tensor.shape)
        self.dtype = np.dtype(str(device_tensor.dtype).replace("torch.", ""))

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def values(self):
        return self.device_tensor.cpu().numpy()


class _Coord:
    def __init__(self, values):
        self.values = values

    def __len__(self):
        return len(self.values)


def open_dataarray_device(store: str, variable: str, eng, dims, time_var: str = "time", lead: int | None = None, coords=None):
    """``xr.open_zarr(store)[variable]`` for the device: the variable's chunks are decoded in HBM
    (``read_array_to_device``), the CF time axis on the host.  ``dims`` names the dimensions, ``coords`` may add spatial
    coordinate arrays (the reference's coordinate arrays are zstd-compressed, which the decoder does not cover)."""
    x = read_array_to_device(os.path.join(store, variable), eng, lead)
    tpath = os.path.join(store, time_var)
    tm = decode_cf_time(read_array(tpath)[: x.shape[0]], array_attrs(tpath))
    c = {dims[0]: tm}
    c.update(coords or {})
    return DeviceDataArray(x, dims, c, name=variable, attrs=array_attrs(os.path.join(store, variable)))
