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
