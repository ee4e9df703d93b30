"""Minimal Zarr v2 reader / writer for the stores the reference reads, writes and tests with: directory
store, C order, ``.`` chunk keys, consolidated metadata, Blosc-1 / LZ4 / byte-shuffle chunks encoded and decoded by
``marex_blosc_compress_h`` / ``marex_blosc_decompress_h`` (host side of the C ABI) or, for reading time-chunked arrays,
decoded in HBM.  Enough to run the hot path on ``tests/data/*.zarr`` of the reference and to write its result the way
``extremes_ds.to_zarr(...)`` does (examples/batch jobs/run_detect.py:55-83); the zstd / bit-shuffle frames of small
coordinate arrays (lat / lon of ``sst_gridded.zarr``) are decoded on the host with the library's own zstd decoder."""
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
        return _decompress_foreign(raw, nbytes)
    if rc != 0 or n.value != nbytes:
        raise DataValidationError("malformed Blosc chunk", details=f"code {rc}, decoded {n.value} of {nbytes} bytes")
    return out.raw


def _decompress_foreign(raw: bytes, nbytes: int) -> bytes:
    """Blosc-1 frames the C decoder declines: an inner zstd codec and / or the bit-shuffle filter (the lat / lon arrays of
    the reference's ``sst_gridded.zarr``: ``{"cname": "zstd", "shuffle": 2}``).  Small coordinate arrays only -- the frame
    is parsed here, the zstd streams go to the library's own decoder (csrc/marex_zstd.hip), and the bit transpose is undone
    with NumPy."""
    import struct

    _, _, flags, typesize, nb, blocksize, cbytes = struct.unpack("<BBBBIII", raw[:16])
    codec = flags >> 5
    if codec not in (1, 4) or nb != nbytes or cbytes != len(raw):
        raise DependencyError("unsupported Blosc codec (LZ4 and zstd frames are decoded)", details=f"flags {flags:#x}")
    lib = _lib.load()

    def inner(stream: bytes, n_out: int) -> bytes:
        """One compressed split: zstd -> the library's own decoder (csrc/marex_zstd.hip); a raw LZ4 block under the bit-shuffle
        filter (no store of the reference has one) -> pyarrow's lz4_raw codec when that package is present."""
        if codec == 4:
            out = C.create_string_buffer(max(n_out, 1))
            got = C.c_int64(0)
            rc = lib.marex_zstd_decompress_h(stream, len(stream), out, n_out, C.byref(got))
            if rc != 0 or got.value != n_out:
                raise DataValidationError("malformed Blosc chunk", details=f"zstd stream: code {rc}, decoded {got.value} of {n_out} bytes")
            return out.raw[:n_out]
        try:
            import pyarrow as pa
        except Exception as e:  # pragma: no cover
            raise DependencyError("a bit-shuffled LZ4 Blosc frame needs pyarrow's lz4_raw codec to be decoded", details=str(e))
        return pa.Codec("lz4_raw").decompress(stream, decompressed_size=n_out).to_pybytes()

    nblocks = (nbytes + blocksize - 1) // blocksize
    out = bytearray()
    for j in range(nblocks):
        bsize = nbytes - j * blocksize if j == nblocks - 1 else blocksize
        leftover = bsize != blocksize
        nsplits = typesize if (not flags & 0x10 and not leftover and typesize <= 16 and blocksize // typesize >= 128) else 1
        ne = bsize // nsplits
        (p,) = struct.unpack("<i", raw[16 + 4 * j: 20 + 4 * j])
        blk = bytearray()
        for _ in range(nsplits):
            (cb,) = struct.unpack("<i", raw[p: p + 4])
            p += 4
            blk += raw[p: p + cb] if cb == ne else inner(raw[p: p + cb], ne)
            p += cb
        n_el = bsize // typesize
        if flags & 0x4:  # bit shuffle: row (byte k, bit i) holds that bit of every element, 8 elements per byte, LSB first;
            n8 = n_el - n_el % 8  # elements beyond a multiple of 8 (and a ragged tail) are stored as they are
            rows = np.frombuffer(bytes(blk[: n8 * typesize]), np.uint8).reshape(typesize * 8, n8 // 8)
            bits = np.unpackbits(rows, axis=1, bitorder="little")  # [byte*8 + bit, element]
            body = np.packbits(bits.reshape(typesize, 8, n8).transpose(2, 0, 1), axis=2, bitorder="little").reshape(-1)
            blk = body.tobytes() + bytes(blk[n8 * typesize:])
        elif flags & 0x1 and typesize > 1:
            body = np.frombuffer(bytes(blk[: n_el * typesize]), np.uint8).reshape(typesize, n_el).T.tobytes()
            blk = body + bytes(blk[n_el * typesize:])
        out += blk
    return bytes(out)


def decode_fill_value(fv, dtype: np.dtype):
    """Zarr v2 ``fill_value`` of the array metadata as a number (or None): floats may be spelled "NaN", "Infinity",
    "-Infinity"; anything else that is not a number (base64 strings of structured / byte dtypes) is outside this reader."""
    if fv is None:
        return None
    if isinstance(fv, str):
        special = {"NaN": float("nan"), "Infinity": float("inf"), "-Infinity": float("-inf")}
        if fv in special and dtype.kind == "f":
            return special[fv]
        raise DependencyError(f"unsupported fill_value {fv!r} for dtype {dtype}", details="only numbers, \"NaN\", \"Infinity\" and \"-Infinity\"")
    if isinstance(fv, bool) or isinstance(fv, (int, float)):
        return fv
    raise DependencyError(f"unsupported fill_value {fv!r} for dtype {dtype}")


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
    fill = decode_fill_value(meta.get("fill_value"), dtype)
    out = np.empty(shape, dtype=dtype)
    if fill is not None:
        out[...] = fill
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


def plan_blosc_frame(raw: bytes, typesize: int, expected_nbytes: int, where: str = "<frame>"):
    """Parse and VALIDATE the header of one Blosc-1 / LZ4 frame for the device decoder (the host decoder
    ``marex_blosc_decompress_h`` makes the same checks for itself).  The device kernels bound their reads only by the
    stream sizes uploaded with the chunk, so every offset and size is checked against the frame here, before anything is
    sent to the GPU: a truncated or corrupt chunk raises ``DataValidationError`` instead of faulting on the device.

    Returns ``(shuffled, streams, blocks)``: ``streams`` = ``(offset of the compressed bytes in the frame, compressed size,
    destination offset in the chunk's byte planes, raw size)``, ``blocks`` = ``(plane offset, first element relative to the
    chunk, elements, bytes)``."""
    import struct

    def bad(msg):
        return DataValidationError("malformed Blosc frame", details=f"{where}: {msg}")

    if len(raw) < 16:
        raise bad(f"{len(raw)} bytes, shorter than the 16-byte header")
    _, _, flags, ts, nbytes, blocksize, cbytes = struct.unpack("<BBBBIII", raw[:16])
    if flags & 0x2 or flags & 0x4 or (flags >> 5) != 1 or ts != typesize:
        raise DependencyError("device read: LZ4 Blosc frames with byte shuffle only", details=f"{where}: flags {flags:#x}, typesize {ts}")
    if cbytes != len(raw):
        raise bad(f"header says {cbytes} compressed bytes, the file holds {len(raw)}")
    if nbytes != expected_nbytes:
        raise bad(f"header says {nbytes} decoded bytes, the chunk shape needs {expected_nbytes}")
    if blocksize <= 0 or nbytes <= 0:
        raise bad(f"blocksize {blocksize}, nbytes {nbytes}")
    nblocks = (nbytes + blocksize - 1) // blocksize
    table_end = 16 + 4 * nblocks
    if table_end > len(raw):
        raise bad(f"block table of {nblocks} entries does not fit {len(raw)} bytes")
    bstarts = struct.unpack(f"<{nblocks}i", raw[16:table_end])
    shuffled = bool(flags & 0x1) and ts > 1
    streams, blocks = [], []
    for j in range(nblocks):
        bsize = nbytes - j * blocksize if j == nblocks - 1 else blocksize
        leftover = bsize != blocksize
        nsplits = ts if (not (flags & 0x10) and not leftover and ts <= 16 and blocksize // ts >= 128 and bsize % ts == 0) else 1
        neblock = bsize // nsplits
        p = bstarts[j]
        if p < table_end or p + 4 > len(raw):
            raise bad(f"block {j} starts at {p}, outside [{table_end}, {len(raw) - 4}]")
        for s in range(nsplits):
            if p + 4 > len(raw):
                raise bad(f"block {j} stream {s}: size field at {p} past the end of the frame")
            (cb,) = struct.unpack("<i", raw[p:p + 4])
            p += 4
            if cb < 0 or cb > neblock or p + cb > len(raw):
                raise bad(f"block {j} stream {s}: {cb} compressed bytes at {p} (raw size {neblock}, frame {len(raw)})")
            streams.append((p, cb, j * blocksize + s * neblock, neblock))
            p += cb
        blocks.append((j * blocksize, j * (blocksize // ts), bsize // ts, bsize))
    return shuffled, streams, blocks


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
    chunk_bytes = int(np.prod(chunks)) * ts
    missing = []  # (first element, one-past-last element) of chunks without a file: left at the fill value, as read_array does
    for ci in range((T + chunks[0] - 1) // chunks[0]):
        f = os.path.join(path, sep.join([str(ci)] + ["0"] * (len(shape) - 1)))
        elem_first = ci * chunks[0] * per_step
        elem_valid_end = T * per_step
        if not os.path.exists(f):
            missing.append((elem_first, min(elem_first + chunks[0] * per_step, elem_valid_end)))
            continue
        raw = open(f, "rb").read()
        sh, streams, blocks = plan_blosc_frame(raw, ts, chunk_bytes, f)
        if shuffled is None:
            shuffled = sh
        elif shuffled != sh:
            raise DependencyError("device read: mixed shuffle settings")
        for (p, cb, dst, neblock) in streams:
            s_src.append(pos + p)
            s_cs.append(cb)
            s_dst.append(planes_size + dst)
            s_raw.append(neblock)
        for (off, e_rel, ne, bsize) in blocks:
            e0 = elem_first + e_rel
            b_off.append(planes_size + off)
            b_e0.append(e0)
            b_ne.append(ne)
            b_valid.append(int(max(0, min(ne, elem_valid_end - e0))))
        planes_size += chunk_bytes
        blobs.append(raw)
        pos += len(raw)
    dev = eng.device
    planes = torch.empty(max(planes_size, 1), dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.empty(T * per_step * ts, dtype=torch.uint8, device=dev)
    tdt = {"float32": torch.float32, "float64": torch.float64, "int32": torch.int32, "int64": torch.int64, "int8": torch.int8,
           "uint8": torch.uint8, "int16": torch.int16, "bool": torch.bool}[dtype.name]
    if missing:
        fv = decode_fill_value(meta.get("fill_value"), dtype)  # the host reader's decoding; no fill value: NaN for floats, else 0
        if fv is None:
            fv = float("nan") if dtype.kind == "f" else 0
        typed = out.view(tdt)
        for e0, e1 in missing:
            typed[e0:e1] = fv
    if not s_src:
        return out.view(tdt).reshape((T,) + shape[1:])
    comp_d = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
    tab = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)  # noqa: E731
    t_src, t_cs, t_dst, t_raw = tab(s_src, np.int64), tab(s_cs, np.int32), tab(s_dst, np.int64), tab(s_raw, np.int32)
    t_off, t_e0, t_ne, t_valid = tab(b_off, np.int64), tab(b_e0, np.int64), tab(b_ne, np.int32), tab(b_valid, np.int32)
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


# ------------------------------------------------------------------------------------------------
# writing (examples/batch jobs/run_detect.py:83: ``extremes_ds.to_zarr(output_file, mode="w")``)
# ------------------------------------------------------------------------------------------------
_BLOSC_LZ4 = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}  # numcodecs' / xarray's default


def _compress(buf, typesize: int) -> bytes:
    lib = _lib.load()
    src = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    out = np.empty(src.size + 16, dtype=np.uint8)
    n = C.c_int64(0)
    rc = lib.marex_blosc_compress_h(src.ctypes.data, src.size, typesize, 1, 0, out.ctypes.data, out.size, C.byref(n))
    if rc != 0:
        raise DataValidationError("Blosc compression failed", details=f"code {rc}, {src.size} bytes")
    return out[: n.value].tobytes()


def _zarr_dtype(dt: np.dtype) -> str:
    dt = np.dtype(dt)
    if dt == np.bool_:
        return "|b1"
    if dt.kind not in "fiu" or dt.byteorder == ">":
        raise DependencyError(f"unsupported dtype {dt} (little-endian float / int / bool arrays are written)")
    return dt.str if dt.itemsize > 1 else "|" + dt.str[1:]


def _json_attr(v):
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [_json_attr(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _json_attr(x) for k, x in v.items()}
    return v


def write_array(path: str, data, chunks=None, dims=None, attrs=None, compress: bool = True, threads: int = 8) -> Dict:
    """Write one Zarr v2 array directory.  ``data`` is a NumPy array or a torch tensor (host or device; a device tensor
    is brought over one chunk of its first dimension at a time, so a field that fills HBM never needs a host copy of
    itself).  ``chunks`` defaults to the whole array; chunks are compressed by a small thread pool (the C call releases the
    GIL).  Returns ``{".zarray": ..., ".zattrs": ...}`` for the consolidated metadata."""
    from concurrent.futures import ThreadPoolExecutor

    is_torch = hasattr(data, "device") and hasattr(data, "cpu")
    shape = tuple(int(n) for n in data.shape)
    dtype = np.dtype(str(data.dtype).replace("torch.", "")) if is_torch else np.dtype(data.dtype)
    zdtype = _zarr_dtype(dtype)
    chunks = tuple(int(min(max(c, 1), max(n, 1))) for c, n in zip(shape if chunks is None else chunks, shape))
    if len(chunks) != len(shape):
        raise DataValidationError("chunks do not match the array rank", details=f"shape {shape}, chunks {chunks}")
    os.makedirs(path, exist_ok=True)
    fill = "NaN" if dtype.kind == "f" else None  # what xarray writes (see the reference's stores)
    zarray = {"zarr_format": 2, "shape": list(shape), "chunks": list(chunks), "dtype": zdtype, "order": "C", "filters": None,
              "fill_value": fill, "compressor": dict(_BLOSC_LZ4) if compress else None}
    zattrs = {k: _json_attr(v) for k, v in (attrs or {}).items()}
    if dims is not None:
        zattrs["_ARRAY_DIMENSIONS"] = list(dims)  # xarray's convention for naming Zarr dimensions
    json.dump(zarray, open(os.path.join(path, ".zarray"), "w"), indent=1)
    json.dump(zattrs, open(os.path.join(path, ".zattrs"), "w"), indent=1)

    def slab(i0):  # host copy of rows [i0*c0, (i0+1)*c0) of the first dimension
        if not shape:
            return np.asarray(data.cpu().numpy() if is_torch else data).reshape(())
        part = data[i0 * chunks[0]: (i0 + 1) * chunks[0]]
        return part.cpu().numpy() if is_torch else np.asarray(part)

    def put(idx, block):
        full = np.zeros(chunks, dtype=dtype) if block.shape != chunks else block  # edge chunks are padded to full size
        if full is not block:
            if dtype.kind == "f":
                full[...] = np.nan
            full[tuple(slice(0, n) for n in block.shape)] = block
        raw = np.ascontiguousarray(full)
        payload = _compress(raw, dtype.itemsize) if compress else raw.tobytes()
        name = ".".join(str(i) for i in idx) if idx else "0"
        with open(os.path.join(path, name), "wb") as f:
            f.write(payload)

    grid = [range((n + c - 1) // c) for n, c in zip(shape, chunks)]
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        for i0 in (grid[0] if shape else [0]):
            part = slab(i0)
            jobs = []
            for rest in itertools.product(*grid[1:]):
                sel = tuple(slice(j * c, min((j + 1) * c, n)) for j, c, n in zip(rest, chunks[1:], shape[1:]))
                block = part[(slice(None),) + sel] if shape else part
                jobs.append(pool.submit(put, ((i0,) + rest) if shape else (), block))
            for j in jobs:
                j.result()
    return {".zarray": zarray, ".zattrs": zattrs}


def encode_cf_time(values) -> tuple:
    """datetime64 axis -> (int64 ``days since`` the first day, CF attrs): the encoding xarray picks for a daily axis."""
    t = np.asarray(values).astype("datetime64[D]")
    t0 = t[0] if t.size else np.datetime64("1970-01-01")
    return (t - t0).astype(np.int64), {"units": f"days since {t0} 00:00:00", "calendar": "proleptic_gregorian"}


def write_dataset(store: str, ds, chunks: Dict[str, int] | None = None, compress: bool = True, threads: int = 8) -> None:
    """``ds.to_zarr(store, mode="w")`` for the Dataset ``preprocess_data`` returns (stand-in or device-resident variables):
    every data variable and coordinate becomes an array directory, attributes go to ``.zattrs``, and ``.zmetadata`` holds
    the consolidated copy that ``xr.open_zarr`` reads first.  ``chunks`` maps dimension names to chunk lengths (default:
    ``time`` in steps of 25 -- the reference's output chunking, detect.py:785-792 -- everything else whole)."""
    import shutil

    chunks = dict({"time": 25}, **(chunks or {}))
    if os.path.isdir(store):  # mode="w" replaces a store -- but only something that IS a Zarr store (or an empty directory)
        entries = os.listdir(store)
        if entries and not any(e in entries for e in (".zgroup", ".zarray", ".zmetadata")):
            raise DataValidationError("refusing to overwrite a directory that is not a Zarr store", details=store)
        shutil.rmtree(store)
    os.makedirs(store)
    meta = {".zgroup": {"zarr_format": 2}, ".zattrs": {k: _json_attr(v) for k, v in getattr(ds, "attrs", {}).items()}}
    json.dump(meta[".zgroup"], open(os.path.join(store, ".zgroup"), "w"))
    json.dump(meta[".zattrs"], open(os.path.join(store, ".zattrs"), "w"), indent=1)
    coord_names = list(ds.coords)
    items = [(k, ds.coords[k], True) for k in coord_names] + [(k, v, False) for k, v in ds.data_vars.items()]
    for name, var, is_coord in items:
        data = getattr(var, "device_tensor", None)
        if data is None:
            data = var.data if hasattr(var, "data") else var.values
        attrs = dict(getattr(var, "attrs", {}))
        dims = tuple(var.dims)
        if not hasattr(data, "cpu") and np.asarray(data).dtype.kind == "M":
            data, tattrs = encode_cf_time(np.asarray(data))
            attrs.update(tattrs)
        if not is_coord:
            aux = [c for c in coord_names if c not in dims and set(ds.coords[c].dims) <= set(dims) and ds.coords[c].dims]
            if aux:
                attrs["coordinates"] = " ".join(aux)  # non-index coordinates (lat / lon of an unstructured mesh)
        ch = None if is_coord else tuple(chunks.get(d, n) for d, n in zip(dims, data.shape))
        m = write_array(os.path.join(store, name), data, ch, dims, attrs, compress, threads)
        meta[f"{name}/.zarray"] = m[".zarray"]
        meta[f"{name}/.zattrs"] = m[".zattrs"]
    json.dump({"zarr_consolidated_format": 1, "metadata": meta}, open(os.path.join(store, ".zmetadata"), "w"), indent=1)


def read_dataset(store: str):
    """``xr.open_zarr(store)`` into the stand-in Dataset (host arrays): variables named like one of their own dimensions
    are coordinates, ``_ARRAY_DIMENSIONS`` names the dimensions, CF time axes are decoded."""
    from .xr_compat import _MiniDataArray, _MiniDataset

    names = sorted(d for d in os.listdir(store) if os.path.exists(os.path.join(store, d, ".zarray")))
    arrays = {}
    for n in names:
        p = os.path.join(store, n)
        at = array_attrs(p)
        v = read_array(p)
        dims = tuple(at.pop("_ARRAY_DIMENSIONS", [f"dim_{i}" for i in range(v.ndim)]))
        if " since " in str(at.get("units", "")):
            v = decode_cf_time(v, at)
            at.pop("units", None)
            at.pop("calendar", None)
        if json.load(open(os.path.join(p, ".zarray")))["dtype"] == "|b1":
            v = v.astype(bool)
        arrays[n] = (dims, v, at)
    aux = set()
    for dims, v, at in arrays.values():
        aux.update(str(at.pop("coordinates", "")).split())
    coords = {n: _MiniDataArray(v, d, None, n, at) for n, (d, v, at) in arrays.items() if d == (n,) or n in aux}
    gattrs = array_attrs(store)
    ds = _MiniDataset(None, coords, gattrs)
    for n, (d, v, at) in arrays.items():
        if n not in coords:
            ds[n] = _MiniDataArray(v, d, {c: coords[c] for c in coords if set(coords[c].dims) <= set(d)}, n, at)
    return ds
