"""Zarr v2 writer: the LZ4 encoder and the Blosc frames it produces are checked with an independent
decoder (pyarrow's ``lz4_raw`` codec = the LZ4 block format) and an independent un-shuffle in NumPy, the array / dataset
layer by round trips and against the metadata layout of the reference's own stores (tests/golden/ref_fixtures)."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from marex_amd import _lib, zarr_io
from marex_amd.xr_compat import _MiniDataArray, _MiniDataset

pa = pytest.importorskip("pyarrow")
HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "ref_fixtures")


def compress(buf: bytes, typesize: int, shuffle: int = 1, blocksize: int = 0) -> bytes:
    lib = _lib.load()
    out = C.create_string_buffer(len(buf) + 16)
    n = C.c_int64(0)
    rc = lib.marex_blosc_compress_h(buf, len(buf), typesize, shuffle, blocksize, out, len(buf) + 16, C.byref(n))
    assert rc == 0
    return out.raw[: n.value]


def independent_decode(frame: bytes) -> bytes:
    """Blosc-1 frame -> bytes with pyarrow's LZ4 block decoder and a NumPy un-shuffle (c-blosc 1.x layout)."""
    ver, verlz, flags, typesize, nbytes, blocksize, cbytes = struct.unpack("<BBBBIII", frame[:16])
    assert ver == 2 and verlz == 1 and cbytes == len(frame)
    if flags & 0x2:
        return frame[16:16 + nbytes]
    assert flags >> 5 == 1 and not flags & 0x4
    codec = pa.Codec("lz4_raw")
    nblocks = (nbytes + blocksize - 1) // blocksize
    out = bytearray()
    for j in range(nblocks):
        bsize = nbytes - j * blocksize if j == nblocks - 1 else blocksize
        leftover = bsize != blocksize
        nsplits = typesize if (not flags & 0x10 and not leftover and typesize <= 16 and blocksize // typesize >= 128) else 1
        ne = bsize // nsplits
        (p,) = struct.unpack("<i", frame[16 + 4 * j: 20 + 4 * j])
        planes = bytearray()
        for _ in range(nsplits):
            (cb,) = struct.unpack("<i", frame[p: p + 4])
            p += 4
            planes += frame[p: p + cb] if cb == ne else codec.decompress(frame[p: p + cb], decompressed_size=ne).to_pybytes()
            p += cb
        if flags & 0x1 and typesize > 1:
            n_el = bsize // typesize
            body = np.frombuffer(bytes(planes[: n_el * typesize]), np.uint8).reshape(typesize, n_el).T.tobytes()
            planes = body + bytes(planes[n_el * typesize:])
        out += planes
    return bytes(out)


def buffers():
    rng = np.random.default_rng(7)
    yield b""
    for n in (1, 4, 5, 12, 13, 14, 17, 64, 255, 256, 270, 4096):
        yield bytes(n)
        yield rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        yield (b"abcd" * n)[:n]
    yield bytes(300000)  # a run far longer than one match-length byte chain step
    yield rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()  # incompressible -> stored frame
    yield (rng.random(500000) > 0.97).astype(np.uint8).tobytes()  # a sparse mask
    yield np.round(rng.normal(size=100000), 2).astype(np.float32).tobytes()
    yield rng.integers(0, 4, 70001, dtype=np.uint8).tobytes() + bytes(70000) + b"xyz" * 30000
    yield np.cumsum(rng.integers(-2, 3, 80000)).astype(np.int64).tobytes()


@pytest.mark.parametrize("typesize,blocksize", [(1, 0), (4, 0), (8, 0), (4, 4096), (4, 400), (2, 1000), (1, 65536), (3, 0)])
def test_frames_decode_with_independent_decoder(typesize, blocksize):
    lib = _lib.load()
    for buf in buffers():
        frame = compress(buf, typesize, 1, blocksize)
        assert len(frame) <= len(buf) + 16
        assert independent_decode(frame) == buf
        out = C.create_string_buffer(max(len(buf), 1))
        n = C.c_int64(0)
        assert lib.marex_blosc_decompress_h(frame, len(frame), out, len(buf), C.byref(n)) == 0 and n.value == len(buf)
        assert out.raw[: len(buf)] == buf


def test_decoder_reads_streams_of_another_encoder():
    """Frames assembled here around pyarrow's LZ4 encoder (different match choices than ours) decode identically."""
    lib = _lib.load()
    codec = pa.Codec("lz4_raw")
    rng = np.random.default_rng(3)
    a = np.round(rng.normal(size=(64, 1024)), 1).astype(np.float32)
    raw = a.tobytes()
    ts, bs = 4, 65536
    frame = bytearray(struct.pack("<BBBBIII", 2, 1, 0x21, ts, len(raw), bs, 0))
    nblocks = len(raw) // bs
    frame += bytes(4 * nblocks)
    for j in range(nblocks):
        struct.pack_into("<i", frame, 16 + 4 * j, len(frame))
        planes = np.frombuffer(raw[j * bs: (j + 1) * bs], np.uint8).reshape(-1, ts).T.copy()
        for k in range(ts):
            c = codec.compress(planes[k].tobytes(), asbytes=True)
            if len(c) >= planes[k].size:
                c = planes[k].tobytes()
            frame += struct.pack("<i", len(c)) + c
    struct.pack_into("<I", frame, 12, len(frame))
    out = C.create_string_buffer(len(raw))
    n = C.c_int64(0)
    assert lib.marex_blosc_decompress_h(bytes(frame), len(frame), out, len(raw), C.byref(n)) == 0
    assert out.raw == raw


def test_compression_ratio_is_sane():
    """Greedy single-probe LZ4 should land near the reference encoder on the data this path writes (a sparse mask)."""
    rng = np.random.default_rng(1)
    m = np.zeros((25, 180, 360), np.uint8)
    for _ in range(60):
        t, y, x = rng.integers(0, 25), rng.integers(0, 160), rng.integers(0, 330)
        m[t, y: y + 20, x: x + 30] = 1
    raw = m.tobytes()
    ours = len(compress(raw, 1))
    theirs = len(pa.Codec("lz4_raw").compress(raw, asbytes=True))
    assert ours < 0.05 * len(raw) and ours < 1.5 * theirs + 64


@pytest.mark.parametrize("dtype,shape,chunks", [
    (np.float32, (70, 20, 40), (30, 20, 40)), (np.bool_, (61, 18, 36), (25, 18, 36)), (np.int32, (3, 405), None),
    (np.float64, (5, 7), (2, 3)), (np.int8, (32, 18, 36), (2, 18, 36)), (np.float32, (64, 90, 72), (16, 45, 36)),
    (np.float32, (), None), (np.int64, (0,), None)])
def test_array_round_trip(tmp_path, dtype, shape, chunks):
    rng = np.random.default_rng(5)
    a = (rng.random(shape) > 0.8) if dtype == np.bool_ else (rng.normal(size=shape) * 10).astype(dtype)
    if np.dtype(dtype).kind == "f" and a.ndim:
        a.reshape(-1)[::7] = np.nan
    p = str(tmp_path / "a")
    zarr_io.write_array(p, a, chunks, dims=[f"d{i}" for i in range(a.ndim)], attrs={"units": "K", "n": np.int64(3)})
    b = zarr_io.read_array(p)
    assert b.shape == a.shape and np.array_equal(a, b.astype(a.dtype), equal_nan=np.dtype(dtype).kind == "f")
    assert zarr_io.array_attrs(p)["_ARRAY_DIMENSIONS"] == [f"d{i}" for i in range(a.ndim)]
    assert zarr_io.array_attrs(p)["n"] == 3
    raw = zarr_io.read_array  # uncompressed variant reads back too
    zarr_io.write_array(p + "_raw", a, chunks, compress=False)
    assert np.array_equal(a, raw(p + "_raw").astype(a.dtype), equal_nan=np.dtype(dtype).kind == "f")


def test_metadata_layout_matches_reference_store(tmp_path):
    """Same keys / conventions as the stores xarray wrote for the reference (extremes_gridded.zarr)."""
    ref = json.load(open(os.path.join(FIX, "extremes_gridded.zarr", ".zmetadata")))
    ev = zarr_io.read_array(os.path.join(FIX, "extremes_gridded.zarr", "extreme_events"))
    lat = zarr_io.read_array(os.path.join(FIX, "extremes_gridded.zarr", "lat"))
    lon = zarr_io.read_array(os.path.join(FIX, "extremes_gridded.zarr", "lon"))
    tpath = os.path.join(FIX, "extremes_gridded.zarr", "time")
    tm = zarr_io.decode_cf_time(zarr_io.read_array(tpath), zarr_io.array_attrs(tpath))
    msk = zarr_io.read_array(os.path.join(FIX, "extremes_gridded.zarr", "mask"))
    coords = {"time": tm, "lat": lat, "lon": lon}
    ds = _MiniDataset({"extreme_events": _MiniDataArray(ev, ("time", "lat", "lon"), coords),
                       "mask": _MiniDataArray(msk, ("lat", "lon"), {"lat": lat, "lon": lon})}, attrs={"threshold_percentile": 95})
    store = str(tmp_path / "out.zarr")
    ds.to_zarr(store, mode="w", chunks={"time": 2})
    ours = json.load(open(os.path.join(store, ".zmetadata")))
    assert ours["zarr_consolidated_format"] == ref["zarr_consolidated_format"] == 1
    assert ours["metadata"][".zgroup"] == ref["metadata"][".zgroup"]
    for key in ("extreme_events/.zarray", "mask/.zarray", "lat/.zarray", "lon/.zarray"):
        o, r = ours["metadata"][key], ref["metadata"][key]
        assert set(o) == set(r)
        for k in ("zarr_format", "shape", "chunks", "dtype", "order", "filters", "compressor", "fill_value"):
            assert o[k] == r[k], (key, k, o[k], r[k])
    for key in ("extreme_events/.zattrs", "mask/.zattrs", "lat/.zattrs", "lon/.zattrs", "time/.zattrs"):
        assert ours["metadata"][key]["_ARRAY_DIMENSIONS"] == ref["metadata"][key]["_ARRAY_DIMENSIONS"]
    assert ours["metadata"]["time/.zattrs"]["units"].startswith("days since ")
    # same bytes back, and files of the same order of size as the reference's own chunks
    back = zarr_io.read_dataset(store)
    assert np.array_equal(back.extreme_events.values, ev) and np.array_equal(back.mask.values, msk)
    assert np.array_equal(back.time.values.astype("datetime64[D]"), tm.astype("datetime64[D]"))
    assert back.attrs["threshold_percentile"] == 95
    size = lambda d: sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if not f.startswith("."))  # noqa: E731
    assert size(os.path.join(store, "extreme_events")) < 1.5 * size(os.path.join(FIX, "extremes_gridded.zarr", "extreme_events"))


def test_dataset_round_trip_unstructured(tmp_path):
    rng = np.random.default_rng(2)
    T, Cn = 40, 405
    tm = np.arange("2001-01-01", "2001-02-10", dtype="datetime64[D]")
    lat, lon = rng.uniform(-90, 90, Cn).astype(np.float32), rng.uniform(0, 360, Cn).astype(np.float32)
    coords = {"time": tm, "lat": ("ncells", lat), "lon": ("ncells", lon)}
    ds = _MiniDataset({"dat_anomaly": _MiniDataArray(rng.normal(size=(T, Cn)).astype(np.float32), ("time", "ncells"), coords),
                       "extreme_events": _MiniDataArray(rng.random((T, Cn)) > 0.95, ("time", "ncells"), coords),
                       "thresholds": _MiniDataArray(rng.normal(size=(Cn, 366)).astype(np.float32), ("ncells", "dayofyear"),
                                                    {"dayofyear": np.arange(1, 367)}),
                       "neighbours": _MiniDataArray(rng.integers(1, Cn, (3, Cn)).astype(np.int32), ("nv", "ncells"))},
                      attrs={"method_anomaly": "shifting_baseline", "preprocessing_steps": ["a", "b"]})
    store = str(tmp_path / "u.zarr")
    ds.to_zarr(store)
    back = zarr_io.read_dataset(store)
    for k in ds.data_vars:
        assert back[k].dims == ds[k].dims
        assert back[k].values.dtype == ds[k].values.dtype
        assert np.array_equal(back[k].values, ds[k].values)
    assert np.array_equal(back.coords["lat"].values, lat) and back.coords["lat"].dims == ("ncells",)
    assert back.attrs["preprocessing_steps"] == ["a", "b"]
    assert json.load(open(os.path.join(store, "extreme_events", ".zarray")))["dtype"] == "|b1"
    assert json.load(open(os.path.join(store, "dat_anomaly", ".zarray")))["chunks"] == [25, Cn]


def test_to_zarr_refuses_to_replace_a_directory_that_is_no_store(tmp_path):
    from marex_amd.exceptions import DataValidationError

    d = tmp_path / "precious"
    d.mkdir()
    (d / "notes.txt").write_text("keep me")
    ds = _MiniDataset({"a": _MiniDataArray(np.zeros((2, 3), np.float32), ("time", "x"))})
    with pytest.raises(DataValidationError, match="not a Zarr store"):
        ds.to_zarr(str(d), mode="w")
    assert (d / "notes.txt").read_text() == "keep me"
    ds.to_zarr(str(tmp_path / "fresh.zarr"))
    ds.to_zarr(str(tmp_path / "fresh.zarr"), mode="w")  # an existing store is replaced
    assert zarr_io.read_array(str(tmp_path / "fresh.zarr" / "a")).shape == (2, 3)


def test_mode_w_minus_never_overwrites(tmp_path):
    """xarray semantics: mode="w-" creates a store and fails if it exists; only mode="w" replaces one."""
    from marex_amd.xr_compat import _MiniDataArray, _MiniDataset

    ds = _MiniDataset({"v": _MiniDataArray(np.arange(6, dtype=np.float32).reshape(2, 3), dims=("a", "b"), coords={})}, attrs={})
    store = str(tmp_path / "s.zarr")
    ds.to_zarr(store, mode="w-")
    first = open(os.path.join(store, "v", ".zarray")).read()
    with pytest.raises(FileExistsError):
        ds.to_zarr(store, mode="w-")
    assert open(os.path.join(store, "v", ".zarray")).read() == first
    ds.to_zarr(store, mode="w")  # replaces
