"""CPU: the Zarr v2 / Blosc-LZ4 reader (host side of the C ABI) on the data files of the reference's own tests."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from marex_amd import _lib, zarr_io

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
STATS = json.load(open(os.path.join(FIX, "decoded_stats.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_fixture_copies_decode_like_the_originals():
    sst = zarr_io.read_array(os.path.join(FIX, "sst_gridded.zarr", "to"))
    tm = zarr_io.read_array(os.path.join(FIX, "sst_gridded.zarr", "time"))
    ev = zarr_io.read_array(os.path.join(FIX, "extremes_gridded.zarr", "extreme_events"))
    mk = zarr_io.read_array(os.path.join(FIX, "extremes_gridded.zarr", "mask"))
    for key, arr in (("sst_to", sst), ("sst_time", tm), ("extreme_events", ev), ("mask", mk)):
        assert list(arr.shape) == STATS[key]["shape"] and str(arr.dtype) == STATS[key]["dtype"]
        assert _sha(arr) == STATS[key]["sha256"], key
    un = os.path.join(FIX, "sst_unstructured.zarr")
    for key, arr in (("sst_unstructured_to", zarr_io.read_array(os.path.join(un, "to"))),
                     ("sst_unstructured_time", zarr_io.read_array(os.path.join(un, "time"))),
                     ("unstructured_extreme_events", zarr_io.read_array(os.path.join(FIX, "extremes_unstructured.zarr", "extreme_events"))),
                     ("unstructured_mask", zarr_io.read_array(os.path.join(FIX, "extremes_unstructured.zarr", "mask"))),
                     ("unstructured_neighbours", zarr_io.read_array(os.path.join(FIX, "extremes_unstructured.zarr", "neighbours")))):
        assert list(arr.shape) == STATS[key]["shape"] and _sha(arr) == STATS[key]["sha256"], key
    # what the data is: 40 years (14 611 days) of daily SST in kelvin on a 20 x 40 patch, CF time axis
    t = zarr_io.decode_cf_time(tm, zarr_io.array_attrs(os.path.join(FIX, "sst_gridded.zarr", "time")))
    assert str(t[0]) == "1982-01-01T12:00:00" and np.all(np.diff(t).astype("timedelta64[s]").astype(int) == 86400)
    assert sst.dtype == np.float32 and 280 < sst.min() < sst.max() < 310 and np.isfinite(sst).all()
    assert set(np.unique(ev)) <= {0, 1} and set(np.unique(mk)) <= {0, 1}


def test_decoder_rejects_damaged_and_foreign_frames():
    lib = _lib.load()
    f = os.path.join(FIX, "sst_gridded.zarr", "to", "0.0.0")
    raw = open(f, "rb").read()
    n = C.c_int64(0)
    out = C.create_string_buffer(30 * 20 * 40 * 4)
    assert lib.marex_blosc_decompress_h(raw, len(raw), out, len(out), C.byref(n)) == 0 and n.value == len(out)
    good = bytes(out.raw)
    assert lib.marex_blosc_decompress_h(raw, len(raw) // 2, out, len(out), C.byref(n)) != 0        # truncated
    assert lib.marex_blosc_decompress_h(raw, len(raw), out, 100, C.byref(n)) != 0                  # destination too small
    zstd = bytearray(raw)
    zstd[2] = (zstd[2] & 0x1F) | (4 << 5)                                                           # codec id: zstd
    assert lib.marex_blosc_decompress_h(bytes(zstd), len(zstd), out, len(out), C.byref(n)) == -6
    bad = bytearray(raw)
    bad[200:260] = b"\\xff" * 60                                                                     # garbage inside a stream
    rc = lib.marex_blosc_decompress_h(bytes(bad), len(bad), out, len(out), C.byref(n))
    assert rc != 0 or bytes(out.raw) != good
    with pytest.raises(Exception):
        zarr_io.read_array(os.path.join(FIX, "sst_gridded.zarr", "nothing_here"))


def test_zstd_bitshuffle_coordinate_arrays():
    """The lat / lon arrays of the reference's gridded SST store ({"cname": "zstd", "shuffle": 2}) decode to its regular
    0.25-degree axes, and the whole store opens as a Dataset."""
    pytest.importorskip("pyarrow")
    lat = zarr_io.read_array(os.path.join(FIX, "sst_gridded.zarr", "lat"))
    lon = zarr_io.read_array(os.path.join(FIX, "sst_gridded.zarr", "lon"))
    assert lat.dtype == np.float32 and lat.shape == (20,) and lon.shape == (40,)
    assert np.array_equal(lat, np.float32(35.125) + np.float32(0.25) * np.arange(20, dtype=np.float32))
    assert np.array_equal(lon, np.float32(-39.875) + np.float32(0.25) * np.arange(40, dtype=np.float32))
    ds = zarr_io.read_dataset(os.path.join(FIX, "sst_gridded.zarr"))
    assert ds.to.dims == ("time", "lat", "lon") and ds.to.shape == (14611, 20, 40)
    assert np.array_equal(ds.to.coords["lat"].values, lat) and str(ds.time.values[0])[:10] == "1982-01-01"


def test_device_read_plan_validates_every_offset_on_the_host():
    """the device decoder bounds its reads only by the stream sizes it is handed, so `plan_blosc_frame` checks
    header, block table and every stream against the frame BEFORE anything is uploaded (no GPU launch in this test)."""
    import struct

    from marex_amd.exceptions import DataValidationError

    f = os.path.join(FIX, "sst_gridded.zarr", "to", "0.0.0")
    raw = open(f, "rb").read()
    nbytes = 30 * 20 * 40 * 4
    shuffled, streams, blocks = zarr_io.plan_blosc_frame(raw, 4, nbytes, f)
    assert shuffled and sum(b[3] for b in blocks) == nbytes and sum(s[3] for s in streams) == nbytes
    assert all(16 <= p and p + cb <= len(raw) and 0 <= cb <= rawsz for p, cb, _, rawsz in streams)

    def rejects(frame, expected=nbytes):
        with pytest.raises(DataValidationError):
            zarr_io.plan_blosc_frame(bytes(frame), 4, expected, "damaged")

    rejects(raw[:10])                                     # shorter than a header
    rejects(raw[: len(raw) // 2])                         # truncated: cbytes != file size
    rejects(raw, expected=nbytes + 4)                     # nbytes does not match the chunk shape
    hdr = bytearray(raw)
    hdr[12:16] = struct.pack("<I", len(raw) // 2)         # cbytes smaller than the file
    rejects(hdr)
    far = bytearray(raw)
    far[16:20] = struct.pack("<i", len(raw) + 64)         # first block starts past the end of the frame
    rejects(far)
    neg = bytearray(raw)
    neg[16:20] = struct.pack("<i", -8)                    # negative block start
    rejects(neg)
    p0 = struct.unpack("<i", raw[16:20])[0]
    big = bytearray(raw)
    big[p0:p0 + 4] = struct.pack("<i", 1 << 28)           # stream claims more compressed bytes than the frame holds
    rejects(big)
    minus = bytearray(raw)
    minus[p0:p0 + 4] = struct.pack("<i", -1)              # negative stream size
    rejects(minus)
    tiny_block = bytearray(raw)
    tiny_block[8:12] = struct.pack("<I", 0)               # blocksize 0
    rejects(tiny_block)
