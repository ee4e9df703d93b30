"""CPU: the Zstandard decoder written from the format description (csrc/marex_zstd.hip, host side) against an independent
implementation (pyarrow's bundled libzstd as ENCODER and reference decoder): literal and sequence modes of every compression
level family, raw / RLE blocks, multi-block and multi-frame streams, skippable frames -- and malformed input (truncations, bit
flips), which must come back as an error code or as different bytes, never as a crash.  (The sanitizer run of the host build
lives in tests/test_zstd_sanitizers.py, a CPU-only file that does not travel to the GPU box.)"""
import ctypes as C

import numpy as np
import pytest

from marex_amd import _lib

pa = pytest.importorskip("pyarrow")


def _dec(lib, b: bytes, cap: int):
    out = C.create_string_buffer(max(cap, 1))
    got = C.c_int64(0)
    rc = lib.marex_zstd_decompress_h(b, len(b), out, cap, C.byref(got))
    return rc, out.raw[: got.value]


def _cases():
    rng = np.random.default_rng(7)
    text = ("The quick brown fox jumps over the lazy dog. " * 40 + "".join(chr(97 + int(c)) for c in rng.integers(0, 26, 2500))).encode()
    return {
        "empty": b"",
        "one byte": b"a",
        "short repeat": b"hello hello hello hello hello hello",
        "zeros (RLE block)": bytes(1000),
        "zeros, several blocks": bytes(300000),
        "noise (raw block)": rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(),
        "2-bit noise (Huffman literals, four streams)": rng.integers(0, 4, 200000, dtype=np.uint8).tobytes(),
        "noise, several blocks": rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(),
        "period 251": (np.arange(100000) % 251).astype(np.uint8).tobytes(),
        "float32 sine": np.sin(np.arange(60000) / 50.0).astype(np.float32).tobytes(),
        "float64 random walk": np.cumsum(rng.normal(0, 1, 60000)).tobytes(),
        "text": text * 15,
        "regular axis (the fixture's kind of array)": np.arange(-89.875, 90, 0.25).tobytes(),
    }


@pytest.mark.parametrize("level", [1, 3, 9, 19])
def test_decoder_matches_libzstd(level):
    lib = _lib.load()
    codec = pa.Codec("zstd", compression_level=level)
    for name, raw in _cases().items():
        comp = codec.compress(raw, asbytes=True)
        rc, out = _dec(lib, comp, len(raw))
        assert rc == 0 and out == raw, (name, level, rc, len(out), len(raw))


def test_concatenated_and_skippable_frames_and_small_output_buffer():
    lib = _lib.load()
    codec = pa.Codec("zstd", compression_level=3)
    a, b = b"first frame " * 100, bytes(range(256)) * 40
    skippable = (0x184D2A53).to_bytes(4, "little") + (5).to_bytes(4, "little") + b"\x01\x02\x03\x04\x05"
    stream = codec.compress(a, asbytes=True) + skippable + codec.compress(b, asbytes=True)
    rc, out = _dec(lib, stream, len(a) + len(b))
    assert rc == 0 and out == a + b
    rc, _ = _dec(lib, stream, len(a) + len(b) - 1)   # output buffer one byte short: refused, nothing written past it
    assert rc == -5
    rc, _ = _dec(lib, b"\x00\x01\x02\x03junk", 100)   # not a zstd stream
    assert rc == -5


def test_malformed_streams_never_crash():
    lib = _lib.load()
    rng = np.random.default_rng(11)
    codec = pa.Codec("zstd", compression_level=9)
    raws = [v for v in _cases().values() if 1000 <= len(v) <= 100000]
    n_err = 0
    for raw in raws:
        comp = bytearray(codec.compress(raw, asbytes=True))
        for _ in range(60):
            bad = bytearray(comp)
            kind = rng.integers(0, 3)
            if kind == 0:
                bad = bad[: int(rng.integers(0, len(bad)))]
            elif kind == 1:
                for _k in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            else:
                i = int(rng.integers(0, len(bad)))
                bad[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            rc, out = _dec(lib, bytes(bad), len(raw))
            assert rc in (0, -5)
            n_err += rc != 0
    assert n_err > 0

