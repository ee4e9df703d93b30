"""CPU only (listed in .gpurunignore: the GPU pool refuses sanitizer builds, and this one never touches a GPU): the HOST build
of the Zstandard decoder under AddressSanitizer + UBSan."""
import numpy as np
import pytest

from tests.test_zstd_decoder import _cases

pa = pytest.importorskip("pyarrow")


def test_decoder_under_address_and_ub_sanitizers(tmp_path):
    """The host build of the decoder with -fsanitize=address,undefined over a corpus of valid, truncated, bit-flipped and
    byte-stuffed streams, each in exact-size heap buffers (tests/host/zstd_sanitizer_harness.cpp): no report, no crash."""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "zasan")
    subprocess.check_call([hipcc, "--cuda-host-only", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1", "-std=c++17",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "marex_amd", "csrc", "marex_zstd.hip"),
                           os.path.join(root, "tests", "host", "zstd_sanitizer_harness.cpp"), "-o", exe],
                          cwd=os.path.join(root, "marex_amd", "csrc"), stderr=subprocess.DEVNULL)
    rng = np.random.default_rng(5)
    k = 0
    for lvl in (1, 19):
        codec = pa.Codec("zstd", compression_level=lvl)
        for raw in _cases().values():
            comp = bytearray(codec.compress(raw, asbytes=True))
            for j in range(24):
                bad = bytearray(comp)
                if j % 4 == 1:
                    bad = bad[: int(rng.integers(0, len(bad) + 1))]
                elif j % 4 == 2:
                    for _ in range(int(rng.integers(1, 5))):
                        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
                elif j % 4 == 3:
                    i = int(rng.integers(0, len(bad) + 1))
                    bad[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
                (tmp_path / f"c{k:05d}_{len(raw)}.bin").write_bytes(bytes(bad))
                k += 1
    out = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
    assert out.returncode == 0 and "streams" in out.stdout and "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-2000:]
