"""CPU: the packed sorting networks of marex_amd/csrc/marex_tails.hip.h are plain host + device code; the host build is
checked exhaustively (0-1 principle over all 2^16 inputs for the 16-key network) and on random packed keys."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def test_sorting_networks_on_the_host(tmp_path):
    exe = str(tmp_path / "tail_networks_check")
    subprocess.check_call([CLANG, "-O2", "-std=c++17", os.path.join(ROOT, "tests", "host", "tail_networks_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "OK", out.stdout + out.stderr
