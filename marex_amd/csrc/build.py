"""Build libmarex_hip.so (gfx950) in-tree:  python -m marex_amd.csrc.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = [os.path.join(HERE, "marex_hip.hip")]
OUT = os.path.join(HERE, "libmarex_hip.so")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",  # arithmetic contract: no fused multiply-add anywhere
    "-fno-fast-math", "-Wall", "-Wno-unused-result",
    "-I", os.path.join(ROOT, "include"),
]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = SRC + [os.path.join(ROOT, "include", "marex_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if force or needs_build():
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, *FLAGS, *SRC, "-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
