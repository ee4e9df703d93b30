"""Build libmarex_hip.so (gfx950) in-tree:  python -m marex_amd.csrc.build

One object per translation unit (compiled in parallel), linked into one shared library; no relocatable device code is
needed because every kernel lives in the unit that launches it and the shared helpers are inline (marex_common.hip.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
UNITS = ["marex_context", "marex_synth", "marex_shifting", "marex_thresholds", "marex_mask", "marex_anomalies",
         "marex_quantiles", "marex_morphology", "marex_blosc", "marex_tails", "marex_zstd"]
SRC = [os.path.join(HERE, u + ".hip") for u in UNITS]
HEADERS = [os.path.join(HERE, "marex_common.hip.h"), os.path.join(HERE, "marex_tails.hip.h"), os.path.join(ROOT, "include", "marex_hip.h")]
OUT = os.path.join(HERE, "libmarex_hip.so")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",  # arithmetic contract: no fused multiply-add anywhere
    "-fno-fast-math", "-Wall", "-Wno-unused-result", "-Wno-unused-function",
    "-I", os.path.join(ROOT, "include"),
]


def _obj(src: str) -> str:
    return src[:-4] + ".o"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(OUT, SRC + HEADERS + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("MAREX_HIPCC_FLAGS", "").split()
    todo = [s for s in SRC if force or extra or _stale(_obj(s), [s] + HEADERS + [os.path.abspath(__file__)])]

    def compile_one(src):
        cmd = [hipcc, *FLAGS, *extra, "-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as pool:
            list(pool.map(compile_one, todo))
    if todo or force or needs_build():
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s) for s in SRC], "-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
