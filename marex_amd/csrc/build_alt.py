"""Alternative builds of libmarex_hip for experiments (never loaded by default):
    python marex_amd/csrc/build_alt.py <name> [extra hipcc flags...]   ->  marex_amd/csrc/alt/libmarex_hip_<name>.so
Use with MAREX_LIB_PATH=marex_amd/csrc/alt/libmarex_hip_<name>.so (e.g. -DMAREX_ABLATION: timing-only ablation bits,
-DMAREX_STAMPS: in-kernel phase timers of the tail threshold kernel in the debug counters)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build as b  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
outdir = os.path.join(b.HERE, "alt")
os.makedirs(outdir, exist_ok=True)


def one(src):
    obj = os.path.join(outdir, os.path.basename(src)[:-4] + f"_{name}.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *b.FLAGS, *extra, "-c", src, "-o", obj])
    return obj


with ThreadPoolExecutor(max_workers=8) as pool:
    objs = list(pool.map(one, b.SRC))
out = os.path.join(outdir, f"libmarex_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
print(out)
