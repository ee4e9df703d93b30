"""The C-ABI library loads and exports exactly what include/marex_hip.h declares (no GPU needed)."""
import os
import re

from marex_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "marex_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(marex_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    decl = declared_functions()
    assert "marex_shifting_baseline_f32" in decl and "marex_hobday_thresholds_f32" in decl
    assert sorted(_lib.PROTOTYPES) == decl


def test_library_exports_every_declared_symbol():
    from marex_amd.csrc import build

    build.build(verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.marex_abi_version() == 2


def test_bad_context_is_an_error_code_not_a_crash():
    lib = _lib.load()
    assert lib.marex_sync(None) != 0
    assert lib.marex_destroy(None) != 0


def test_host_modules_import_without_gpu():
    """engine / detect / dist import cleanly on a CPU-only box (the GPU is touched at first compute call)."""
    import marex_amd
    import marex_amd.detect
    import marex_amd.dist
    import marex_amd.engine  # noqa: F401

    assert callable(marex_amd.preprocess_data)


def _header_signatures():
    """name -> list of ctypes classes parsed from the C declarations in include/marex_hip.h."""
    import ctypes as C

    src = open(os.path.join(ROOT, "include", "marex_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(marex_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        kinds = []
        if params and params != "void":
            for prm in params.split(","):
                prm = prm.strip()
                if "*" in prm:
                    kinds.append("ptr")
                elif re.match(r"(u?int64_t)\b", prm):
                    kinds.append("u64" if prm.startswith("uint64_t") else "i64")
                elif prm.startswith("int "):
                    kinds.append("i32")
                elif prm.startswith("float "):
                    kinds.append("f32")
                elif prm.startswith("double "):
                    kinds.append("f64")
                else:
                    raise AssertionError(f"unparsed parameter {prm!r} of {name}")
        sigs[name] = kinds
    return sigs


def test_ctypes_prototypes_match_the_header_parameter_lists():
    """A wrong argtypes list sends garbage pointers to a kernel (GPU fault): check arity AND kinds on the CPU."""
    import ctypes as C

    kind_of = {C.c_void_p: "ptr", C.c_int64: "i64", C.c_uint64: "u64", C.c_int: "i32", C.c_float: "f32", C.c_double: "f64"}
    sigs = _header_signatures()
    assert sorted(sigs) == sorted(_lib.PROTOTYPES)
    for name, (_, argtypes) in _lib.PROTOTYPES.items():
        got = ["ptr" if (a not in kind_of) else kind_of[a] for a in argtypes]  # POINTER(...) types count as pointers
        assert got == sigs[name], f"{name}: binding {got} != header {sigs[name]}"


def test_workspace_bytes_matches_what_the_engine_allocates():
    """marex_workspace_bytes: a C caller can size every buffer of a block without reading engine.py -- the numbers
    are the shapes `HotPath` allocates (lists [366][NPER][nch][C][8 x u16], aux [366][C] u32, blocked bin matrix ...)."""
    import ctypes as C

    from marex_amd.engine import HotPath

    class Cfg(C.Structure):
        _fields_ = [("T", C.c_int64), ("T_out", C.c_int64), ("C", C.c_int64), ("max_bucket", C.c_int), ("list_rows", C.c_int)]

    lib = _lib.load()
    out = (C.c_size_t * 8)()
    T, T_out, Cn = 36500, 31022, 124 * 1440
    assert lib.marex_workspace_bytes(C.byref(Cfg(T, T_out, Cn, 85, 15)), out) == 0
    nper = lib.marex_tail_lists(85, 15)
    assert nper == 6
    assert out[0] == T_out * Cn * 4 and out[1] == T_out * Cn and out[2] == 366 * Cn * 4
    assert out[3] == 366 * nper * 2 * Cn * 16 and out[4] == 366 * Cn * 4 and out[5] == 0 and out[6] == Cn * 5
    assert out[7] == out[0] + out[1] + 2 * out[2] + out[3] + out[4] + out[6]
    assert lib.marex_workspace_bytes(C.byref(Cfg(3652, 1826, 1000, 5, 0)), out) == 0     # bin-matrix path
    nblk, rows, w = HotPath.bins_shape(1826, 1000)
    assert out[5] == nblk * rows * w * 2 and out[3] == 0 and out[4] == 0
    assert lib.marex_workspace_bytes(None, out) != 0 and lib.marex_workspace_bytes(C.byref(Cfg(10, 20, 5, 5, 0)), out) != 0
