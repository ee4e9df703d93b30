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
