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
    assert lib.marex_abi_version() == 1


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
