"""ctypes binding of ``libmarex_hip.so`` (C ABI declared in ``include/marex_hip.h``).

The product path has no CPU fallback: if the shared library is missing or does not export the
declared symbols, loading raises :class:`marex_amd.exceptions.DependencyError`.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .exceptions import DependencyError, ProcessingError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAREX_LIB_PATH") or os.path.join(_HERE, "csrc", "libmarex_hip.so")  # override: experiments

_p = C.c_void_p
_i32 = C.c_int
_i64 = C.c_int64
_u64 = C.c_uint64
_f32 = C.c_float
_f64 = C.c_double

#: name -> (restype, argtypes); must list every function declared in include/marex_hip.h
PROTOTYPES = {
    "marex_abi_version": (_i32, []),
    "marex_workspace_bytes": (_i32, [_p, _p]),
    "marex_create": (_i32, [_i32, C.POINTER(_p)]),
    "marex_destroy": (_i32, [_p]),
    "marex_last_error": (C.c_char_p, [_p]),
    "marex_set_stream": (_i32, [_p, _p]),
    "marex_sync": (_i32, [_p]),
    "marex_timing_enable": (_i32, [_p, _i32]),
    "marex_timing_reset": (_i32, [_p]),
    "marex_timing_get": (_i32, [_p, _i32, C.POINTER(_f64), C.POINTER(_i64)]),
    "marex_synth_sst_f32": (_i32, [_p, _p, _p, _p, _p, _p, _p, _u64, _i64, _i64, _i64, _p]),
    "marex_shifting_baseline_f32": (
        _i32,
        [_p, _p, _i64, _i64, _p, _i32, _i32, _i32, _i32, _p, _i32, _i64, _p, _p, _p, _p],
    ),
    "marex_hobday_thresholds_f32": (
        _i32,
        [_p, _p, _i64, _i64, _i32, _i32, _p, _i32, _p, _p, _i32, _f64, _i32, _i32, _f32, _f32, _i32, _i32, _p, _p],
    ),
    "marex_mask_ge_doy_f32": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "marex_transpose_f32": (_i32, [_p, _p, _i64, _i64, _p]),
    "marex_fixed_baseline_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    "marex_digitize_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _i32, _i64, _p]),
    "marex_detrend_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _i32, _i32, _p, _p, _p]),
    "marex_detrend_deferred_mean_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _i32, _p, _p, _p, _p]),
    "marex_detrend_fixed_baseline_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _p, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p]),
    "marex_fixed_baseline_tails_f32": (_i32, [_p, _p, _p, _i32, _i64, _i64, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "marex_detrend_fixed_baseline_tails_f32": (
        _i32, [_p, _p, _i64, _i64, _p, _p, _p, _i32, _i32, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, _p, _p]),
    "marex_fixed_baseline_sub_f32": (_i32, [_p, _p, _p, _i32, _i64, _i64, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    "marex_hobday_exact_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _i32, _f32, _f64, _i32, _p, _p]),
    "marex_global_threshold_f32": (_i32, [_p, _p, _i64, _i64, _f64, _i32, _p, _p, _i32, _f64, _f64, _p, _p, _p]),
    "marex_mask_ge_const_f32": (_i32, [_p, _p, _p, _i64, _i64, _p, _p]),
    "marex_std_rolling_doy_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _i32, _p, _p]),
    "marex_div_doy_f32": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _p]),
    "marex_fill_holes_u8": (_i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "marex_time_closing_u8": (_i32, [_p, _p, _i64, _i64, _i32, _p]),
    "marex_label2d_i32": (_i32, [_p, _p, _i64, _i32, _i32, _i32, _p, _p]),
    "marex_filter_by_area_u8": (_i32, [_p, _p, _p, _i64, _f64, _i32, _p]),
    "marex_fill_holes_mesh_u8": (_i32, [_p, _p, _p, _p, _i64, _i64, _i32, _p]),
    "marex_label_mesh_i32": (_i32, [_p, _p, _p, _p, _i64, _i64, _p, _p]),
    "marex_validation_summary": (_i32, [_p, _p, _p, _i64, _i64, _p]),
    "marex_mask_ge_doy_bins_f32": (_i32, [_p, _p, _p, _p, _i32, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "marex_set_option": (_i32, [_p, C.c_char_p, _i32]),
    "marex_clear_option": (_i32, [_p, C.c_char_p]),
    "marex_debug_counters": (_i32, [_p, _p, _i32]),
    "marex_tail_lists": (_i32, [_i32, _i32]),
    "marex_tail_extract_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _i32, _p, _i32, _i32, _p, _p]),
    "marex_shifting_baseline_tails_f32": (
        _i32, [_p, _p, _i64, _i64, _p, _i32, _i32, _i32, _p, _i32, _i64, _p, _p, _p, _p, _p, _i32, _p, _p],
    ),
    "marex_hobday_thresholds_tails_f32": (
        _i32,
        [_p, _p, _p, _i32, _p, _i64, _i64, _i32, _i32, _i32, _p, _i32, _f64, _i32, _i32, _f32, _f32, _i32, _i32, _p, _p],
    ),
    "marex_mask_ge_doy_tails_f32": (_i32, [_p, _p, _p, _i32, _i32, _p, _p, _i32, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "marex_blosc_decompress_h": (_i32, [_p, _i64, _p, _i64, _p]),
    "marex_zstd_decompress_h": (_i32, [_p, _i64, _p, _i64, _p]),
    "marex_blosc_compress_h": (_i32, [_p, _i64, _i32, _i32, _i64, _p, _i64, _p]),
    "marex_lz4_decode_streams": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _p, _p]),
    "marex_unshuffle_place": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
}

KERNEL_IDS = {
    "synth": 0,
    "shifting": 1,
    "thresholds": 2,
    "mask": 3,
    "transpose": 4,
    "fixed": 5,
    "detrend": 6,
    "exact": 7,
    "global": 8,
    "stdnorm": 9,
    "morph": 10,
    "tails": 11,
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library and bind every prototype; raises DependencyError when impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DependencyError(
            "HIP extension libmarex_hip.so is not built",
            details=f"expected at {LIB_PATH}",
            suggestions=["run `python -m marex_amd.csrc.build` (needs hipcc, --offload-arch=gfx950)"],
        )
    # torch owns the HBM buffers we are handed, so our kernels must launch through the SAME HIP runtime
    # instance: import torch first so that libamdhip64 (same SONAME) is already mapped when we load.
    import torch  # noqa: F401

    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise DependencyError("HIP extension libmarex_hip.so failed to load", details=str(exc)) from exc
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise DependencyError(f"libmarex_hip.so does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context:
    """One opaque ``marex_ctx`` bound to a device; launches on the stream given to ``set_stream``."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.handle = _p()
        rc = self.lib.marex_create(int(device), C.byref(self.handle))
        if rc != 0:
            raise ProcessingError(f"marex_create(device={device}) failed with code {rc} (no usable HIP device?)")
        self.device = int(device)
        # mirror of the context's option table for host-side switches: MAREX_<NAME>=<int> of the environment at creation
        # (the library seeds itself the same way), then whatever set_option changes
        self.py_opts = {}
        for k, v in os.environ.items():
            if k.startswith("MAREX_") and len(k) > 6:
                try:
                    self.py_opts[k[6:]] = int(v)
                except ValueError:
                    pass

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.marex_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, what: str) -> None:
        if rc != 0:
            msg = self.lib.marex_last_error(self.handle)
            raise ProcessingError(f"{what} failed (code {rc})", details=msg.decode() if msg else None)

    def set_stream(self, stream_ptr: int) -> None:
        self.check(self.lib.marex_set_stream(self.handle, _p(stream_ptr)), "marex_set_stream")

    def sync(self) -> None:
        self.check(self.lib.marex_sync(self.handle), "marex_sync")

    def set_option(self, name: str, value: Optional[int]) -> None:
        """Set (or, with ``None``, clear) a tuning / diagnostic option of this context (include/marex_hip.h)."""
        if value is None:
            self.check(self.lib.marex_clear_option(self.handle, name.encode()), "marex_clear_option")
            self.py_opts.pop(name, None)
        else:
            self.check(self.lib.marex_set_option(self.handle, name.encode(), int(value)), "marex_set_option")
            self.py_opts[name] = int(value)

    def options(self, **opts):
        """Context manager: ``with ctx.options(THR_DD=5, THR_TILE=16): ...`` -- afterwards every option has the value it had
        before (from ``set_option`` or seeded from ``MAREX_<NAME>``); options that were absent are cleared."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            before = {k: self.py_opts.get(k) for k in opts}
            for k, v in opts.items():
                self.set_option(k, v)
            try:
                yield self
            finally:
                for k, old in before.items():
                    self.set_option(k, old)

        return _cm()

    def debug_counters(self, reset: bool = True):
        """Event counters of the tail kernels as a list of 8 ints (synchronises)."""
        buf = (C.c_uint64 * 8)()
        self.check(self.lib.marex_debug_counters(self.handle, buf, int(reset)), "marex_debug_counters")
        return [int(v) for v in buf]

    def timing_enable(self, on: bool = True) -> None:
        self.check(self.lib.marex_timing_enable(self.handle, int(on)), "marex_timing_enable")

    def timing_reset(self) -> None:
        self.check(self.lib.marex_timing_reset(self.handle), "marex_timing_reset")

    def timing_get(self, kernel: str):
        ms, n = _f64(0.0), _i64(0)
        self.check(
            self.lib.marex_timing_get(self.handle, KERNEL_IDS[kernel], C.byref(ms), C.byref(n)), "marex_timing_get"
        )
        return ms.value, n.value
