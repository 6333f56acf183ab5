"""ctypes binding of libipm_hip.so (C ABI declared in include/ipm_hip.h).

The HIP library IS the product path: there is no CPU fallback.  ``load()`` raises
``IpmLibraryError`` when the shared object has not been built (run
``python -c "import __graft_entry__ as g; g.build()"``) and every compute entry
point raises ``IpmError`` with the library's message on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libipm_hip.so")

# symbols declared in include/ipm_hip.h (checked by tests/test_abi.py against the header)
EXPORTS = [
    "ipm_abi_version", "ipm_device_count", "ipm_default_options", "ipm_workspace_bytes", "ipm_workspace_bytes_csc", "ipm_workspace_bytes_opts",
    "ipm_create", "ipm_destroy", "ipm_last_error", "ipm_set_A_dense", "ipm_set_A_csc",
    "ipm_set_bc", "ipm_set_state", "ipm_get_state", "ipm_init_state", "ipm_newton_direction",
    "ipm_iterate", "ipm_solve", "ipm_solve_batch", "ipm_batch_create", "ipm_batch_destroy", "ipm_batch_last_error", "ipm_batch_add", "ipm_batch_step", "ipm_batch_stats", "ipm_get_history", "ipm_get_schedule", "ipm_order_rows", "ipm_get_factor_info", "ipm_solve_linear", "ipm_normal_solve", "ipm_form_normal_matrix", "ipm_get_factor",
    "ipm_set_profiling", "ipm_get_phase_ms", "ipm_debug_get_stamps", "ipm_debug_ff_schedule", "ipm_debug_ff_trace", "ipm_debug_get_block_inverse", "ipm_debug_ls_merge",
]

IPM_OK = 0
STATUS_RUNNING, STATUS_CONVERGED, STATUS_MAX_ITER, STATUS_NAN = 0, 1, 2, 3
FLAG_NO_DEVICE_POLLING = 1      # include/ipm_hip.h: IPM_FLAG_NO_DEVICE_POLLING
FLAG_NO_AUTO_REGULARIZE = 2     # include/ipm_hip.h: IPM_FLAG_NO_AUTO_REGULARIZE
FLAG_SINGLE_STREAM = 4          # include/ipm_hip.h: IPM_FLAG_SINGLE_STREAM
FLAG_SPARSE_FACTOR = 8          # include/ipm_hip.h: IPM_FLAG_SPARSE_FACTOR
FLAG_LOCKSTEP = 16              # include/ipm_hip.h: IPM_FLAG_LOCKSTEP
ERR_WORKSPACE = -4
ABI_VERSION = 4
HISTORY_CAPACITY = 1024         # IPM_HISTORY_CAPACITY
ERR_INVALID_INPUT = -6


class IpmLibraryError(RuntimeError):
    """libipm_hip.so is missing or does not export the ABI."""


class IpmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libipm_hip status %d: %s" % (code, msg))
        self.code = code


class Options(C.Structure):
    _fields_ = [("eta", C.c_double), ("pivot_guard_eps", C.c_double), ("pivot_guard_big", C.c_double),
                ("check_every", C.c_int32), ("flags", C.c_int32), ("sparse_nnz", C.c_int64),
                ("regularize", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_int32), ("pivots_fixed", C.c_int32),
                ("auto_regularized", C.c_int32), ("objective", C.c_double), ("rp_norm", C.c_double),
                ("rd_norm", C.c_double), ("gap", C.c_double), ("b_norm", C.c_double),
                ("c_norm", C.c_double), ("mu", C.c_double), ("mu_aff", C.c_double),
                ("sigma", C.c_double), ("alpha_aff_p", C.c_double), ("alpha_aff_d", C.c_double),
                ("alpha_p", C.c_double), ("alpha_d", C.c_double), ("solve_ms", C.c_double),
                ("objective_last_finite", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class IterRecord(C.Structure):
    """ipm_iter_record: one per completed iteration (the line the reference prints, main.py:808-809, :1186)."""
    _fields_ = [("k", C.c_int32), ("pivots_fixed", C.c_int32), ("objective", C.c_double), ("rp_norm", C.c_double),
                ("rd_norm", C.c_double), ("gap", C.c_double), ("mu", C.c_double), ("sigma", C.c_double),
                ("alpha_aff_p", C.c_double), ("alpha_aff_d", C.c_double), ("alpha_p", C.c_double),
                ("alpha_d", C.c_double)]


_lib = None


def load():
    """Load the shared library once; fail loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64/libhsa-runtime64 and preloads them by path.  Two HSA
    # runtimes cannot share one process (the second sees no device), so when torch is
    # installed it is imported FIRST: libipm_hip.so's NEEDED libamdhip64.so.7 then resolves to
    # the copy torch already mapped.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    if not os.path.exists(LIB_PATH):
        raise IpmLibraryError(
            "%s not found: the HIP extension has not been built. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise IpmLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise IpmLibraryError("%s lacks symbols %s" % (LIB_PATH, missing))
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd = C.POINTER(C.c_double)
    lib.ipm_abi_version.restype = C.c_int
    lib.ipm_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.ipm_default_options.argtypes = [C.POINTER(Options)]
    lib.ipm_default_options.restype = None
    lib.ipm_workspace_bytes.argtypes = [i64, i64, C.POINTER(C.c_size_t)]
    lib.ipm_workspace_bytes_csc.argtypes = [i64, i64, i64, C.POINTER(C.c_size_t)]
    lib.ipm_workspace_bytes_opts.argtypes = [i64, i64, C.POINTER(Options), C.POINTER(C.c_size_t)]
    lib.ipm_create.argtypes = [C.c_int, i64, i64, C.POINTER(Options), vp, C.c_size_t, vp, C.POINTER(vp)]
    lib.ipm_destroy.argtypes = [vp]
    lib.ipm_last_error.argtypes = [vp]
    lib.ipm_last_error.restype = C.c_char_p
    lib.ipm_set_A_dense.argtypes = [vp, vp, i64, C.c_int]
    lib.ipm_set_A_csc.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), pd, i64]
    lib.ipm_set_bc.argtypes = [vp, pd, pd]
    lib.ipm_set_state.argtypes = [vp, pd, pd, pd]
    lib.ipm_get_state.argtypes = [vp, pd, pd, pd]
    lib.ipm_init_state.argtypes = [vp, dbl]
    lib.ipm_newton_direction.argtypes = [vp, C.c_int, pd, pd, pd, C.POINTER(Stats)]
    lib.ipm_iterate.argtypes = [vp, i32, C.POINTER(Stats)]
    lib.ipm_solve.argtypes = [vp, dbl, dbl, dbl, i32, C.POINTER(Stats)]
    lib.ipm_solve_batch.argtypes = [C.POINTER(vp), i32, dbl, dbl, dbl, i32, C.POINTER(Stats)]
    lib.ipm_batch_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.ipm_batch_destroy.argtypes = [vp]
    lib.ipm_batch_last_error.argtypes = [vp]
    lib.ipm_batch_last_error.restype = C.c_char_p
    lib.ipm_batch_add.argtypes = [vp, vp, dbl, dbl, dbl, i32, C.POINTER(i32)]
    lib.ipm_batch_step.argtypes = [vp, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(i32)]
    lib.ipm_batch_stats.argtypes = [vp, i32, C.POINTER(Stats)]
    lib.ipm_get_history.argtypes = [vp, C.POINTER(IterRecord), i32, C.POINTER(i32)]
    lib.ipm_get_schedule.argtypes = [vp, C.POINTER(i32)]
    lib.ipm_order_rows.argtypes = [i64, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), pd]
    lib.ipm_get_factor_info.argtypes = [vp, C.POINTER(i64)]
    lib.ipm_solve_linear.argtypes = [vp, pd, i64, pd, pd, C.POINTER(i32)]
    lib.ipm_normal_solve.argtypes = [vp, pd, pd, pd, C.c_int, C.POINTER(i32)]
    lib.ipm_form_normal_matrix.argtypes = [vp, pd, pd, i64]
    lib.ipm_get_factor.argtypes = [vp, pd, i64]
    lib.ipm_set_profiling.argtypes = [vp, C.c_int]
    lib.ipm_get_phase_ms.argtypes = [vp, pd]
    lib.ipm_debug_get_stamps.argtypes = [vp, C.POINTER(C.c_longlong)]
    lib.ipm_debug_ff_schedule.argtypes = [i32, i32, i32, C.POINTER(C.c_ubyte), i32, C.POINTER(i32), C.POINTER(i32), pd]
    lib.ipm_debug_get_block_inverse.argtypes = [vp, i32, pd]
    lib.ipm_debug_ff_trace.argtypes = [vp, C.POINTER(C.c_longlong), i64, C.POINTER(i64), C.POINTER(C.c_ubyte), C.POINTER(i32)]
    lib.ipm_debug_ls_merge.argtypes = [i32, C.POINTER(i32), C.POINTER(i32), i32, i32, C.POINTER(i32), i32, C.POINTER(i32)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("ipm_default_options", "ipm_last_error", "ipm_batch_last_error"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(handle, code):
    if code != IPM_OK:
        msg = load().ipm_last_error(handle)
        raise IpmError(code, (msg or b"").decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    load().ipm_device_count(C.byref(n))
    return n.value
