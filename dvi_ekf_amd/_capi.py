"""ctypes binding of libdvi_eskf.so (include/dvi_eskf.h).

The shared library is the product: if it is missing or does not export every
entry point, importing this module's ``lib()`` raises -- there is no CPU or
PyTorch fallback for the filter arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdvi_eskf.so")
# Experiment builds (tools/ab_lib.sh, tools/prof_variant.sh) are selected per process with DVI_ESKF_LIB=<path> instead
# of being copied over the committed library: an interrupted A/B run then leaves nothing behind.  lib_path() is what
# bench.py reports, so that a line measured with another build says so.
LIB_OVERRIDE_ENV = "DVI_ESKF_LIB"


def lib_path() -> str:
    return os.environ.get(LIB_OVERRIDE_ENV) or LIB_PATH

F64, F32 = 0, 1
FIX_Q1, FIX_Q3, FIX_Q6, FIX_Q9 = 1, 2, 4, 8      # dvi_eskf_params.quirks_fixed
STATUS_SINGULAR, STATUS_NONFINITE = 1, 2


class Params(C.Structure):
    """dvi_eskf_params"""
    _fields_ = [
        ("L", C.c_double), ("theta_cam", C.c_double), ("frozen", C.c_int32 * 6), ("hold_frozen", C.c_int32),
        ("sigma_a", C.c_double * 3), ("sigma_om", C.c_double * 3), ("quirks_fixed", C.c_int32),
    ]


class Schedule(C.Structure):
    """dvi_eskf_schedule"""
    _fields_ = [
        ("n_epochs", C.c_int64), ("n_pred", C.POINTER(C.c_int32)),
        ("dt", C.c_void_p), ("dt_per_instance", C.c_int),
        ("om", C.c_void_p), ("acc", C.c_void_p), ("imu_per_instance", C.c_int),
        ("pos", C.c_void_p), ("quat_xyzw", C.c_void_p), ("notch", C.c_void_p), ("meas_per_instance", C.c_int),
        ("status", C.c_void_p), ("t_log", C.POINTER(C.c_double)), ("rows_out", C.c_void_p),
        ("t_pred", C.POINTER(C.c_double)), ("rows_pred_out", C.c_void_p), ("rows_instances", C.c_int64),
    ]


_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
_dp = C.POINTER(C.c_double)
_hpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/dvi_eskf.h one to one
SIGNATURES = {
    "dvi_eskf_create": (_i, [C.POINTER(Params), _i64, _i, _i, _hpp]),
    "dvi_eskf_clone": (_i, [_vp, _vp, _hpp]),
    "dvi_eskf_copy_state": (_i, [_vp, _vp, _vp]),
    "dvi_eskf_copy_instances": (_i, [_vp, _vp, _i64, _vp]),
    "dvi_eskf_destroy": (None, [_vp]),
    "dvi_eskf_batch": (_i64, [_vp]),
    "dvi_eskf_dtype": (_i, [_vp]),
    "dvi_eskf_set_noise": (_i, [_vp, _dp, _dp]),
    "dvi_eskf_set_noise_per_instance": (_i, [_vp, _dp, _dp, _dp, _vp]),
    "dvi_eskf_set_state": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dvi_eskf_get_state": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dvi_eskf_predict": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp]),
    "dvi_eskf_update": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "dvi_eskf_step": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "dvi_eskf_epoch": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "dvi_eskf_step_from": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "dvi_eskf_run_epochs": (_i, [_vp, C.POINTER(Schedule), _vp]),
    "dvi_eskf_run_epochs_from": (_i, [_vp, _vp, C.POINTER(Schedule), _vp]),
    "dvi_eskf_get_calibration": (_i, [_vp, _vp, _vp]),
    "dvi_eskf_gather_calibration": (_i, [_vp, _vp, _vp, _vp]),
    "dvi_eskf_flush": (_i, [_vp, _vp]),
    "dvi_eskf_reserve": (_i, [_vp, _i]),
    "dvi_eskf_keep_jacobians": (_i, [_vp, _i]),
    "dvi_eskf_get_jacobians": (_i, [_vp, _vp, _vp, _vp]),
    "dvi_eskf_raw_P": (_vp, [_vp]),
    "dvi_eskf_raw_x": (_vp, [_vp]),
    "dvi_eskf_set_lanes_per_instance": (_i, [_vp, _i]),
    "dvi_eskf_log_rows": (_i, [_vp, C.c_double, _vp, _vp]),
    "dvi_eskf_update_mse": (_i, [_vp, _dp, _dp, _vp, _vp]),
    "dvi_eskf_imu_synthesize": (_i, [C.POINTER(Params), _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _i, _dp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp]),
    "dvi_eskf_randn": (_i, [_i, _i, C.c_uint64, _i64, _i64, _i64, _i64, _i, _vp, _vp]),
    "dvi_eskf_strerror": (C.c_char_p, [_i]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP library is required (build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C dvi_ekf_amd/csrc`)")
        # PyTorch must be loaded first: the library has to bind to the HIP runtime that
        # PyTorch ships, otherwise torch's streams / device pointers belong to another runtime
        import torch  # noqa: F401
        if os.environ.get(LIB_OVERRIDE_ENV):
            import sys
            print(f"dvi_ekf_amd: {LIB_OVERRIDE_ENV} selects {path} instead of the package's own library", file=sys.stderr)
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class EskfError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != 0:
        raise EskfError(f"{what}: {lib().dvi_eskf_strerror(code).decode()} ({code})")
