"""ctypes binding of include/fluid_amd.h (libfluid_amd.so).

There is no fallback: if the HIP library is missing or a call fails, this
module raises.  Nothing here (or anywhere in this package) touches oracle/.
"""
import ctypes as C
import os

import numpy as np

from ._build import LIB

OK, E_INVALID, E_NOMEM, E_HIP, E_COMM = 0, 1, 2, 3, 4
U, V, DENS, U_PREV, V_PREV, DENS_PREV, TMP0, TMP1, TMP2, TMP3, TMP4, TMP5 = range(12)
NFIELDS = 12
JACOBI_STREAM, JACOBI_LDS, JACOBI_NAIVE, JACOBI_TB = 0, 1, 2, 3
STORAGE_F32, STORAGE_F16 = 0, 1
PARAM_TB_MAX_SWEEPS, PARAM_TB_ROWS, PARAM_HALO, PARAM_TB_FAST_DIVISION, PARAM_TB_MIN_CELLS, PARAM_TB_EDGE_ROWS_PCT = 0, 1, 2, 3, 4, 5
PARAM_TB_LANE_COLUMNS = 6
PARAM_TB_T16_MIN_CELLS = 7
PARAM_TB_AUTOTUNE = 8
PARAM_FUSE_DIVERGENCE = 9
PARAM_SLAB_OVERLAP = 10
PARAM_EARLY_ADVECT = 11
PARAM_FUSE_ADD_SOURCE = 12
PARAM_XCHG_OVERLAP = 13
PARAM_F16_PRESSURE_SCALE = 14
XCHG_HALO, XCHG_GATHER, XCHG_MAX, XCHG_MAX_BEGIN, XCHG_MAX_END = 0, 1, 2, 3, 4
RCCL_ID_BYTES = 128
FIELD_NAMES = ("u", "v", "dens", "u_prev", "v_prev", "dens_prev", "tmp0", "tmp1", "tmp2", "tmp3", "tmp4", "tmp5")


class FluidError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfluid_amd error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("n", C.c_int), ("rank", C.c_int), ("nranks", C.c_int), ("halo", C.c_int),
                ("jacobi_variant", C.c_int), ("stream", C.c_void_p), ("arena", C.c_void_p),
                ("arena_bytes", C.c_size_t), ("storage", C.c_int)]


TIME_SOURCE, TIME_DIFFUSION, TIME_DIVERGENCE, TIME_PROJECTION, TIME_ADVECTION = range(5)
TIMING_CATEGORIES = ("source", "diffusion", "divergence", "projection", "advection")


class Timing(C.Structure):
    _fields_ = [("jacobi_ms", C.c_double), ("sweeps", C.c_longlong), ("solves", C.c_longlong),
                ("category_ms", C.c_double * 5), ("category_calls", C.c_longlong * 5),
                ("jacobi_launches", C.c_longlong), ("jacobi_field_launches", C.c_longlong),
                ("pressure_ms", C.c_double), ("pressure_sweeps", C.c_longlong)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                          C.POINTER(C.c_float))

_HOSTF = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_ctx = C.c_void_p
_i, _f = C.c_int, C.c_float

# name -> argtypes; every listed function returns int status.  This table is
# also what tests/test_abi.py checks against include/fluid_amd.h.
SIGNATURES = {
    "step": [_i, _f, _f, _f, _HOSTF, _HOSTF, _HOSTF],
    "step_src": [_i, _f, _f, _f, _i] + [_HOSTF] * 6,
    "fluid_release_cached": [],
    "fluid_coefficients": [_i, _f, _f, C.POINTER(_f), C.POINTER(_f)],
    "fluid_layout": [_i, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_size_t)],
    "fluid_create": [_i, C.POINTER(_ctx)],
    "fluid_create_ex": [C.POINTER(Config), C.POINTER(_ctx)],
    "fluid_destroy": [_ctx],
    "fluid_synchronize": [_ctx],
    "fluid_owned_rows": [_ctx, C.POINTER(_i), C.POINTER(_i)],
    "fluid_field_ptr": [_ctx, _i, C.POINTER(C.c_void_p)],
    "fluid_scalar_ptr": [_ctx, C.POINTER(C.c_void_p)],
    "fluid_upload": [_ctx, _i, _HOSTF],
    "fluid_download": [_ctx, _i, _HOSTF],
    "fluid_upload_rows": [_ctx, _i, _HOSTF, _i, _i],
    "fluid_download_rows": [_ctx, _i, _HOSTF, _i, _i],
    "fluid_fill": [_ctx, _i, _f],
    "fluid_step": [_ctx, _f, _f, _f, _i, _i, _i],
    "fluid_vel_step": [_ctx, _f, _f, _i],
    "fluid_dens_step": [_ctx, _f, _f, _i],
    "fluid_op_set_bnd": [_ctx, _i, _i],
    "fluid_op_add_source": [_ctx, _i, _i, _f],
    "fluid_op_jacobi_sweep": [_ctx, _i, _i, _i, _i, _f, _f],
    "fluid_op_diffuse": [_ctx, _i, _i, _i, _f, _f, _i],
    "fluid_op_advect": [_ctx, _i, _i, _i, _i, _i, _f],
    "fluid_op_divergence": [_ctx, _i, _i, _i, _i],
    "fluid_op_subtract_gradient": [_ctx, _i, _i, _i],
    "fluid_residual": [_ctx, _i, _i, _f, _f, C.POINTER(_f)],
    "fluid_absmax_velocity": [_ctx, _i, _i, C.POINTER(_f)],
    "fluid_set_jacobi_variant": [_ctx, _i],
    "fluid_division_mode": [_ctx, _f, _f, C.POINTER(_i)],
    "fluid_autotune_pending": [_ctx, C.POINTER(_i)],
    "fluid_plan_sweeps": [_i, _i, _i, _i, _i, _i, _i, C.POINTER(_i), _i, C.POINTER(_i)],
    "fluid_set_param": [_ctx, _i, _i],
    "fluid_timing_enable": [_ctx, _i],
    "fluid_timing_read": [_ctx, C.POINTER(Timing), _i],
    "fluid_op_diffuse_tol": [_ctx, _i, _i, _i, _f, _f, _f, _i, _i, C.POINTER(_i), C.POINTER(_f)],
    "fluid_set_exchange": [_ctx, EXCHANGE_FN, C.c_void_p],
    "fluid_exchange_now": [_ctx, _i, C.POINTER(_i), _i, _i],
    "fluid_exchange_stream": [_ctx, C.POINTER(C.c_void_p)],
    "fluid_split_launches": [_ctx, C.POINTER(C.c_longlong)],
    "fluid_rccl_available": [],
    "fluid_rccl_unique_id": [C.c_void_p, C.c_size_t],
    "fluid_exchange_rccl_attach": [_ctx, C.c_void_p, C.c_size_t],
    "fluid_exchange_rccl_attach_comm": [_ctx, C.c_void_p],
    "fluid_exchange_rccl_detach": [_ctx],
    "fluid_exchange_rccl_calls": [_ctx, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)],
}
# symbols with a non-status return type
OTHER_SYMBOLS = {"fluid_last_error": (C.c_char_p, []), "fluid_arena_bytes": (C.c_size_t, [_i]),
                 "fluid_arena_bytes_ex": (C.c_size_t, [_i, _i])}

_lib = None


def lib():
    """The loaded library; raises if libfluid_amd.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImportError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB)
        try:
            # torch wheels bundle their own libamdhip64/libhsa-runtime64; two HIP
            # runtimes in one process cannot both own the GPU.  Loading torch
            # first makes our DT_NEEDED libamdhip64.so.7 resolve to that copy.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = C.c_int, args
        for name, (res, args) in OTHER_SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise FluidError(rc, lib().fluid_last_error().decode(errors="replace"))
