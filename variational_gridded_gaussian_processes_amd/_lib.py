"""ctypes binding of libvggp_hip.so (include/vggp.h).  No fallback: if the library is missing or
fails, the caller gets an exception -- nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvggp_hip.so")

VGGP_OK, VGGP_EINVAL, VGGP_ENOTPD, VGGP_EHIP, VGGP_ENOMEM, VGGP_ESTATE, VGGP_ENOCONV, VGGP_ERCCL = 0, -1, -2, -3, -4, -5, -6, -7
UNIQUE_ID_BYTES = 128
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)      # vggp_allreduce_fn
KIND = {"matern12": 0, "matern32": 1, "matern52": 2, "rbf": 3}
NSTAGE = 20
FLAG_B0_F32_KDELTA = 1
FLAG_BLOCK_JACOBI = 2
FLAG_SCATTERED = 4
BASIS = {"points": 0, "b0": 1, "one": 2, "vff": 3, "b1": 4}


class VggpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libvggp_hip: {msg} (code {code})")
        self.code = code


class Desc(C.Structure):
    _fields_ = [("kind1", C.c_int32), ("basis1", C.c_int32), ("kind2", C.c_int32), ("basis2", C.c_int32),
                ("n1", C.c_int64), ("n2", C.c_int64), ("m1", C.c_int64), ("m2", C.c_int64),
                ("n_total", C.c_int64),
                ("x1", C.c_void_p), ("x2", C.c_void_p), ("grid1", C.c_void_p), ("grid2", C.c_void_p),
                ("warm_start", C.c_int32), ("flags", C.c_int32)]


class Info(C.Structure):
    _fields_ = [("jitter1", C.c_double), ("jitter2", C.c_double),
                ("sweeps1", C.c_int32), ("sweeps2", C.c_int32),
                ("rounds1", C.c_int32), ("rounds2", C.c_int32),
                ("status", C.c_int32), ("polished", C.c_int32)]


# every symbol include/vggp.h declares: name -> (restype, argtypes)
_P, _I64, _D, _I = C.c_void_p, C.c_int64, C.c_double, C.c_int
SYMBOLS = {
    "vggp_version": (_I, []),
    "vggp_last_error": (C.c_char_p, []),
    "vggp_create": (_I, [C.POINTER(_P), _I, _I, _I, _P]),
    "vggp_unique_id": (_I, [_P]),
    "vggp_set_allreduce": (_I, [_P, ALLREDUCE_FN, _P]),
    "vggp_allreduce": (_I, [_P, _P, _I64, _P]),
    "vggp_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "vggp_destroy": (_I, [_P]),
    "vggp_plan": (_I, [_P, C.POINTER(Desc)]),
    "vggp_payload_len": (_I64, [_P]),
    "vggp_workspace_bytes": (_I64, [_P]),
    "vggp_elbo_step": (_I, [_P, _P, _D, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), C.POINTER(Info), _P]),
    "vggp_elbo_partials": (_I, [_P, _P, C.POINTER(_D), _P, _P]),
    "vggp_elbo_finish": (_I, [_P, _P, _D, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), C.POINTER(Info), _P]),
    "vggp_elbo_step_masked": (_I, [_P, _P, _P, _D, _D, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), C.POINTER(Info), _P]),
    "vggp_elbo_step_masked_iter": (_I, [_P, _P, _P, _D, _D, C.POINTER(_D), _I, _D, _I, C.POINTER(_D), C.POINTER(_D), C.POINTER(Info), _P]),
    "vggp_elbo_step_scattered": (_I, [_P, _P, _D, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), C.POINTER(Info), _P]),
    "vggp_qv_masked": (_I, [_P, _P, _P, _P]),
    "vggp_qv": (_I, [_P, _P, _P, _P]),
    "vggp_qv_cov": (_I, [_P, _P, _P]),
    "vggp_zgrad": (_I, [_P, _P, _P, _P, _P]),
    "vggp_zgrad_scattered": (_I, [_P, _P, _P, _P, _P]),
    "vggp_set_inducing": (_I, [_P, C.c_int, _P, C.c_int64]),
    "vggp_posterior": (_I, [_P, _P, _P, _I64, _P, _P, _P]),
    "vggp_readout": (_I, [_P, _P, _I64, _P, _I64, _P, _P, _P, _P, _I, _P]),
    "vggp_readout_masked": (_I, [_P, _P, _I64, _P, _I64, _P, _P, _P, _P, _I, _P]),
    "vggp_posterior_masked": (_I, [_P, _P, _P, _I64, _P, _P, _P]),
    "vggp_posterior_cov": (_I, [_P, _P, _P, _I64, _P, _P]),
    "vggp_posterior_cov_masked": (_I, [_P, _P, _P, _I64, _P, _P]),
    "vggp_qv_cov_masked": (_I, [_P, _P, _P]),
    "vggp_factor_build": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _D, _I, _P, _P, _P, _P, _P]),
    "vggp_cholesky_inverse": (_I, [_P, _P, _I64, _P, _P, C.POINTER(_D), _P]),
    "vggp_eigh": (_I, [_P, _P, _I64, _P, _P, C.POINTER(C.c_int32), _I, _P]),
    "vggp_gemm": (_I, [_P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _P]),
    "vggp_kron_solve": (_I, [_P, _P, _I64, _P, _I64, _P, _P, _P]),
    "vggp_trsm": (_I, [_P, _P, _I64, _P, _I64, _P, _I, _P]),
    "vggp_sumsq": (_I, [_P, _P, _I64, C.POINTER(_D), _P]),
    "vggp_profile": (_I, [_P, _I]),
    "vggp_profile_read": (_I, [_P, C.POINTER(_D), C.POINTER(C.c_int32), _I]),
    "vggp_stage_name": (C.c_char_p, [_I]),
    "vggp_project_kernel_name": (C.c_char_p, []),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m variational_gridded_gaussian_processes_amd.build` "
            "(needs hipcc); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != VGGP_OK:
        msg = load().vggp_last_error()
        raise VggpError(rc, msg.decode() if msg else "unknown error")
