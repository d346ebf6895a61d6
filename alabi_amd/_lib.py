"""ctypes binding of libalabi_hip.so (the C ABI declared in include/alabi_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails,
an exception is raised.  PyTorch is used only for device memory / streams; every
numerical kernel of the hot path lives in the library.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libalabi_hip.so")

OK, NOT_PD, BAD_ARG, HIP_ERROR, NOT_COMPUTED, TIMEOUT = 0, 1, 2, 3, 4, 5
UTILITY_CODES = {"bape": 0, "agp": 1, "jones": 2}
KERNEL_CODES = {"ExpSquaredKernel": 0, "Matern32Kernel": 1, "Matern52Kernel": 2, "RationalQuadraticKernel": 3}
MAX_DIM = 64

_vp, _i, _ll, _d = C.c_void_p, C.c_int, C.c_longlong, C.c_double
_ull = C.c_ulonglong
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int)
_pll = C.POINTER(C.c_longlong)

# name -> (restype, argtypes); must list EVERY symbol of include/alabi_hip.h
SIGNATURES = {
    "alabi_abi_version": (_i, []),
    "alabi_status_string": (C.c_char_p, [_i]),
    "alabi_last_error": (C.c_char_p, []),
    "alabi_device_info": (_i, [_pi, _pi, C.c_char_p]),
    "alabi_gp_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "alabi_gp_destroy": (_i, [_vp]),
    "alabi_gp_set_hyper": (_i, [_vp, _d, _d, _d, _pd]),
    "alabi_gp_set_kernel": (_i, [_vp, _i, _d]),
    "alabi_gp_compute": (_i, [_vp, _vp, _i, _vp]),
    "alabi_gp_last_pivot": (_i, [_vp, _pi]),
    "alabi_gp_last_factor_path": (_i, [_vp, _pi]),
    "alabi_gp_set_y": (_i, [_vp, _vp, _vp]),
    "alabi_gp_predict": (_i, [_vp, _vp, _ll, _vp, _vp, _vp]),
    "alabi_gp_fit_predict": (_i, [_vp, _vp, _i, _vp, _vp, _ll, _vp, _pd, _vp]),
    "alabi_gp_predict_grad": (_i, [_vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp]),
    "alabi_gp_predict_grad_point": (_i, [_vp, _vp, _vp, _vp]),
    "alabi_gp_logdet": (_i, [_vp, _pd, _vp]),
    "alabi_gp_nll": (_i, [_vp, _pd, _vp]),
    "alabi_gp_grad_log_likelihood": (_i, [_vp, _pd, _vp]),
    "alabi_gp_append": (_i, [_vp, _vp, _vp]),
    "alabi_gp_batch_create": (_i, [_i, _i, _ll, C.POINTER(_vp)]),
    "alabi_gp_batch_destroy": (_i, [_vp]),
    "alabi_gp_batch_fit_predict": (_i, [_vp, _vp, _vp, _i, _i, _pd, _vp, _pll, _vp, _pll, _vp, _pd, _pi, _vp]),
    "alabi_gp_batch_get_factor": (_i, [_vp, _i, _vp, _vp]),
    "alabi_gp_batch_get_alpha": (_i, [_vp, _i, _vp, _vp]),
    "alabi_gp_batch_timeouts": (_i, [_vp, _pi]),
    "alabi_cv_fold_lists": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "alabi_gp_set_mean": (_i, [_vp, _d]),
    "alabi_gp_get_alpha": (_i, [_vp, _vp, _vp]),
    "alabi_gp_get_factor": (_i, [_vp, _vp, _vp]),
    "alabi_gp_get_inverse": (_i, [_vp, _vp, _vp]),
    "alabi_gp_n": (_i, [_vp, _pi]),
    "alabi_kernel_matrix": (_i, [_vp, _i, _vp, _i, _i, _i, _d, _d, _pd, _vp, _vp]),
    "alabi_utility_scan": (_i, [_vp, _i, _vp, _ll, _pd, _d, _vp, _vp, _vp, _pd, _pll, _vp]),
    "alabi_utility_polish": (_i, [_vp, _i, _vp, _vp, _d, _i, _vp, _pd, _pi, _vp]),
    "alabi_utility_eval": (_i, [_i, _vp, _ll, _i, _pd, _d, _vp, _vp, _vp, _vp]),
    "alabi_ens_create": (_i, [_vp, _i, _i, _i, _pd, _ull, C.POINTER(_vp)]),
    "alabi_ens_destroy": (_i, [_vp]),
    "alabi_ens_set_logp_affine": (_i, [_vp, _d, _d]),
    "alabi_ens_set_normal_prior": (_i, [_vp, _pd, _pd]),
    "alabi_ens_set_logp_map": (_i, [_vp, _i]),
    "alabi_ens_surrogate": (_i, [_vp, _vp, _i, _vp, _vp]),
    "alabi_ens_propose": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "alabi_ens_accept": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "alabi_ens_set_stream": (_i, [_vp, _i]),
    "alabi_ens_last_path": (_i, [_vp, _pi]),
    "alabi_ens_group_plan": (_i, [_vp, _pi]),
    "alabi_ens_lnprob": (_i, [_vp, _vp, _vp, _vp]),
    "alabi_ens_run": (_i, [_vp, _vp, _vp, _ll, _ll, _i, _d, _vp, _vp, _vp, _vp]),
    "alabi_chain_autocorr": (_i, [_vp, _ll, _i, _i, _vp, _vp]),
    "alabi_ens_draw": (_i, [_vp, _ll, _i, _d, _vp]),
    "alabi_ens_half_step": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "alabi_dist_unique_id": (_i, [_vp]),
    "alabi_dist_comm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "alabi_dist_comm_create_callback": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "alabi_dist_comm_destroy": (_i, [_vp]),
    "alabi_dist_comm_stats": (_i, [_vp, _pll]),
    "alabi_ens_run_sharded": (_i, [_vp, _vp, _vp, _vp, _ll, _ll, _i, _d, _vp, _vp, _vp, _vp]),
    "alabi_ens_step_lists": (_i, [_vp, _i, _vp, _pi, _vp]),
    "alabi_ens_step_with_randoms": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _d, _vp, _vp]),
    "alabi_ens_export_draws": (_i, [_vp, _ll, _d, _vp, _pi, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


class AlabiHipError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().alabi_status_string(status).decode()
        detail = lib().alabi_last_error().decode() if status == HIP_ERROR else ""
        super().__init__(f"{where}: {msg}" + (f" ({detail})" if detail else ""))


def build(verbose=False):
    """Compile libalabi_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libalabi_hip.so failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return LIB_PATH


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C alabi_amd/csrc`).  There is no CPU fallback for the hot path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status, where):
    if status != OK:
        raise AlabiHipError(status, where)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def host_doubles(values):
    arr = (C.c_double * len(values))(*[float(v) for v in values])
    return arr


_raw_stream = None


def current_stream():
    """torch's current stream of the current device as a hipStream_t.  Through torch._C._cuda_getCurrentRawStream where it exists
    (0.3 us; torch.cuda.current_stream().cuda_stream builds a Stream object: ~10 us, and every call into the library needs it --
    it was 0.4 ms of a 4 ms active-learning iteration)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
