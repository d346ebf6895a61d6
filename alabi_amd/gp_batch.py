"""Many GP fits in one library call: the k-fold cross-validation search of ``init_gp`` / ``_opt_gp``.

Reference: alabi/gp_utils.py:511-637 (one worker call = k folds of compute / log_likelihood / predict) mapped over the
candidates by a process pool at gp_utils.py:640-700.  Here the (candidate, fold) jobs of a whole search stage go to
``alabi_gp_batch_fit_predict`` (alabi_amd/csrc/gp_batch.hip): batched assembly, ONE launch of the Cholesky task queue for all
matrices, batched solve and held-out mean, one read-back.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .gp import _dev

__all__ = ["HipGPBatch"]


class HipGPBatch:
    def __init__(self, ndim, kernel="ExpSquaredKernel", workspace_bytes=0):
        if kernel not in _lib.KERNEL_CODES:
            raise ValueError(f"Kernel '{kernel}' is not a valid option. Valid options: {', '.join(_lib.KERNEL_CODES)}")
        self.ndim = int(ndim)
        self.kernel_name = kernel
        h = C.c_void_p()
        _lib.check(_lib.lib().alabi_gp_batch_create(self.ndim, _lib.KERNEL_CODES[kernel], int(workspace_bytes), C.byref(h)),
                   "alabi_gp_batch_create")
        self._handle = h
        self.njobs = 0

    def close(self):
        if getattr(self, "_handle", None) is not None:
            try:
                _lib.lib().alabi_gp_batch_destroy(self._handle)
            except Exception:  # noqa: BLE001
                pass
        self._handle = None

    def __del__(self):
        self.close()

    @property
    def timeouts(self):
        n = C.c_int(0)
        _lib.check(_lib.lib().alabi_gp_batch_timeouts(self._handle, C.byref(n)), "alabi_gp_batch_timeouts")
        return n.value

    def fit_predict(self, theta_dev, y_dev, hyper, train_sets, val_sets):
        """``hyper`` [B, 4 + d]: mean, log white noise, log amplitude, log alpha, log M (HipGP.full_hyper); ``train_sets`` /
        ``val_sets``: B integer arrays of rows of ``theta_dev`` [n, d] / ``y_dev`` [n] (device, float64).
        Returns (log-likelihood [B] (-inf where K is not positive definite), status [B], mu, val_off) with ``mu`` ONE device
        tensor holding every job's held-out means, job b at mu[val_off[b]:val_off[b + 1]]."""
        B = len(train_sets)
        if len(val_sets) != B:
            raise ValueError("val_sets must be as long as train_sets")
        n = int(theta_dev.shape[0])
        tr_off = np.zeros(B + 1, dtype=np.int64); np.cumsum([len(t) for t in train_sets], out=tr_off[1:])
        va_off = np.zeros(B + 1, dtype=np.int64); np.cumsum([len(v) for v in val_sets], out=va_off[1:])
        tr_all = np.concatenate([np.asarray(t).ravel() for t in train_sets]).astype(np.int32, copy=False) if B else np.zeros(0, np.int32)
        va_all = np.concatenate([np.asarray(v).ravel() for v in val_sets]).astype(np.int32, copy=False) if B else np.zeros(0, np.int32)
        for a in (tr_all, va_all):
            if a.size and (a.min() < 0 or a.max() >= n):
                raise ValueError("row index out of range")
        dev = theta_dev.device
        return self.fit_predict_indexed(theta_dev, y_dev, hyper, torch.as_tensor(tr_all, device=dev), tr_off,
                                        torch.as_tensor(va_all, device=dev) if va_all.size else None, va_off)

    def fit_predict_indexed(self, theta_dev, y_dev, hyper, tr_dev, tr_off, va_dev, va_off):
        """The same with the row lists already on the device: ``tr_dev`` / ``va_dev`` int32 tensors holding every job's rows one
        after the other (rows must lie in [0, n): NOT checked here), ``tr_off`` / ``va_off`` host int64 offsets [B + 1]."""
        hyper = np.ascontiguousarray(np.asarray(hyper, dtype=np.float64))
        tr_off = np.ascontiguousarray(tr_off, dtype=np.int64)
        va_off = np.ascontiguousarray(va_off, dtype=np.int64)
        B = len(tr_off) - 1
        if hyper.shape != (B, 4 + self.ndim) or len(va_off) != B + 1:
            raise ValueError("hyper must be [jobs, 4 + ndim] and the offset arrays [jobs + 1]")
        n = int(theta_dev.shape[0])
        if theta_dev.dtype != torch.float64 or y_dev.dtype != torch.float64 or tuple(theta_dev.shape) != (n, self.ndim) \
                or tuple(y_dev.shape) != (n,) or not theta_dev.is_contiguous() or not y_dev.is_contiguous():
            raise ValueError("theta_dev [n, ndim] / y_dev [n] must be contiguous float64 device tensors")
        if tr_dev.dtype != torch.int32 or tr_dev.numel() != tr_off[-1] or (va_off[-1] > 0 and (va_dev is None or va_dev.dtype != torch.int32
                                                                                              or va_dev.numel() != va_off[-1])):
            raise ValueError("row lists must be int32 device tensors matching the offsets")
        dev = theta_dev.device
        mu = torch.empty(int(va_off[-1]), dtype=torch.float64, device=dev) if va_off[-1] > 0 else None
        nll = np.empty(B, dtype=np.float64)
        status = np.empty(B, dtype=np.int32)
        st = _lib.lib().alabi_gp_batch_fit_predict(
            self._handle, _lib.ptr(theta_dev), _lib.ptr(y_dev), n, B, hyper.ctypes.data_as(_lib._pd), _lib.ptr(tr_dev),
            tr_off.ctypes.data_as(_lib._pll), _lib.ptr(va_dev), va_off.ctypes.data_as(_lib._pll), _lib.ptr(mu),
            nll.ctypes.data_as(_lib._pd), status.ctypes.data_as(_lib._pi), _lib.current_stream())
        _lib.check(st, "alabi_gp_batch_fit_predict")
        self.njobs = B
        ll = np.where(status == 0, -nll, -np.inf)
        ll[~np.isfinite(ll)] = -np.inf
        return ll, status, mu, va_off

    def get_factor(self, job, n_train):
        out = torch.empty((n_train, n_train), dtype=torch.float64, device=_dev())
        _lib.check(_lib.lib().alabi_gp_batch_get_factor(self._handle, int(job), _lib.ptr(out), _lib.current_stream()),
                   "alabi_gp_batch_get_factor")
        return out

    def get_alpha(self, job, n_train):
        out = torch.empty(n_train, dtype=torch.float64, device=_dev())
        _lib.check(_lib.lib().alabi_gp_batch_get_alpha(self._handle, int(job), _lib.ptr(out), _lib.current_stream()),
                   "alabi_gp_batch_get_alpha")
        return out
