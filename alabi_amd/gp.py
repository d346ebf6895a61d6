"""HipGP: the george.GP object protocol that alabi touches, backed by libalabi_hip.so.

Mirrors (member for member) what the reference calls on ``george.GP`` -- SURVEY.md
section 8(b) seam #1: construction with ``kernel = var(y) * ExpSquaredKernel(metric, ndim)``,
``mean=``, ``white_noise=`` (alabi/core.py:1141, alabi/gp_utils.py:230-233); ``compute``
(core.py:1158); ``predict`` (core.py:85, :1601); ``log_likelihood`` /
``grad_log_likelihood`` (core.py:1248, :1261); the parameter-vector protocol
(core.py:705-733, :1050); ``kernel.get_value`` (utility.py:549); ``solver.get_inverse``
(utility.py:610); private ``_x``, ``_y``, ``_alpha`` (utility.py:577); deepcopy / pickle.

All arithmetic runs in the HIP library on the current CUDA(HIP) device; this class only
moves arrays and keeps the hyper-parameter vector.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

__all__ = ["HipGP"]


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("alabi_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback for the GP hot path")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a, ndim=None):
    if isinstance(a, torch.Tensor):
        t = a.to(device=_dev(), dtype=torch.float64)
    else:
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=_dev())
    if ndim == 2 and t.dim() == 1:
        t = t.reshape(1, -1)
    return t.contiguous()


class _KernelView:
    """gp.kernel: get_value(x1, x2) without white noise, parameter names (utility.py:549, gp_utils.py:333)."""

    def __init__(self, gp):
        self._gp = gp

    @property
    def ndim(self):
        return self._gp.ndim

    def get_parameter_names(self, include_frozen=False):
        if not getattr(self._gp, "fit_amp", True):
            extra = ["log_alpha"] if self._gp.kernel_name == "RationalQuadraticKernel" else []
            return tuple(extra + [f"metric:log_M_{i}_{i}" for i in range(self._gp.ndim)])
        extra = ["k2:log_alpha"] if self._gp.kernel_name == "RationalQuadraticKernel" else []
        return tuple(["k1:log_constant"] + extra + [f"k2:metric:log_M_{i}_{i}" for i in range(self._gp.ndim)])

    def get_parameter_vector(self, include_frozen=False):
        extra = [self._gp.log_alpha] if self._gp.kernel_name == "RationalQuadraticKernel" else []
        head = [self._gp.log_constant] if getattr(self._gp, "fit_amp", True) else []
        return np.concatenate([head, extra, self._gp.log_M])

    def get_value(self, x1, x2=None, diag=False):
        gp = self._gp
        a = _to_dev(x1, 2)
        if diag:
            return np.full(a.shape[0], np.exp(gp.log_constant))
        b = a if x2 is None else _to_dev(x2, 2)
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=a.device)
        st = _lib.lib().alabi_kernel_matrix(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0], gp.ndim,
                                            _lib.KERNEL_CODES[gp.kernel_name], float(gp.log_alpha),
                                            float(gp.log_constant), _lib.host_doubles(gp.log_M), _lib.ptr(out),
                                            _lib.current_stream())
        _lib.check(st, "alabi_kernel_matrix")
        return out.cpu().numpy()


class _SolverView:
    """gp.solver: log_determinant, get_inverse (utility.py:610), apply_inverse."""

    def __init__(self, gp):
        self._gp = gp

    @property
    def log_determinant(self):
        return self._gp._logdet()

    @property
    def factor_path(self):
        """How the last compute factorised: "steps" (launch per step), "queue" (one-launch task queue) or "queue-timeout"
        (the queue's wait ran out; factorised again step by step)."""
        out = C.c_int(0)
        _lib.check(_lib.lib().alabi_gp_last_factor_path(self._gp._handle, C.byref(out)), "alabi_gp_last_factor_path")
        return {0: None, 1: "steps", 2: "queue", 3: "queue-timeout"}[out.value]

    def get_factor(self):
        gp = self._gp
        gp._require_computed()
        n = gp._n
        out = torch.empty((n, n), dtype=torch.float64, device=_dev())
        _lib.check(_lib.lib().alabi_gp_get_factor(gp._handle, _lib.ptr(out), _lib.current_stream()), "alabi_gp_get_factor")
        return out

    def get_inverse_device(self):
        """K^-1 [N,N] (device tensor) = W^T W on the matrix cores from the library's cached W = L^-1 (alabi_gp_get_inverse)."""
        gp = self._gp
        gp._require_computed()
        out = torch.empty((gp._n, gp._n), dtype=torch.float64, device=_dev())
        _lib.check(_lib.lib().alabi_gp_get_inverse(gp._handle, _lib.ptr(out), _lib.current_stream()), "alabi_gp_get_inverse")
        return out

    def get_inverse(self):
        """george's solver.get_inverse() (reference: alabi/utility.py:610)."""
        return self.get_inverse_device().cpu().numpy()


class HipGP:
    def __init__(self, ndim, mean=0.0, white_noise=-12.0, log_constant=0.0, log_M=None,
                 fit_mean=True, fit_white_noise=True, kernel="ExpSquaredKernel", log_alpha=1.0, fit_amp=True):
        if kernel not in _lib.KERNEL_CODES:
            raise ValueError(f"Kernel '{kernel}' is not a valid option. Valid options: {', '.join(_lib.KERNEL_CODES)}")
        self.kernel_name = kernel
        self.log_alpha = float(log_alpha)      # RationalQuadraticKernel only (reference: log_alpha=1, core.py:1003)
        self.ndim = int(ndim)
        if not (1 <= self.ndim <= _lib.MAX_DIM):
            raise ValueError(f"ndim must be in [1, {_lib.MAX_DIM}]")
        self.mean_value = float(mean)
        self.white_noise_value = float(white_noise)
        self.log_constant = float(log_constant)
        self.log_M = np.zeros(self.ndim) if log_M is None else np.array(log_M, dtype=np.float64).ravel().copy()
        if self.log_M.size != self.ndim:
            raise ValueError("log_M must have ndim entries")
        self.fit_mean = bool(fit_mean)
        self.fit_white_noise = bool(fit_white_noise)
        # fit_amp=False: the reference builds the bare kernel (no ``kernel *= var(y)``, gp_utils.py:230), so george's model
        # has no constant factor and no log_constant parameter; the amplitude stays at exp(log_constant) here (0.0 = 1).
        self.fit_amp = bool(fit_amp)
        self.kernel = _KernelView(self)
        self.solver = _SolverView(self)
        self._x = None          # numpy [N,d] (host copy: pickling, protocol)
        self._y = None          # numpy [N]
        self._alpha_host = None
        self._handle = None
        self._cap = 0
        self._n = 0
        self._x_dev = None
        self.computed = False
        self.dirty = True
        self._y_set = False

    # ------------------------------------------------------------------ handle lifetime
    def _ensure_handle(self, n):
        if self._handle is not None and n <= self._cap:
            return
        self._release()
        cap = max(int(n * 1.25) + 64, 128)
        h = C.c_void_p()
        _lib.check(_lib.lib().alabi_gp_create(cap, self.ndim, C.byref(h)), "alabi_gp_create")
        self._handle = h
        self._cap = cap

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            try:
                _lib.lib().alabi_gp_destroy(self._handle)
            except Exception:
                pass
        self._handle = None
        self._cap = 0

    def __del__(self):
        self._release()

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in ("_handle", "_x_dev", "kernel", "solver", "_last_grad_buf"):
            st[k] = None
        st["_cap"] = 0
        st["_was_computed"] = self.computed
        st["computed"] = False
        st["_y_set"] = False
        st["dirty"] = True
        return st

    def __setstate__(self, st):
        was = st.pop("_was_computed", False)
        self.__dict__.update(st)
        self.kernel = _KernelView(self)
        self.solver = _SolverView(self)
        self._restore = bool(was and self._x is not None)

    def _require_computed(self):
        if getattr(self, "_restore", False):
            self._restore = False
            self.compute(self._x)
        if not self.computed or self.dirty:
            if self._x is None:
                raise RuntimeError("You need to compute the model first")
            self.compute(self._x)

    # ------------------------------------------------------------ parameter-vector protocol
    def get_parameter_names(self, include_frozen=False):
        names = []
        if self.fit_mean or include_frozen:
            names.append("mean:value")
        if self.fit_white_noise or include_frozen:
            names.append("white_noise:value")
        if getattr(self, "fit_amp", True):
            names.append("kernel:k1:log_constant")
        k2 = "kernel:k2" if getattr(self, "fit_amp", True) else "kernel"     # no product kernel without the constant factor
        if self.kernel_name == "RationalQuadraticKernel":
            names.append(f"{k2}:log_alpha")
        names += [f"{k2}:metric:log_M_{i}_{i}" for i in range(self.ndim)]
        return tuple(names)

    def get_parameter_vector(self, include_frozen=False):
        v = []
        if self.fit_mean or include_frozen:
            v.append(self.mean_value)
        if self.fit_white_noise or include_frozen:
            v.append(self.white_noise_value)
        if getattr(self, "fit_amp", True):
            v.append(self.log_constant)
        if self.kernel_name == "RationalQuadraticKernel":
            v.append(self.log_alpha)
        v.extend(self.log_M.tolist())
        return np.array(v, dtype=np.float64)

    def get_parameter_dict(self, include_frozen=False):
        return OrderedDict(zip(self.get_parameter_names(include_frozen), self.get_parameter_vector(include_frozen)))

    def set_parameter_vector(self, vector, include_frozen=False):
        p = np.asarray(vector, dtype=np.float64).ravel()
        n_expected = len(self.get_parameter_names(include_frozen))
        if p.size != n_expected:
            raise ValueError(f"dimension mismatch: expected {n_expected} parameters, got {p.size}")
        if np.array_equal(p, self.get_parameter_vector(include_frozen)):
            return          # unchanged (optimisers evaluate the objective and its gradient at the same point): keep the factor
        i = 0
        if self.fit_mean or include_frozen:
            self.mean_value = float(p[i]); i += 1
        if self.fit_white_noise or include_frozen:
            self.white_noise_value = float(p[i]); i += 1
        if getattr(self, "fit_amp", True):
            self.log_constant = float(p[i]); i += 1
        if self.kernel_name == "RationalQuadraticKernel":
            self.log_alpha = float(p[i]); i += 1
        self.log_M = p[i:i + self.ndim].copy()
        self.dirty = True
        self._y_set = False
        self._alpha_host = None

    def full_hyper(self, vector=None, include_frozen=False):
        """[mean, log white noise, log amplitude, log alpha, log M_1..d] that ``set_parameter_vector(vector)`` would give (the
        current values for ``vector=None``) WITHOUT touching this object: one row of HipGPBatch.fit_predict's table."""
        mean, wn, amp, la, lm = self.mean_value, self.white_noise_value, self.log_constant, self.log_alpha, self.log_M
        if vector is not None:
            p = np.asarray(vector, dtype=np.float64).ravel()
            n_expected = len(self.get_parameter_names(include_frozen))
            if p.size != n_expected:
                raise ValueError(f"dimension mismatch: expected {n_expected} parameters, got {p.size}")
            i = 0
            if self.fit_mean or include_frozen:
                mean = float(p[i]); i += 1
            if self.fit_white_noise or include_frozen:
                wn = float(p[i]); i += 1
            if getattr(self, "fit_amp", True):
                amp = float(p[i]); i += 1
            if self.kernel_name == "RationalQuadraticKernel":
                la = float(p[i]); i += 1
            lm = p[i:i + self.ndim]
        return np.concatenate([[mean, wn, amp, la], np.asarray(lm, dtype=np.float64)])

    def __len__(self):
        return len(self.get_parameter_names())

    # ------------------------------------------------------------------ compute / predict
    def _push_hyper(self):
        _lib.check(_lib.lib().alabi_gp_set_kernel(self._handle, _lib.KERNEL_CODES[self.kernel_name], self.log_alpha),
                   "alabi_gp_set_kernel")
        st = _lib.lib().alabi_gp_set_hyper(self._handle, self.mean_value, self.white_noise_value, self.log_constant,
                                           _lib.host_doubles(self.log_M))
        _lib.check(st, "alabi_gp_set_hyper")

    def compute(self, x, yerr=0.0, quiet=False):
        """Assemble K and factorise it (george GP.compute; reference core.py:1158)."""
        xd = _to_dev(x, 2)
        if xd.shape[1] != self.ndim:
            raise ValueError(f"x has {xd.shape[1]} columns, GP has ndim={self.ndim}")
        n = int(xd.shape[0])
        self._ensure_handle(n)
        self._push_hyper()
        self._x_dev = xd
        self._x = x if (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.ndim == 2) else xd.cpu().numpy()
        self._n = n
        self._restore = False
        st = _lib.lib().alabi_gp_compute(self._handle, _lib.ptr(xd), n, _lib.current_stream())
        self._y_set = False
        self._alpha_host = None
        if st == _lib.NOT_PD:
            self.computed = False
            piv = C.c_int(0)
            _lib.lib().alabi_gp_last_pivot(self._handle, C.byref(piv))
            if quiet:
                return False
            raise np.linalg.LinAlgError(f"{piv.value}-th leading minor of the array is not positive definite")
        _lib.check(st, "alabi_gp_compute")
        self.computed = True
        self.dirty = False
        return True

    def fit_predict_device(self, x_dev, y_dev, xs_dev):
        """compute(x) + log_likelihood(y) + predict(y, xs) in ONE library call on device tensors (alabi_gp_fit_predict): the
        per-fold work of the k-fold CV search (gp_utils.py:568-600).  Returns (log-likelihood, mu[M] device tensor); the
        log-likelihood is -inf when K is not positive definite (mu is then None)."""
        n = int(x_dev.shape[0])
        m = int(xs_dev.shape[0])
        self._ensure_handle(n)
        self._push_hyper()
        self._x_dev, self._x, self._n = x_dev, None, n
        self._restore = False
        self._y_set, self._alpha_host, self._y, self._y_key = False, None, None, None
        mu = torch.empty(m, dtype=torch.float64, device=x_dev.device)
        out = C.c_double(0.0)
        st = _lib.lib().alabi_gp_fit_predict(self._handle, _lib.ptr(x_dev), n, _lib.ptr(y_dev), _lib.ptr(xs_dev), m, _lib.ptr(mu),
                                             C.byref(out), _lib.current_stream())
        if st == _lib.NOT_PD:
            self.computed = False
            return -np.inf, None
        _lib.check(st, "alabi_gp_fit_predict")
        self.computed, self.dirty = True, False
        ll = -out.value
        return (ll if np.isfinite(ll) else -np.inf), mu

    def compute_from(self, prev, x, quiet=False):
        """``compute(x)``; when ``prev`` (another HipGP) holds the factorisation of ``x[:-1]`` with the same kernel
        hyper-parameters, its device handle is taken over and the last row is APPENDED in O(N^2) (alabi_gp_append) instead of
        refactorising -- the reference's refit after every active-learning iteration (core.py:1780 -> :1158).  ``prev`` is
        left uncomputed (it refactorises on demand if it is used again)."""
        try:
            xa = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
            ok = (os.environ.get("ALABI_NO_APPEND", "0") != "1"
                  and prev is not None and prev is not self and isinstance(prev, HipGP) and prev.computed and not prev.dirty
                  and prev._handle is not None and self._handle is None and prev.ndim == self.ndim
                  and prev.kernel_name == self.kernel_name and prev.log_alpha == self.log_alpha
                  and prev.white_noise_value == self.white_noise_value and prev.log_constant == self.log_constant
                  and np.array_equal(prev.log_M, self.log_M) and xa.ndim == 2 and xa.shape[0] == prev._n + 1
                  and prev._n % 64 != 0 and prev._n + 1 <= prev._cap and isinstance(prev._x, np.ndarray)
                  and prev._x.shape == (prev._n, self.ndim) and np.array_equal(prev._x, xa[:-1]))
        except Exception:  # noqa: BLE001
            ok = False
        if ok:
            handle, cap, n_prev = prev._handle, prev._cap, prev._n
            x_last = torch.as_tensor(xa[-1], device=_dev())
            st = _lib.lib().alabi_gp_set_mean(handle, self.mean_value)
            if st == _lib.OK:
                st = _lib.lib().alabi_gp_append(handle, _lib.ptr(x_last), _lib.current_stream())
            if st == _lib.OK:
                prev._handle, prev._cap = None, 0
                prev.computed, prev.dirty, prev._y_set, prev._x_dev = False, True, False, None
                self._handle, self._cap = handle, cap
                self._x, self._x_dev, self._n = xa, None, n_prev + 1
                self._restore = False
                self._y_set, self._alpha_host = False, None
                self.computed, self.dirty = True, False
                self.appended = getattr(prev, "appended", 0) + 1
                return True
        return self.compute(x, quiet=quiet)

    def recompute(self, quiet=False, **kw):
        if self._x is None:
            raise RuntimeError("You need to compute the model first")
        if self.dirty or not self.computed or getattr(self, "_restore", False):
            try:
                ok = self.compute(self._x_dev if self._x_dev is not None else self._x, quiet=quiet)
            except np.linalg.LinAlgError:
                if quiet:
                    return False
                raise
            return bool(ok)
        return True

    def _set_y(self, y):
        if isinstance(y, torch.Tensor):
            key = (y.data_ptr(), y._version, tuple(y.shape))
            same = self._y_set and getattr(self, "_y_key", None) == key
            yd = None if same else _to_dev(y).reshape(-1)
            yh = None
            self._y_key = key
        else:
            yh = np.asarray(y, dtype=np.float64).ravel()
            same = self._y_set and self._y is not None and np.array_equal(yh, self._y)
            yd = None
            self._y_key = None
        if same:
            return
        if yd is None:
            yd = _to_dev(yh)
        if yd.numel() != self._n:
            raise ValueError(f"dimension mismatch: y has {yd.numel()} entries, GP was computed on {self._n} points")
        _lib.check(_lib.lib().alabi_gp_set_y(self._handle, _lib.ptr(yd), _lib.current_stream()), "alabi_gp_set_y")
        self._y = yh if yh is not None else yd.cpu().numpy()
        self._y_set = True
        self._alpha_host = None

    def predict_device(self, y, t, return_var=False):
        """predict() with device tensors in and out (no host round trip)."""
        self._require_computed()
        self._set_y(y)
        td = _to_dev(t, 2)
        if td.shape[1] != self.ndim:
            raise ValueError(f"t has {td.shape[1]} columns, GP has ndim={self.ndim}")
        m = int(td.shape[0])
        mu = torch.empty(m, dtype=torch.float64, device=td.device)
        var = torch.empty(m, dtype=torch.float64, device=td.device) if return_var else None
        st = _lib.lib().alabi_gp_predict(self._handle, _lib.ptr(td), m, _lib.ptr(mu), _lib.ptr(var), _lib.current_stream())
        _lib.check(st, "alabi_gp_predict")
        return (mu, var) if return_var else mu

    def predict_grad_device(self, y, t):
        """(mu[M], var[M], dmu[M,d], dvar[M,d]) at the query points t, gradients with respect to t.

        Closed-form replacement of grad_gp_mean_prediction / grad_gp_var_prediction (utility.py:558-623):
        dmu = (dk/dx)^T alpha, dvar = -2 (dk/dx)^T K^-1 k, device tensors in and out."""
        self._require_computed()
        self._set_y(y)
        td = _to_dev(t, 2)
        if td.shape[1] != self.ndim:
            raise ValueError(f"t has {td.shape[1]} columns, GP has ndim={self.ndim}")
        m, d = int(td.shape[0]), self.ndim
        buf = torch.empty(m * (2 + 2 * d), dtype=torch.float64, device=td.device)     # ONE allocation: the four results are views of it
        mu, var = buf[:m], buf[m:2 * m]
        dmu, dvar = buf[2 * m:2 * m + m * d].view(m, d), buf[2 * m + m * d:].view(m, d)
        st = _lib.lib().alabi_gp_predict_grad(self._handle, _lib.ptr(td), m, _lib.ptr(mu), _lib.ptr(var), _lib.ptr(dmu),
                                              _lib.ptr(dvar), _lib.current_stream())
        _lib.check(st, "alabi_gp_predict_grad")
        self._last_grad_buf = buf
        return mu, var, dmu, dvar

    def predict_grad_host(self, y, t):
        """The same as NumPy arrays, brought back with ONE device-to-host copy (the polish step of find_next_point calls this ~30
        times per active-learning iteration: four separate copies cost more than the kernels)."""
        if not isinstance(t, torch.Tensor) and np.size(t) == self.ndim:
            # one point: host buffers on both sides, no tensor on the way (alabi_gp_predict_grad_point)
            self._require_computed()
            self._set_y(y)
            d = self.ndim
            x = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(d))
            out = np.empty(2 + 2 * d, dtype=np.float64)
            _lib.check(_lib.lib().alabi_gp_predict_grad_point(self._handle, x.ctypes.data, out.ctypes.data, _lib.current_stream()),
                       "alabi_gp_predict_grad_point")
            return out[0:1], out[1:2], out[2:2 + d].reshape(1, d), out[2 + d:].reshape(1, d)
        mu, var, dmu, dvar = self.predict_grad_device(y, t)
        m, d = int(mu.shape[0]), self.ndim
        h = self._last_grad_buf.cpu().numpy()
        return h[:m], h[m:2 * m], h[2 * m:2 * m + m * d].reshape(m, d), h[2 * m + m * d:].reshape(m, d)

    def predict(self, y, t, return_cov=True, return_var=False, cache=True, kernel=None):
        """george GP.predict (core.py:85, :95, :1441, :1601).  return_var wins over return_cov."""
        if return_var:
            mu, var = self.predict_device(y, t, return_var=True)
            return mu.cpu().numpy(), var.cpu().numpy()
        if return_cov:
            raise NotImplementedError("HipGP.predict(return_cov=True) is not on alabi's path; "
                                      "pass return_cov=False or return_var=True")
        return self.predict_device(y, t, return_var=False).cpu().numpy()

    # ----------------------------------------------------------------------- likelihood
    def _logdet(self):
        self._require_computed()
        out = C.c_double(0.0)
        _lib.check(_lib.lib().alabi_gp_logdet(self._handle, C.byref(out), _lib.current_stream()), "alabi_gp_logdet")
        return out.value

    def log_likelihood(self, y, quiet=False):
        """george GP.log_likelihood (core.py:1248, gp_utils.py:139)."""
        try:
            ok = self.recompute(quiet=quiet)
        except np.linalg.LinAlgError:
            if quiet:
                return -np.inf
            raise
        if not ok:
            return -np.inf
        self._set_y(y)
        out = C.c_double(0.0)
        _lib.check(_lib.lib().alabi_gp_nll(self._handle, C.byref(out), _lib.current_stream()), "alabi_gp_nll")
        ll = -out.value
        return ll if np.isfinite(ll) else -np.inf

    def nll(self, y):
        return -self.log_likelihood(y, quiet=True)

    def grad_log_likelihood(self, y, quiet=False):
        """d logL / d p over the unfrozen vector (george protocol; reference call sites core.py:1261, gp_utils.py:165).

        Analytic, on the device: 0.5 tr((alpha alpha^T - K^-1) dK/dp) with K^-1 = L^-T L^-1 formed block by block on the
        matrix cores and contracted with the re-evaluated kernel derivatives (csrc/gp_grad.hip); one factorisation."""
        if not self.recompute(quiet=quiet):
            return np.zeros(len(self.get_parameter_vector()))      # george: quiet=True gives zeros when K is not PD
        self._set_y(y)
        out = (C.c_double * (self.ndim + 4))()
        _lib.check(_lib.lib().alabi_gp_grad_log_likelihood(self._handle, out, _lib.current_stream()),
                   "alabi_gp_grad_log_likelihood")
        full = np.array(out[:], dtype=np.float64)
        g = []
        if self.fit_mean:
            g.append(full[0])
        if self.fit_white_noise:
            g.append(full[1])
        if getattr(self, "fit_amp", True):
            g.append(full[2])
        if self.kernel_name == "RationalQuadraticKernel":
            g.append(full[3])
        g.extend(full[4:4 + self.ndim])
        return np.array(g)

    def grad_log_likelihood_fd(self, y, h=1e-5):
        """Central differences of the native likelihood (2 x len(p) factorisations): the check for the analytic gradient."""
        p0 = self.get_parameter_vector()
        g = np.zeros_like(p0)
        for i in range(p0.size):
            pp = p0.copy(); pp[i] += h
            self.set_parameter_vector(pp)
            fp = self.log_likelihood(y, quiet=True)
            pm = p0.copy(); pm[i] -= h
            self.set_parameter_vector(pm)
            fm = self.log_likelihood(y, quiet=True)
            g[i] = (fp - fm) / (2.0 * h)
        self.set_parameter_vector(p0)
        self.recompute(quiet=True)
        return g

    # -------------------------------------------------------------------- private members
    @property
    def _alpha(self):
        """alpha = K^-1 (y - mean) as a host array (reference reads gp._alpha at utility.py:577)."""
        if self._alpha_host is None:
            if not (self.computed and self._y_set):
                return None
            out = torch.empty(self._n, dtype=torch.float64, device=_dev())
            _lib.check(_lib.lib().alabi_gp_get_alpha(self._handle, _lib.ptr(out), _lib.current_stream()), "alabi_gp_get_alpha")
            self._alpha_host = out.cpu().numpy()
        return self._alpha_host

    @property
    def handle(self):
        self._require_computed()
        return self._handle
