"""fp64 NumPy/SciPy restatement of the GP algebra alabi delegates to george.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED at this
boundary: george is an un-vendored third-party dependency of the reference
(/root/reference/setup.py:15) and is not installed; the reference has no tests.
Semantics restated from george 0.4.x (``george/gp.py``, ``george/solvers/basic.py``,
``george/kernels.py``) and anchored on the reference call sites cited per function.

Hyper-parameter vector layout (reference: alabi/core.py:1050-1066 and the order
printed in docs/source/save_reload.py:117-120):

    [ mean:value, white_noise:value, kernel:k1:log_constant,
      kernel:k2:metric:log_M_0_0, ..., kernel:k2:metric:log_M_{d-1}_{d-1} ]

with frozen entries (fit_mean / fit_white_noise False) removed.  All kernel and
noise parameters are natural logs; ``metric`` is a squared length scale.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular

__all__ = ["sqexp_kernel", "stationary_kernel", "OracleGP", "NotPositiveDefinite"]


class NotPositiveDefinite(np.linalg.LinAlgError):
    pass


def stationary_kernel(x1, x2, log_amp, log_M, kernel="ExpSquaredKernel", log_alpha=1.0):
    """amp * f(r2) for the four kernels init_gp offers (alabi/core.py:1000-1014), george definitions:
    ExpSquared exp(-r2/2); Matern32 (1+s)exp(-s), s=sqrt(3 r2); Matern52 (1+s+s^2/3)exp(-s), s=sqrt(5 r2);
    RationalQuadratic (1 + r2/(2 alpha))^-alpha."""
    x1 = np.atleast_2d(np.asarray(x1, dtype=np.float64))
    x2 = np.atleast_2d(np.asarray(x2, dtype=np.float64))
    inv_m = np.exp(-np.asarray(log_M, dtype=np.float64))
    r2 = np.zeros((x1.shape[0], x2.shape[0]))
    for k in range(x1.shape[1]):
        diff = x1[:, k][:, None] - x2[:, k][None, :]
        r2 += diff * diff * inv_m[k]
    if kernel == "ExpSquaredKernel":
        f = np.exp(-0.5 * r2)
    elif kernel == "Matern32Kernel":
        s = np.sqrt(3.0 * r2); f = (1.0 + s) * np.exp(-s)
    elif kernel == "Matern52Kernel":
        s = np.sqrt(5.0 * r2); f = (1.0 + s + s * s / 3.0) * np.exp(-s)
    elif kernel == "RationalQuadraticKernel":
        a = np.exp(log_alpha); f = (1.0 + 0.5 * r2 / a) ** (-a)
    else:
        raise ValueError(kernel)
    return np.exp(log_amp) * f


def sqexp_kernel(x1, x2, log_amp, log_M):
    """k(x,x') = exp(log_amp) * exp(-0.5 * sum_i (x_i-x'_i)^2 / exp(log_M_i)).

    Restates ``ConstantKernel * ExpSquaredKernel(metric=diag)`` as built by the
    reference at alabi/core.py:1000 and alabi/gp_utils.py:230-231 (``kernel *= var(y)``).
    """
    x1 = np.atleast_2d(np.asarray(x1, dtype=np.float64))
    x2 = np.atleast_2d(np.asarray(x2, dtype=np.float64))
    inv_m = np.exp(-np.asarray(log_M, dtype=np.float64))
    r2 = np.zeros((x1.shape[0], x2.shape[0]))
    for k in range(x1.shape[1]):
        diff = x1[:, k][:, None] - x2[:, k][None, :]
        r2 += diff * diff * inv_m[k]
    return np.exp(log_amp) * np.exp(-0.5 * r2)


class OracleGP:
    """Restatement of the george.GP protocol alabi uses (SURVEY.md section 8b seam #1)."""

    def __init__(self, ndim, mean=0.0, log_white_noise=-12.0, log_amp=0.0, log_M=None,
                 fit_mean=True, fit_white_noise=True, kernel="ExpSquaredKernel", log_alpha=1.0):
        self.kernel_name = kernel
        self.log_alpha = float(log_alpha)
        self.ndim = int(ndim)
        self.mean = float(mean)
        self.log_white_noise = float(log_white_noise)
        self.log_amp = float(log_amp)
        self.log_M = np.zeros(self.ndim) if log_M is None else np.array(log_M, dtype=np.float64)
        self.fit_mean = bool(fit_mean)
        self.fit_white_noise = bool(fit_white_noise)
        self._x = None
        self._L = None
        self._alpha = None
        self._y = None

    # ---- parameter vector protocol (george ModelSet) -------------------------------
    def get_parameter_names(self, include_frozen=False):
        names = []
        if self.fit_mean or include_frozen:
            names.append("mean:value")
        if self.fit_white_noise or include_frozen:
            names.append("white_noise:value")
        names.append("kernel:k1:log_constant")
        if self.kernel_name == "RationalQuadraticKernel":
            names.append("kernel:k2:log_alpha")
        names += [f"kernel:k2:metric:log_M_{i}_{i}" for i in range(self.ndim)]
        return tuple(names)

    def get_parameter_vector(self, include_frozen=False):
        v = []
        if self.fit_mean or include_frozen:
            v.append(self.mean)
        if self.fit_white_noise or include_frozen:
            v.append(self.log_white_noise)
        v.append(self.log_amp)
        if self.kernel_name == "RationalQuadraticKernel":
            v.append(self.log_alpha)
        v += list(self.log_M)
        return np.array(v, dtype=np.float64)

    def set_parameter_vector(self, p, include_frozen=False):
        p = np.asarray(p, dtype=np.float64).ravel()
        i = 0
        if self.fit_mean or include_frozen:
            self.mean = float(p[i]); i += 1
        if self.fit_white_noise or include_frozen:
            self.log_white_noise = float(p[i]); i += 1
        self.log_amp = float(p[i]); i += 1
        if self.kernel_name == "RationalQuadraticKernel":
            self.log_alpha = float(p[i]); i += 1
        self.log_M = p[i:i + self.ndim].copy()
        if self.log_M.size != self.ndim:
            raise ValueError("parameter vector has the wrong length")
        self._alpha = None

    def _k(self, a, b):
        return stationary_kernel(a, b, self.log_amp, self.log_M, self.kernel_name, self.log_alpha)

    # ---- compute / predict (george BasicSolver semantics) ---------------------------
    def get_matrix(self, x):
        """K = k(X,X) + exp(white_noise) * I  (reference call site: core.py:1158)."""
        K = self._k(x, x)
        K[np.diag_indices_from(K)] += np.exp(self.log_white_noise)
        return K

    def compute(self, x):
        x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float64)))
        K = self.get_matrix(x)
        try:
            L = cholesky(K, lower=True, overwrite_a=True, check_finite=False)
        except np.linalg.LinAlgError as e:
            raise NotPositiveDefinite(str(e))
        if not np.all(np.isfinite(np.diag(L))):
            raise NotPositiveDefinite("non-finite pivot")
        self._x = x
        self._L = L
        self._alpha = None
        self._y = None
        self.log_determinant = 2.0 * np.sum(np.log(np.diag(L)))
        return self

    def recompute(self):
        return self.compute(self._x)

    def _compute_alpha(self, y):
        y = np.asarray(y, dtype=np.float64).ravel()
        if self._alpha is None or self._y is None or not np.array_equal(y, self._y):
            self._y = y.copy()
            self._alpha = cho_solve((self._L, True), y - self.mean, check_finite=False)
        return self._alpha

    def predict(self, y, t, return_var=False, return_cov=False):
        """mu* = K* alpha + m ; var* = k** - sum(K*^T o K^-1 K*^T)  (core.py:85, :1601).

        White noise is on the diagonal of K only and is NOT added to var*.
        """
        alpha = self._compute_alpha(y)
        xs = np.atleast_2d(np.asarray(t, dtype=np.float64))
        Kxs = self._k(xs, self._x)
        mu = Kxs @ alpha + self.mean
        if not (return_var or return_cov):
            return mu
        KinvKxs = cho_solve((self._L, True), Kxs.T, check_finite=False)
        if return_var:
            var = np.full(xs.shape[0], np.exp(self.log_amp))
            var -= np.sum(Kxs.T * KinvKxs, axis=0)
            return mu, var
        cov = self._k(xs, xs) - Kxs @ KinvKxs
        return mu, cov

    def predict_var_halfsolve(self, y, t):
        """Same quantity through one triangular solve: var* = k** - ||L^-1 k*||^2.

        This is the form the HIP kernel evaluates; kept here to bound the algebraic
        difference between the two formulations on ill-conditioned K.
        """
        alpha = self._compute_alpha(y)
        xs = np.atleast_2d(np.asarray(t, dtype=np.float64))
        Kxs = self._k(xs, self._x)
        V = solve_triangular(self._L, Kxs.T, lower=True, check_finite=False)
        return Kxs @ alpha + self.mean, np.exp(self.log_amp) - np.sum(V * V, axis=0)

    def log_likelihood(self, y, quiet=True):
        """-0.5 r^T K^-1 r - 0.5 log|K| - N/2 log 2pi  (core.py:1248, gp_utils.py:139)."""
        y = np.asarray(y, dtype=np.float64).ravel()
        r = y - self.mean
        alpha = self._compute_alpha(y)
        ll = -0.5 * (len(r) * np.log(2.0 * np.pi) + self.log_determinant) - 0.5 * float(r @ alpha)
        return ll if np.isfinite(ll) else -np.inf

    def grad_log_likelihood(self, y, quiet=True):
        """d logL / d p over the unfrozen vector (core.py:1261, gp_utils.py:165).

        A = alpha alpha^T - K^-1 ; dL/dp = 0.5 tr(A dK/dp) ; dL/dmean = sum(alpha).
        """
        y = np.asarray(y, dtype=np.float64).ravel()
        alpha = self._compute_alpha(y)
        n = len(y)
        Kinv = cho_solve((self._L, True), np.eye(n), check_finite=False)
        A = np.outer(alpha, alpha) - Kinv
        Kk = sqexp_kernel(self._x, self._x, self.log_amp, self.log_M)
        g = []
        if self.fit_mean:
            g.append(float(np.sum(alpha)))
        if self.fit_white_noise:
            g.append(0.5 * float(np.trace(A)) * np.exp(self.log_white_noise))
        g.append(0.5 * float(np.sum(A * Kk)))
        inv_m = np.exp(-self.log_M)
        for k in range(self.ndim):
            diff = self._x[:, k][:, None] - self._x[:, k][None, :]
            g.append(0.5 * float(np.sum(A * Kk * (0.5 * diff * diff * inv_m[k]))))
        return np.array(g)

    def get_inverse(self):
        """solver.get_inverse() (reference: utility.py:610)."""
        return cho_solve((self._L, True), np.eye(self._L.shape[0]), check_finite=False)
