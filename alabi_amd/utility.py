"""Host-side mirror of the parts of alabi/utility.py that sit on the hot path's boundary.

* scalers ``no_scaler`` / ``log_scaler`` / ``nlog_scaler`` (alabi/utility.py:45-72),
* ``prior_sampler`` (utility.py:79-199; skopt is replaced by NumPy / scipy.stats.qmc),
* ``lnprior_uniform`` / ``prior_transform_uniform`` (utility.py:218-367), ``logsubexp`` (:489-504),
* the acquisition functions ``bape_utility`` / ``agp_utility`` / ``jones_utility`` with the
  reference's one-point signature ``f(theta, predict_gp, bounds)`` (utility.py:629-946),
* ``minimize_objective`` multistart (utility.py:969-1163),
* and the batched device form ``utility_scan`` (HIP: predict-variance + epilogue + arg-min),
  which is what ``SurrogateModel.find_next_point`` uses by default.
"""
from __future__ import annotations

import ctypes as C
import time
import warnings

import numpy as np
import torch
from scipy.optimize import minimize
from scipy.stats import norm, qmc
from sklearn.preprocessing import FunctionTransformer

from . import _lib

__all__ = ["agp_utility", "bape_utility", "jones_utility", "assign_utility", "minimize_objective",
           "prior_sampler", "lnprior_uniform", "lnprior_normal", "prior_transform_uniform", "logsubexp",
           "NewFunctionTransformer", "nlog_scaler", "log_scaler", "no_scaler",
           "utility_scan", "utility_eval_device", "grad_gp_mean_prediction", "grad_gp_var_prediction",
           "grad_agp_utility", "grad_bape_utility", "utility_value_and_grad", "polish_point"]


class NewFunctionTransformer(FunctionTransformer):
    """FunctionTransformer carrying a printable name (utility.py:45-58)."""

    def __init__(self, name="scaler", func=None, inverse_func=None, *, validate=False, accept_sparse=False,
                 check_inverse=True, feature_names_out=None, kw_args=None, inv_kw_args=None):
        super().__init__(func=func, inverse_func=inverse_func, validate=validate, accept_sparse=accept_sparse,
                         check_inverse=check_inverse, feature_names_out=feature_names_out, kw_args=kw_args,
                         inv_kw_args=inv_kw_args)
        self.name = name

    def __str__(self):
        return self.name

    __repr__ = __str__


def _nlog(x): return np.log10(-x)
def _nlog_inv(x): return -10 ** x
def _log(x): return np.log10(x)
def _log_inv(x): return 10 ** x
def _ident(x): return x


nlog_scaler = NewFunctionTransformer(name="nlog_scaler", func=_nlog, inverse_func=_nlog_inv)
log_scaler = NewFunctionTransformer(name="log_scaler", func=_log, inverse_func=_log_inv)
no_scaler = NewFunctionTransformer(name="no_scaler", func=_ident, inverse_func=_ident)


def prior_sampler(bounds=None, nsample=1, sampler="uniform", random_state=None):
    """Samples in the box ``bounds`` -> array (nsample, ndim)  (utility.py:79-199).

    uniform: NumPy RandomState (time-seeded when random_state is None, as the reference);
    sobol / lhs / halton: scipy.stats.qmc (the reference used skopt, not installed here).
    """
    b = np.asarray(bounds, dtype=np.float64)
    ndim = len(b)
    nsample = int(nsample)
    if random_state is None:
        random_state = int(time.time() * 1000000) % (2 ** 32)
    lo, span = b[:, 0], b[:, 1] - b[:, 0]
    if sampler == "uniform":
        rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
        u = rs.rand(nsample, ndim)
    elif sampler == "sobol":
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            u = qmc.Sobol(d=ndim, scramble=False).random(nsample + 1)[1:]
    elif sampler == "lhs":
        u = qmc.LatinHypercube(d=ndim, seed=random_state if not isinstance(random_state, np.random.RandomState) else None).random(nsample)
    elif sampler == "halton":
        u = qmc.Halton(d=ndim, scramble=False).random(nsample + 1)[1:]
    elif sampler == "grid":
        per = max(int(np.floor(nsample ** (1.0 / ndim))), 1)
        axes = [np.linspace(0.0, 1.0, per) for _ in range(ndim)]
        u = np.stack([m.ravel() for m in np.meshgrid(*axes, indexing="ij")], axis=1)[:nsample]
    else:
        raise ValueError(f"Sampler method '{sampler}' not implemented. Valid options for 'sampler' are: "
                         "uniform, sobol, lhs, halton, grid.")
    return lo + span * u


def lnprior_uniform(x, bounds):
    """0.0 strictly inside the open box, -inf otherwise (utility.py:218-275)."""
    b = np.asarray(bounds, dtype=np.float64)
    ndim = len(b)
    x = np.array([x]).ravel() if ndim == 1 else np.array(x).squeeze()
    inside = True
    for i in range(ndim):
        inside = inside and bool((x[i] > b[i][0]) and (x[i] < b[i][1]))
    return 0.0 if inside else -np.inf


def lnprior_normal(x, bounds, data):
    """Uniform box plus independent normal priors on the coordinates whose ``data[i] = (mean, std)`` is not
    ``(None, None)`` (utility.py:370-378)."""
    lnp = lnprior_uniform(x, bounds)
    for ii in range(len(x)):
        if data[ii][0] is not None:
            lnp += norm.logpdf(x[ii], data[ii][0], data[ii][1])
    return lnp


def prior_transform_uniform(theta, bounds):
    """Unit hypercube -> box (utility.py:278-367)."""
    theta = np.asarray(theta, dtype=float)
    b = np.asarray(bounds, dtype=float)
    if theta.ndim not in (1, 2):
        raise ValueError(f"theta must be 1D or 2D array, got {theta.ndim}D array with shape {theta.shape}")
    if theta.shape[-1] != len(b):
        raise ValueError(f"Bounds length ({len(b)}) must match theta dimensions ({theta.shape[-1]})")
    return (b[:, 1] - b[:, 0]) * theta + b[:, 0]


def logsubexp(x1, x2):
    """log(exp(x1) - exp(x2)), -inf if x1 <= x2 (utility.py:489-504)."""
    if x1 <= x2:
        return -np.inf
    return x1 + np.log(1.0 - np.exp(x2 - x1))


def _mu_var(predict_gp, theta):
    mu, var = predict_gp(np.asarray(theta, dtype=np.float64).reshape(1, -1))
    return float(np.asarray(mu).ravel()[0]), float(np.asarray(var).ravel()[0])


def bape_utility(theta, predict_gp, bounds):
    """-[2 mu + var + log(exp(var) - 1)]; +inf outside the box (utility.py:729-810)."""
    theta = np.asarray(theta).flatten()
    if not np.isfinite(lnprior_uniform(theta, bounds)):
        return np.inf
    mu, var = _mu_var(predict_gp, theta)
    return float(-((2.0 * mu + var) + logsubexp(var, 0.0)))


def agp_utility(theta, predict_gp, bounds):
    """-[mu + 0.5 log(2 pi e var)]; +inf outside the box (utility.py:629-701)."""
    theta = np.asarray(theta)
    if not np.isfinite(lnprior_uniform(theta, bounds)):
        return np.inf
    mu, var = _mu_var(predict_gp, theta)
    with np.errstate(invalid="ignore", divide="ignore"):
        return float(-(mu + 0.5 * np.log(2.0 * np.pi * np.e * var)))


def jones_utility(theta, predict_gp, bounds, y_best, zeta=0.01):
    """Negative expected improvement; 0.0 when the predictive std is not positive (utility.py:853-946)."""
    theta = np.asarray(theta)
    if not np.isfinite(lnprior_uniform(theta, bounds)):
        return np.inf
    mu, var = _mu_var(predict_gp, theta)
    with np.errstate(invalid="ignore"):
        std = np.sqrt(var)
    if not (std > 0):
        return 0.0
    z = (mu - y_best - zeta) / std
    return float(-((mu - y_best - zeta) * norm.cdf(z) + std * norm.pdf(z)))


def _predict_grad(xs, gp):
    xs = np.asarray(xs, dtype=np.float64).reshape(1, -1)
    mu, var, dmu, dvar = gp.predict_grad_device(gp._y, xs)
    return float(mu[0]), float(var[0]), dmu[0].cpu().numpy(), dvar[0].cpu().numpy()


def grad_gp_mean_prediction(xs, gp):
    """grad mu(x) = (dk/dx)^T alpha (utility.py:558-583), closed-form kernel derivative on the GPU
    (the reference differences the kernel numerically with step 1e-6, utility.py:511-555)."""
    return _predict_grad(xs, gp)[2]


def grad_gp_var_prediction(xs, gp):
    """grad var(x) = -2 (dk/dx)^T K^-1 k (utility.py:586-623) from the cached L^-1, no explicit K^-1."""
    return _predict_grad(xs, gp)[3]


def grad_agp_utility(theta, gp, bounds):
    """-(d_mu + 0.5 d_var), the reference's expression (utility.py:704-726: d_var is NOT divided by var there);
    inf[d] outside the box."""
    theta = np.asarray(theta, dtype=np.float64).flatten()
    if not np.isfinite(lnprior_uniform(theta, bounds)):
        return np.full(len(theta), np.inf)
    _, _, d_mu, d_var = _predict_grad(theta, gp)
    return (-(d_mu + 0.5 * d_var)).flatten()


def grad_bape_utility(theta, gp, bounds):
    """-2 d_mu - (1 + e^var / (e^var - 1)) d_var (utility.py:813-850); inf[d] outside the box.
    e^var / (e^var - 1) is evaluated as 1 / (1 - e^-var): the same number, but finite for var > 709 where the
    reference's literal expression is inf / inf = NaN."""
    theta = np.asarray(theta, dtype=np.float64).flatten()
    if not np.isfinite(lnprior_uniform(theta, bounds)):
        return np.full(len(theta), np.inf)
    _, var, d_mu, d_var = _predict_grad(theta, gp)
    with np.errstate(all="ignore"):
        return -2.0 * d_mu - (1.0 - 1.0 / np.expm1(-var)) * d_var


def assign_utility(algorithm):
    """name -> (utility, grad_utility)  (utility.py:949-966); jones has no gradient in the reference either."""
    table = {"bape": (bape_utility, grad_bape_utility), "agp": (agp_utility, grad_agp_utility),
             "jones": (jones_utility, None)}
    if algorithm not in table:
        print(f"ERROR: Unknown utility function: {algorithm}. Defaulting to BAPE.")
        return table["bape"]
    return table[algorithm]


# ---- device batch forms ---------------------------------------------------------------------

def utility_eval_device(algorithm, theta, bounds, mu, var, y_best=0.0):
    """Epilogue only: u[M] from caller-supplied (mu, var) device tensors."""
    from .gp import _to_dev
    th = _to_dev(theta, 2)
    mu = _to_dev(mu); var = _to_dev(var)
    m, d = int(th.shape[0]), int(th.shape[1])
    u = torch.empty(m, dtype=torch.float64, device=th.device)
    b = np.ascontiguousarray(np.asarray(bounds, dtype=np.float64).reshape(d, 2))
    st = _lib.lib().alabi_utility_eval(_lib.UTILITY_CODES[algorithm], _lib.ptr(th), m, d, _lib.host_doubles(b.ravel()),
                                       float(y_best), _lib.ptr(mu), _lib.ptr(var), _lib.ptr(u), _lib.current_stream())
    _lib.check(st, "alabi_utility_eval")
    return u


def utility_scan(gp, y, theta, bounds, algorithm="bape", y_best=0.0, return_all=False, best_on_device=False):
    """Evaluate the acquisition function on M candidates on the GPU and return the arg-min.

    Returns (theta_best[d] numpy, u_best, index) and, with return_all, also (u, mu, var) device tensors.
    best_on_device: theta_best stays a device tensor (a row of ``theta``; NaN row if no candidate is finite) -- the zoom stages of
    find_next_point chain several scans and need the point on the host only once, at the end.
    Non-finite utilities (outside the box, var <= 0 for bape, ...) never win, as in
    utility.minimize_objective (utility.py:1149-1163).  index is -1 if no candidate is finite.
    """
    from .gp import _to_dev
    th = _to_dev(theta, 2)
    m, d = int(th.shape[0]), int(th.shape[1])
    gp._require_computed(); gp._set_y(y)          # K is factorised and alpha matches y (no launch when y is unchanged)
    b = np.ascontiguousarray(np.asarray(bounds, dtype=np.float64).reshape(d, 2))
    u = mu = var = None
    if return_all:
        u = torch.empty(m, dtype=torch.float64, device=th.device)
        mu = torch.empty_like(u); var = torch.empty_like(u)
    best_val = C.c_double(0.0)
    best_idx = C.c_longlong(-1)
    st = _lib.lib().alabi_utility_scan(gp.handle, _lib.UTILITY_CODES[algorithm], _lib.ptr(th), m,
                                       _lib.host_doubles(b.ravel()), float(y_best), _lib.ptr(u), _lib.ptr(mu),
                                       _lib.ptr(var), C.byref(best_val), C.byref(best_idx), _lib.current_stream())
    _lib.check(st, "alabi_utility_scan")
    idx = int(best_idx.value)
    if best_on_device:
        best_theta = th[idx] if idx >= 0 else torch.full((d,), float("nan"), dtype=torch.float64, device=th.device)
    else:
        best_theta = th[idx].cpu().numpy() if idx >= 0 else np.full(d, np.nan)
    out = (best_theta, float(best_val.value), idx)
    return out + (u, mu, var) if return_all else out


def utility_value_and_grad(algorithm, mu, var, dmu, dvar, y_best=0.0, zeta=0.01):
    """Acquisition value and its TRUE gradient in the query point from (mu, var, dmu[d], dvar[d]) by the chain rule:
    bape  u = -(2 mu + var + log(e^var - 1)),          du = -2 dmu - (1 + 1/(1 - e^-var)) dvar
    agp   u = -(mu + 0.5 log(2 pi e var)),             du = -(dmu + dvar / (2 var))
    jones u = -((mu - yb - zeta) Phi(z) + s phi(z)),   du = -(Phi(z) dmu + phi(z) dvar / (2 s)),  s = sqrt(var)
    (grad_agp_utility above keeps the reference's literal -(dmu + 0.5 dvar); this one is what an optimiser needs.)
    Non-finite values (var <= 0 ...) come back as (inf, zeros)."""
    dmu = np.asarray(dmu, dtype=np.float64); dvar = np.asarray(dvar, dtype=np.float64)
    bad = (np.inf, np.zeros_like(dmu))
    with np.errstate(all="ignore"):
        if algorithm == "bape":
            if not var > 0.0:
                return bad
            u = -((2.0 * mu + var) + logsubexp(var, 0.0))
            g = -2.0 * dmu - (1.0 - 1.0 / np.expm1(-var)) * dvar
        elif algorithm == "agp":
            if not var > 0.0:
                return bad
            u = -(mu + 0.5 * np.log(2.0 * np.pi * np.e * var))
            g = -(dmu + 0.5 * dvar / var)
        elif algorithm == "jones":
            if not var > 0.0:
                return bad
            sd = np.sqrt(var)
            z = (mu - y_best - zeta) / sd
            u = -((mu - y_best - zeta) * norm.cdf(z) + sd * norm.pdf(z))
            g = -(norm.cdf(z) * dmu + norm.pdf(z) * dvar / (2.0 * sd))
        else:
            raise ValueError(algorithm)
    if not (np.isfinite(u) and np.all(np.isfinite(g))):
        return bad
    return float(u), g


def polish_point(gp, y, theta0, bounds, algorithm="bape", y_best=0.0, maxiter=30, method="native"):
    """Bound-constrained quasi-Newton descent from ``theta0`` on the acquisition function with value and gradient from ONE device
    call per evaluation (alabi_gp_predict_grad_point): the continuous-optimum step of utility.py:1030-1163 on top of a batched scan.
    method "native" (default): the library's projected L-BFGS next to the kernels (alabi_utility_polish) -- through scipy an
    evaluation cost 80 us at N = 100, 31 us of it device work, and the polish was 70 % of an active-learning iteration;
    method "scipy": scipy's L-BFGS-B around predict_grad_host (the earlier implementation, kept for comparison).
    The box is shrunk by 1e-9 of its width because the reference's objective is +inf ON the boundary (utility.py:268-275).
    Returns (theta, u); never worse than the start."""
    if method == "native":
        gp._require_computed(); gp._set_y(y)
        d = gp.ndim
        x0n = np.ascontiguousarray(np.asarray(theta0, dtype=np.float64).reshape(d))
        bn = np.ascontiguousarray(np.asarray(bounds, dtype=np.float64).reshape(d, 2))
        xo = np.empty(d, dtype=np.float64)
        uo = C.c_double(0.0); ne = C.c_int(0)
        st = _lib.lib().alabi_utility_polish(gp.handle, _lib.UTILITY_CODES[algorithm], x0n.ctypes.data, bn.ctypes.data, float(y_best),
                                             int(maxiter), xo.ctypes.data, C.byref(uo), C.byref(ne), _lib.current_stream())
        _lib.check(st, "alabi_utility_polish")
        gp._last_polish_nevals = int(ne.value)                    # (diagnostics: evaluations the polish made)
        return xo, float(uo.value)
    b = np.asarray(bounds, dtype=np.float64)
    eps = 1e-9 * (b[:, 1] - b[:, 0])
    box = np.column_stack([b[:, 0] + eps, b[:, 1] - eps])
    x0 = np.clip(np.asarray(theta0, dtype=np.float64).ravel(), box[:, 0], box[:, 1])

    def fun(x):
        mu, var, dmu, dvar = gp.predict_grad_host(y, x.reshape(1, -1))
        u, g = utility_value_and_grad(algorithm, float(mu[0]), float(var[0]), dmu[0], dvar[0], y_best)
        return (u, g) if np.isfinite(u) else (1e100, np.zeros_like(x))

    u0, _ = fun(x0)
    res = minimize(fun, x0, jac=True, method="L-BFGS-B", bounds=box, options={"maxiter": int(maxiter), "ftol": 1e-12, "gtol": 1e-8})
    if np.all(np.isfinite(res.x)) and np.isfinite(res.fun) and res.fun < u0:
        return res.x, float(res.fun)
    return x0, float(u0)


# ---- multistart local optimiser (reference semantics, one point per objective call) ----------

def minimize_objective_single(idx, obj_fn, bounds, starting_point, method, options, grad_obj_fn=None):
    res = minimize(fun=obj_fn, x0=np.array(starting_point).flatten(), jac=grad_obj_fn, bounds=bounds,
                   method=method, options=options)
    x_opt, f_opt = res.x, res.fun
    if not (np.all(np.isfinite(x_opt)) and np.all(np.isfinite(f_opt))):
        print("Warning: Acquisition function optimization infinite fail", x_opt, f_opt)
        return np.nan, np.nan
    if not np.isfinite(lnprior_uniform(x_opt, bounds)):
        print("Warning: Acquisition function optimization prior fail", x_opt)
        return np.nan, np.nan
    if res.nit > 5:
        return x_opt, f_opt
    print(f"Warning: Aquisition function ran for {res.nit} iterations. Optimizer success: {res.success}")
    return (np.nan, np.nan) if res.nit <= 1 else (x_opt, f_opt)


def minimize_objective(obj_fn, bounds=None, nopt=1, method="l-bfgs-b", ps=None, options=None,
                       grad_obj_fn=None, pool=None):
    """Best of ``nopt`` local minimisations from random starts (utility.py:1030-1163)."""
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    m = str(method).lower()
    if options is None:
        options = ({"maxiter": 100, "ftol": 1e-6, "gtol": 1e-5} if m == "l-bfgs-b"
                   else {"maxiter": 200, "xatol": 1e-6, "fatol": 1e-6} if m == "nelder-mead" else {"maxiter": 100})
    else:
        options = dict(options)
        for old, new in (("max_iter", "maxiter"), ("max_eval", "maxfev"), ("max_fun", "maxfun")):
            if old in options:
                options[new] = options.pop(old)
    if m == "nelder-mead":
        options["adaptive"] = True
        grad_obj_fn = None
    elif m == "l-bfgs-b":
        options.setdefault("maxcor", 10)
    starts = prior_sampler(bounds, nsample=nopt, sampler="lhs") if ps is None else ps(nsample=nopt)
    starts = np.array([np.asarray(p).flatten() for p in starts])
    results = [minimize_objective_single(i, obj_fn, bounds, starts[i], method, options, grad_obj_fn) for i in range(nopt)]
    valid = [(t, o) for t, o in results if np.all(np.isfinite(t)) and np.isfinite(o)]
    if not valid:
        print(f"Warning: All {nopt} optimization attempts failed. Returning NaN.")
        return np.nan, np.nan
    k = int(np.argmin([o for _, o in valid]))
    return valid[k]
