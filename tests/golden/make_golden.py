#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.npz by EXECUTING the reference's own code.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference package cannot be imported as ``alabi`` here (george / emcee / skopt are
not installed), but its pure NumPy/SciPy modules load by file
path once inert placeholder modules named ``skopt*`` are registered (they are only
touched by ``prior_sampler``, which is not called).  Nothing from the reference is
copied: this script stores INPUTS and the OUTPUTS the reference functions returned.

Vectors written (all float64 unless noted):
  util_*      : (mu, var, theta, bounds, y_best) grids -> bape / agp / jones values,
                including the edge branches (var <= 0, theta on / outside the boundary)
  lse_*       : logsubexp(x1, x2)
  lnprior_*   : lnprior_uniform
  ptu_*       : prior_transform_uniform (1-D and 2-D input)
  reg_*       : gp_utils.regularization_term / regularization_gradient
  burn_*      : mcmc_utils.estimate_burnin on a stub sampler with a fixed tau
  bench_*     : benchmarks.py likelihoods at fixed points
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/alabi"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _placeholders():
    for name in ["skopt", "skopt.space", "skopt.space.space", "skopt.sampler"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skopt.space"].Space = object
    sys.modules["skopt.space.space"].Real = object
    for n in ["Sobol", "Lhs", "Halton", "Hammersly", "Grid"]:
        setattr(sys.modules["skopt.sampler"], n, object)
    # gp_utils imports `george` and `alabi.utility` at module import; neither is used by
    # the two regulariser functions evaluated here.
    sys.modules.setdefault("george", types.ModuleType("george"))
    sys.modules["george"].kernels = types.ModuleType("george.kernels")
    sys.modules.setdefault("george.kernels", sys.modules["george"].kernels)


def main():
    warnings.filterwarnings("ignore")
    _placeholders()
    ut = _load("ref_utility", f"{REF}/utility.py")
    alabi_pkg = types.ModuleType("alabi")
    alabi_pkg.utility = ut
    sys.modules.setdefault("alabi", alabi_pkg)
    sys.modules.setdefault("alabi.utility", ut)
    try:
        gpu = _load("ref_gp_utils", f"{REF}/gp_utils.py")
    except Exception as e:  # ordinary import error, not a denial
        print("gp_utils did not load:", repr(e))
        gpu = None
    mcu = _load("ref_mcmc_utils", f"{REF}/mcmc_utils.py")
    bm = _load("ref_benchmarks", f"{REF}/benchmarks.py")

    rng = np.random.RandomState(20261003)
    out = {}

    # ---------------- acquisition functions ----------------
    d = 3
    bounds = np.array([[0.0, 1.0], [-2.0, 2.0], [10.0, 11.0]])
    n = 400
    theta = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * rng.rand(n, d)
    mu = rng.normal(0.0, 30.0, n)
    var = np.exp(rng.uniform(-40.0, 3.0, n))
    # edge branches
    var[:20] = 0.0
    var[20:40] = -np.exp(rng.uniform(-30, -5, 20))          # cancellation-negative variance
    var[40:50] = np.exp(rng.uniform(4.0, 6.5, 10))           # large variance
    theta[50:60, 0] = bounds[0, 0]                           # exactly on the lower face
    theta[60:70, 1] = bounds[1, 1]                           # exactly on the upper face
    theta[70:80, 2] = bounds[2, 1] + rng.rand(10)            # outside
    theta[80:90, 0] = bounds[0, 0] - rng.rand(10)            # outside
    mu[90:95] = np.array([1e300, -1e300, 0.0, 1e-300, -0.0])
    y_best = 1.7
    bape = np.empty(n); agp = np.empty(n); jones = np.empty(n)
    for i in range(n):
        pg = (lambda m, v: (lambda x: (np.array([m]), np.array([v]))))(mu[i], var[i])
        bape[i] = ut.bape_utility(theta[i], pg, bounds)
        agp[i] = ut.agp_utility(theta[i], pg, bounds)
        jones[i] = ut.jones_utility(theta[i], pg, bounds, y_best)
    out.update(util_theta=theta, util_bounds=bounds, util_mu=mu, util_var=var,
               util_y_best=np.array(y_best), util_bape=bape, util_agp=agp, util_jones=jones)

    # ---------------- logsubexp ----------------
    x1 = np.concatenate([rng.uniform(-50, 50, 100), [0.0, 1.0, -1.0, 700.0, 1e-20]])
    x2 = np.concatenate([rng.uniform(-50, 50, 100), [0.0, 1.0, -2.0, 0.0, 0.0]])
    out.update(lse_x1=x1, lse_x2=x2, lse_out=np.array([ut.logsubexp(a, b) for a, b in zip(x1, x2)]))

    # ---------------- lnprior_uniform ----------------
    pts = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * rng.uniform(-0.2, 1.2, (200, d))
    pts[:5, 0] = bounds[0, 0]; pts[5:10, 2] = bounds[2, 1]
    out.update(lnprior_x=pts, lnprior_bounds=bounds,
               lnprior_out=np.array([ut.lnprior_uniform(p, bounds) for p in pts]))
    b1 = np.array([[-2.0, 1.0]])
    x1d = rng.uniform(-3, 2, 50)
    out.update(lnprior1_x=x1d, lnprior1_bounds=b1,
               lnprior1_out=np.array([ut.lnprior_uniform(v, b1) for v in x1d]))

    # ---------------- prior_transform_uniform ----------------
    u1 = rng.rand(d); u2 = rng.rand(17, d)
    out.update(ptu_bounds=bounds, ptu_u1=u1, ptu_out1=ut.prior_transform_uniform(u1, bounds),
               ptu_u2=u2, ptu_out2=ut.prior_transform_uniform(u2, bounds))

    # ---------------- regulariser ----------------
    if gpu is not None:
        hp = rng.uniform(-3, 3, (25, 13)); li = np.arange(3, 13)
        out.update(reg_hp=hp, reg_idx=li,
                   reg_term=np.array([gpu.regularization_term(h, li) for h in hp]),
                   reg_grad=np.array([gpu.regularization_gradient(h, li) for h in hp]),
                   reg_term_k=np.array([gpu.regularization_term(h, li, amp_0=0.5, mu_0=0.3, sigma_0=1.5) for h in hp]),
                   reg_grad_k=np.array([gpu.regularization_gradient(h, li, amp_0=0.5, mu_0=0.3, sigma_0=1.5) for h in hp]))

    # ---------------- estimate_burnin ----------------
    taus = [np.array([113.5, 98.2]), np.array([3.1, 1.2, 7.9]), np.array([0.4, 0.9]),
            np.array([np.nan, 40.0, 12.5])]
    ib = []; it = []
    for tau in taus:
        class _S:  # the only member estimate_burnin touches
            def get_autocorr_time(self, tol=0, _t=tau):
                return _t
        a, b = mcu.estimate_burnin(_S())
        ib.append(a); it.append(b)
    out.update(burn_tau=np.array([np.pad(t, (0, 3 - len(t)), constant_values=-1.0) for t in taus]),
               burn_ntau=np.array([len(t) for t in taus]), burn_iburn=np.array(ib), burn_ithin=np.array(it))

    # ---------------- benchmark likelihoods ----------------
    p2 = rng.uniform(-5, 5, (40, 2))
    out.update(bench_rosen_x=p2, bench_rosen=np.array([bm.rosenbrock_fn(p) for p in p2]))
    p2s = rng.uniform(-6, 6, (40, 2))
    out.update(bench_shells_x=p2s, bench_shells=np.array([bm.gaussian_shells_fn(p) for p in p2s]))
    pe = rng.rand(40, 2)
    out.update(bench_eggbox_x=pe, bench_eggbox=np.array([bm.eggbox_fn(p) for p in pe]))
    pm = rng.uniform(0, 5, (40, 2))
    out.update(bench_multimodal_x=pm, bench_multimodal=np.array([bm.multimodal_fn(p) for p in pm]))
    pt = rng.uniform(-2, 1, 40)
    out.update(bench_test1d_x=pt, bench_test1d=bm.test1d_fn(pt))
    pg = rng.rand(40, 2)
    out.update(bench_gauss2d_x=pg, bench_gauss2d=np.array([bm.gaussian_2d_fn(p) for p in pg]))
    # random_gaussian_covariance: recipe uses the global NumPy RNG; pin it via the seed
    np.random.seed(2)
    cov10 = bm.random_gaussian_covariance(10)
    out.update(bench_cov10_seed=np.array(2), bench_cov10=cov10)
    # rosenbrock_nd (Pagani et al.) on a fixed (a, b)
    a = 1.0 / 20.0
    b = np.ones((3, 2)) * (100.0 / 20.0)
    xnd = rng.uniform(-2, 2, (10, (b.shape[0] - 1) * b.shape[1] + 1))
    out.update(bench_rnd_a=np.array(a), bench_rnd_b=b, bench_rnd_x=xnd, bench_rnd=bm.rosenbrock_nd(xnd, a, b))

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
