"""The oracle AND the product's host-side functions reproduce every golden vector that was generated
by executing the reference's own code (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import utility_oracle as uo


def _eq(a, b, rtol=1e-13, atol=0.0):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    fin = np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)])   # +-inf must match exactly
    np.testing.assert_allclose(a[fin], b[fin], rtol=rtol, atol=atol)


def _pg(m, v):
    return lambda x: (np.array([m]), np.array([v]))


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_acquisition_scalar_forms(golden, impl):
    if impl == "oracle":
        mod = uo
    else:
        from alabi_amd import utility as mod
    g = golden
    n = len(g["util_mu"])
    bape = [mod.bape_utility(g["util_theta"][i], _pg(g["util_mu"][i], g["util_var"][i]), g["util_bounds"]) for i in range(n)]
    agp = [mod.agp_utility(g["util_theta"][i], _pg(g["util_mu"][i], g["util_var"][i]), g["util_bounds"]) for i in range(n)]
    jones = [mod.jones_utility(g["util_theta"][i], _pg(g["util_mu"][i], g["util_var"][i]), g["util_bounds"],
                               float(g["util_y_best"])) for i in range(n)]
    _eq(bape, g["util_bape"]); _eq(agp, g["util_agp"]); _eq(jones, g["util_jones"], rtol=1e-12, atol=1e-300)


def test_acquisition_batch_forms(golden):
    g = golden
    for algo, key in (("bape", "util_bape"), ("agp", "util_agp"), ("jones", "util_jones")):
        u = uo.utility_batch(algo, g["util_mu"], g["util_var"], g["util_theta"], g["util_bounds"], float(g["util_y_best"]))
        _eq(u, g[key], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_logsubexp_and_priors(golden, impl):
    if impl == "oracle":
        mod = uo
    else:
        from alabi_amd import utility as mod
    g = golden
    _eq([mod.logsubexp(a, b) for a, b in zip(g["lse_x1"], g["lse_x2"])], g["lse_out"])
    _eq([mod.lnprior_uniform(p, g["lnprior_bounds"]) for p in g["lnprior_x"]], g["lnprior_out"])
    _eq([mod.lnprior_uniform(v, g["lnprior1_bounds"]) for v in g["lnprior1_x"]], g["lnprior1_out"])
    _eq(mod.prior_transform_uniform(g["ptu_u1"], g["ptu_bounds"]), g["ptu_out1"], rtol=1e-15)
    _eq(mod.prior_transform_uniform(g["ptu_u2"], g["ptu_bounds"]), g["ptu_out2"], rtol=1e-15)


def test_box_prior_batch_matches_reference(golden):
    from oracle.stretch_oracle import box_lnprior_batch
    g = golden
    _eq(box_lnprior_batch(g["lnprior_x"], g["lnprior_bounds"]), g["lnprior_out"])


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_regulariser(golden, impl):
    if impl == "oracle":
        mod = uo
    else:
        from alabi_amd import gp_utils as mod
    g = golden
    _eq([mod.regularization_term(h, g["reg_idx"]) for h in g["reg_hp"]], g["reg_term"])
    _eq([mod.regularization_gradient(h, g["reg_idx"]) for h in g["reg_hp"]], g["reg_grad"])
    _eq([mod.regularization_term(h, g["reg_idx"], amp_0=0.5, mu_0=0.3, sigma_0=1.5) for h in g["reg_hp"]], g["reg_term_k"])
    _eq([mod.regularization_gradient(h, g["reg_idx"], amp_0=0.5, mu_0=0.3, sigma_0=1.5) for h in g["reg_hp"]], g["reg_grad_k"])


def test_burnin(golden):
    from alabi_amd import mcmc_utils
    g = golden
    for row, n, ib, it in zip(g["burn_tau"], g["burn_ntau"], g["burn_iburn"], g["burn_ithin"]):
        tau = row[:n]
        assert uo.estimate_burnin_from_tau(tau) == (int(ib), int(it))

        class _S:
            def get_autocorr_time(self, tol=0, _t=tau):
                return _t
        a, b = mcmc_utils.estimate_burnin(_S())
        assert (int(a), int(b)) == (int(ib), int(it))


def test_benchmark_likelihoods(golden):
    from alabi_amd import benchmarks as bm
    g = golden
    _eq([bm.rosenbrock_fn(p) for p in g["bench_rosen_x"]], g["bench_rosen"])
    _eq([bm.gaussian_shells_fn(p) for p in g["bench_shells_x"]], g["bench_shells"])
    _eq([bm.eggbox_fn(p) for p in g["bench_eggbox_x"]], g["bench_eggbox"])
    _eq([bm.multimodal_fn(p) for p in g["bench_multimodal_x"]], g["bench_multimodal"])
    _eq(bm.test1d_fn(g["bench_test1d_x"]), g["bench_test1d"])
    _eq([bm.gaussian_2d_fn(p) for p in g["bench_gauss2d_x"]], g["bench_gauss2d"], rtol=1e-12)
    _eq(bm.rosenbrock_nd(g["bench_rnd_x"], float(g["bench_rnd_a"]), g["bench_rnd_b"]), g["bench_rnd"])
    np.random.seed(int(g["bench_cov10_seed"]))
    _eq(bm.random_gaussian_covariance(10), g["bench_cov10"], rtol=1e-12)
    # the 2-D member of the N-d shell family is the reference's function
    sh = bm.gaussian_shells_nd(2)
    _eq([sh["fn"](p) for p in g["bench_shells_x"]], g["bench_shells"])
    # gaussian_nd uses the pinned covariance recipe
    gn = bm.gaussian_nd(10, seed=int(g["bench_cov10_seed"]))
    _eq(gn["cov"], g["bench_cov10"], rtol=1e-12)


def _grad_gp(g):
    from oracle.gp_oracle import OracleGP
    gp = OracleGP(g["grad_X"].shape[1], float(g["grad_mean"]), float(g["grad_log_wn"]), float(g["grad_log_amp"]), g["grad_log_M"])
    gp.compute(g["grad_X"])
    gp._compute_alpha(g["grad_y"])
    return gp


def test_acquisition_gradients_reference_shaped(golden_grad):
    """The oracle's restatement of utility.py:511-850 (numerical kernel gradient, explicit inverse) against the arrays
    the reference's functions returned on the same GP."""
    g = golden_grad
    gp = _grad_gp(g)
    th, b = g["grad_theta"], g["grad_bounds"]
    _eq([uo.grad_gp_mean_prediction(t, gp) for t in th], g["grad_dmu"], rtol=1e-9, atol=1e-9)
    _eq([uo.grad_gp_var_prediction(t, gp) for t in th], g["grad_dvar"], rtol=1e-9, atol=1e-9)
    _eq([uo.grad_agp_utility(t, gp, b) for t in th], g["grad_agp"], rtol=1e-9, atol=1e-9)
    _eq([uo.grad_bape_utility(t, gp, b) for t in th], g["grad_bape"], rtol=1e-9, atol=1e-9)
    assert np.all(np.isinf(g["grad_bape"][-3:])) and np.all(np.isinf(g["grad_agp"][-3:]))


def test_acquisition_gradients_closed_form(golden_grad):
    """The closed form the HIP path evaluates agrees with the reference's finite-difference result to the accuracy of
    the differencing (step 1e-6: relative 1e-6 of the gradient scale)."""
    g = golden_grad
    gp = _grad_gp(g)
    mu, var, dmu, dvar = uo.analytic_predict_grad(gp, g["grad_y"], g["grad_theta"])
    sm = np.max(np.abs(g["grad_dmu"])); sv = np.max(np.abs(g["grad_dvar"]))
    assert np.max(np.abs(dmu - g["grad_dmu"])) <= 1e-6 * sm
    assert np.max(np.abs(dvar - g["grad_dvar"])) <= 1e-6 * sv
    inside = np.isfinite(g["grad_bape"][:, 0])
    with np.errstate(all="ignore"):
        e = np.exp(var)
        bape = -2.0 * dmu - (1.0 + e / (e - 1.0))[:, None] * dvar
    ok = inside & (var > 1e-9 * np.exp(float(g["grad_log_amp"])))     # at a training point var ~ 0 and the weight blows up
    assert np.max(np.abs(bape[ok] - g["grad_bape"][ok]) / (np.abs(g["grad_bape"][ok]) + np.max(np.abs(bape[ok])))) <= 1e-5
