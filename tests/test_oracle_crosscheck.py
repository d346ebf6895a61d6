"""The parts of the oracle the reference cannot pin (george / emcee are absent) are cross-checked against
independent implementations that ARE installed (SciPy LAPACK, scikit-learn's GaussianProcessRegressor)
and against analytic known answers.  See oracle/__init__.py "PARITY STATUS"."""
import numpy as np
import pytest
from scipy.linalg import cho_factor, cho_solve

from oracle.gp_oracle import NotPositiveDefinite, OracleGP, sqexp_kernel
from oracle import stretch_oracle as so
from oracle.autocorr_oracle import integrated_time as oracle_tau
from conftest import make_problem


def _gp(X, y, h):
    gp = OracleGP(X.shape[1], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    gp.compute(X)
    return gp


def test_gp_against_sklearn():
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel
    X, y, h = make_problem(120, 3, 0, log_wn=-8.0)
    gp = _gp(X, y, h)
    Xs = np.random.RandomState(1).uniform(-3, 3, (40, 3))
    mu, var = gp.predict(y, Xs, return_var=True)
    k = ConstantKernel(np.exp(h["log_amp"])) * RBF(length_scale=np.sqrt(np.exp(h["log_M"]))) + WhiteKernel(np.exp(h["log_white_noise"]))
    sk = GaussianProcessRegressor(kernel=k, optimizer=None, alpha=0.0).fit(X, y - h["mean"])
    mu_sk, sd_sk = sk.predict(Xs, return_std=True)
    np.testing.assert_allclose(mu, mu_sk + h["mean"], rtol=1e-8, atol=1e-8)
    # sklearn's predictive variance includes the WhiteKernel level; george's / alabi's does not
    np.testing.assert_allclose(var, sd_sk ** 2 - np.exp(h["log_white_noise"]), rtol=1e-6, atol=1e-8)
    # marginal likelihood
    assert abs(gp.log_likelihood(y) - sk.log_marginal_likelihood_value_) < 1e-7 * abs(gp.log_likelihood(y))


def test_gp_against_scipy_and_known_answers():
    X, y, h = make_problem(200, 4, 3, log_wn=-10.0)
    gp = _gp(X, y, h)
    K = gp.get_matrix(X)
    cf = cho_factor(K, lower=True)
    alpha = cho_solve(cf, y - h["mean"])
    np.testing.assert_allclose(gp._compute_alpha(y), alpha, rtol=1e-9)
    sign, ld = np.linalg.slogdet(K)
    assert sign > 0 and abs(gp.log_determinant - ld) < 1e-8 * abs(ld)
    # interpolation at the training points, variance ~ nugget scale
    mu, var = gp.predict(y, X, return_var=True)
    np.testing.assert_allclose(mu, y, atol=5e-4)
    assert np.all(np.abs(var) < 1e-3 * np.exp(h["log_amp"]))
    # far from the data the prior is recovered
    far = np.full((1, 4), 500.0)
    mu_f, var_f = gp.predict(y, far, return_var=True)
    assert abs(mu_f[0] - h["mean"]) < 1e-12 and abs(var_f[0] - np.exp(h["log_amp"])) < 1e-12
    # both variance formulations agree to conditioning
    mu2, var2 = gp.predict_var_halfsolve(y, X[:50] + 0.3)
    mu1, var1 = gp.predict(y, X[:50] + 0.3, return_var=True)
    np.testing.assert_allclose(mu1, mu2, rtol=1e-12)
    np.testing.assert_allclose(var1, var2, atol=1e-7 * np.exp(h["log_amp"]))


def test_gp_closed_form_two_points():
    x = np.array([[0.0], [1.0]]); y = np.array([1.0, 3.0])
    amp, m2, wn, mean = 2.0, 0.5, 1e-3, 0.25
    gp = OracleGP(1, mean, np.log(wn), np.log(amp), [np.log(m2)]).compute(x)
    k01 = amp * np.exp(-0.5 / m2)
    K = np.array([[amp + wn, k01], [k01, amp + wn]])
    xs = np.array([[0.4]])
    ks = amp * np.exp(-0.5 * (xs[0, 0] - x[:, 0]) ** 2 / m2)
    mu_ref = ks @ np.linalg.solve(K, y - mean) + mean
    var_ref = amp - ks @ np.linalg.solve(K, ks)
    mu, var = gp.predict(y, xs, return_var=True)
    assert abs(mu[0] - mu_ref) < 1e-13 and abs(var[0] - var_ref) < 1e-13


def test_gp_gradient_matches_finite_differences():
    X, y, h = make_problem(60, 2, 5, log_wn=-6.0)
    gp = _gp(X, y, h)
    p0 = gp.get_parameter_vector()
    g = gp.grad_log_likelihood(y)
    for i in range(len(p0)):
        e = np.zeros_like(p0); e[i] = 1e-6
        gp.set_parameter_vector(p0 + e); gp.recompute(); fp = gp.log_likelihood(y)
        gp.set_parameter_vector(p0 - e); gp.recompute(); fm = gp.log_likelihood(y)
        assert abs((fp - fm) / 2e-6 - g[i]) < 1e-4 * max(1.0, abs(g[i]))


def test_not_positive_definite_is_reported():
    X = np.zeros((5, 2))                      # five identical points, tiny nugget -> singular
    gp = OracleGP(2, 0.0, -80.0, 0.0, [0.0, 0.0])
    with pytest.raises(NotPositiveDefinite):
        gp.compute(X)


def test_parameter_vector_protocol():
    gp = OracleGP(3, 1.0, -12.0, 0.5, [0.1, 0.2, 0.3])
    assert gp.get_parameter_names() == ("mean:value", "white_noise:value", "kernel:k1:log_constant",
                                        "kernel:k2:metric:log_M_0_0", "kernel:k2:metric:log_M_1_1",
                                        "kernel:k2:metric:log_M_2_2")
    gp2 = OracleGP(3, fit_mean=False, fit_white_noise=False)
    assert len(gp2.get_parameter_vector()) == 4
    assert sqexp_kernel(np.zeros((1, 3)), np.zeros((1, 3)), 0.5, [0, 0, 0])[0, 0] == np.exp(0.5)


# ------------------------------------------------------------------------------ stretch move
def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32 10 rounds, ctr = key = 0  and  ctr = key = 0xffffffff
    out = so.philox4x32_10(np.zeros((1, 4), dtype=np.uint64), (0, 0))[0]
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    out = so.philox4x32_10(np.full((1, 4), 0xFFFFFFFF, dtype=np.uint64), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


def test_draws_are_a_balanced_split_and_uniform():
    W = 64
    for step in range(20):
        order, n0, u_z, partner, u_acc = so.draw_step_randoms(1234, step, W)
        assert sorted(order.tolist()) == list(range(W)) and n0 == W // 2
        assert np.all(np.diff(order[:n0]) > 0) and np.all(np.diff(order[n0:]) > 0)
        assert np.all((u_z >= 0) & (u_z < 1)) and np.all((u_acc >= 0) & (u_acc < 1))
        assert partner.min() >= 0 and partner.max() < W // 2
    # label of walker 0 is a fair coin over steps; u_z uniform
    lab = [0 in so.draw_step_randoms(7, s, 16)[0][:8] for s in range(2000)]
    assert abs(np.mean(lab) - 0.5) < 0.05
    uz = np.concatenate([so.draw_step_randoms(7, s, 16)[2] for s in range(500)])
    assert abs(uz.mean() - 0.5) < 0.02 and abs(uz.var() - 1 / 12) < 0.01
    # odd ensembles: set 0 gets the extra walker, as emcee's arange(W) % 2
    assert so.draw_step_randoms(1, 0, 7)[1] == 4


def test_array_step_reproduces_emcee_literal_step_bit_for_bit():
    rng = np.random.RandomState(42)
    W, d = 24, 3
    bounds = np.array([[-4.0, 4.0]] * d)
    A = rng.randn(d, d); P = A @ A.T + np.eye(d)

    def lnp_one(t):
        return -0.5 * t @ P @ t + so.box_lnprior_batch(t[None, :], bounds)[0]

    def lnp_batch(q):
        return np.array([lnp_one(t) for t in q])

    coords = rng.uniform(-2, 2, (W, d))
    logp = lnp_batch(coords)
    rs = np.random.RandomState(99)
    for _ in range(25):
        rec = {}
        c1, l1, a1 = so.emcee_literal_step(coords, logp, lnp_one, rs, record=rec)
        order, n0, u_z, partner, u_acc = so.literal_draws_to_arrays(rec)
        c2, l2, a2 = so.stretch_step_arrays(coords, logp, order, n0, u_z, partner, u_acc, lnp_batch)
        assert np.array_equal(c1, c2) and np.array_equal(l1, l2) and np.array_equal(a1, a2)
        coords, logp = c1, l1


def test_stretch_move_samples_a_gaussian():
    d, W = 3, 32
    cov = np.array([[1.0, 0.6, 0.0], [0.6, 2.0, -0.3], [0.0, -0.3, 0.5]])
    P = np.linalg.inv(cov)
    lnp = lambda q: -0.5 * np.einsum("ni,ij,nj->n", q, P, q)  # noqa: E731
    p0 = np.random.RandomState(0).randn(W, d)
    chain, _, nacc, _, _ = so.run_ensemble(p0, 6000, lnp, seed=5)
    flat = chain[1000:].reshape(-1, d)
    assert np.all(np.abs(flat.mean(axis=0)) < 0.12)
    np.testing.assert_allclose(np.cov(flat.T), cov, atol=0.2)
    acc = nacc.mean() / 6000
    assert 0.3 < acc < 0.8
    tau = oracle_tau(chain[1000:])
    assert np.all(tau > 1) and np.all(tau < 200)


def test_autocorr_matches_device_independent_implementation():
    from alabi_amd.mcmc_utils import integrated_time
    rng = np.random.RandomState(3)
    n, w, d = 4000, 6, 2
    x = np.zeros((n, w, d))
    for t in range(1, n):                      # AR(1), tau = (1+phi)/(1-phi) = 19
        x[t] = 0.9 * x[t - 1] + rng.randn(w, d)
    a = oracle_tau(x); b = integrated_time(x, tol=0)
    np.testing.assert_allclose(a, b, rtol=1e-10)
    assert np.all(np.abs(a - 19.0) < 6.0)
