"""The four kernels init_gp offers (alabi/core.py:1000-1014): ExpSquared, Matern-3/2, Matern-5/2, RationalQuadratic.
Every device kernel (assembly, Cholesky, predict mean / variance, acquisition scan, ensemble) is checked against the
oracle for each family."""
import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu
KERNELS = ["ExpSquaredKernel", "Matern32Kernel", "Matern52Kernel", "RationalQuadraticKernel"]


def _pair(X, y, h, kernel, log_alpha=0.4):
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    d = X.shape[1]
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], kernel=kernel, log_alpha=log_alpha)
    g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], kernel=kernel, log_alpha=log_alpha).compute(X)
    return g, o


@pytest.mark.parametrize("kernel", KERNELS)
def test_gp_parity_per_kernel(kernel):
    from oracle.gp_oracle import stationary_kernel
    X, y, h = make_problem(300, 4, 17, log_wn=-10.0)
    g, o = _pair(X, y, h, kernel)
    amp = np.exp(h["log_amp"])
    X2 = np.random.RandomState(0).uniform(-3, 3, (50, 4))
    np.testing.assert_allclose(g.kernel.get_value(X[:40], X2), stationary_kernel(X[:40], X2, h["log_amp"], h["log_M"], kernel, 0.4),
                               rtol=1e-12, atol=1e-300)
    K = o.get_matrix(X)
    L = g.solver.get_factor().cpu().numpy()
    assert np.max(np.abs(L @ L.T - K)) <= 1e-12 * np.max(np.abs(K))
    Xs = np.random.RandomState(1).uniform(-3.1, 3.1, (5000, 4))        # tile path for the mean, several variance tiles
    mu, var = g.predict(y, Xs, return_var=True)
    mu_o, var_o = o.predict(y, Xs, return_var=True)
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert np.max(np.abs(var - var_o)) <= 1e-6 * amp
    assert np.max(np.abs(g.predict(y, Xs, return_cov=False) - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert abs(g.log_likelihood(y) - o.log_likelihood(y)) <= 1e-9 * abs(o.log_likelihood(y)) + 1e-7
    assert g.get_parameter_names() == o.get_parameter_names()
    if kernel == "RationalQuadraticKernel":
        assert "kernel:k2:log_alpha" in g.get_parameter_names()
        p = g.get_parameter_vector(); p[3] = -0.7                         # change alpha through the vector protocol
        g.set_parameter_vector(p); o.set_parameter_vector(p); o.recompute()
        np.testing.assert_allclose(g.predict(y, Xs[:100], return_cov=False), o.predict(y, Xs[:100]), rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("kernel", KERNELS[1:])
def test_ensemble_per_kernel(kernel, monkeypatch):
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    X, y, h = make_problem(200, 3, 5, log_wn=-9.0)
    g, o = _pair(X, y, h, kernel)
    bounds = np.array([[-3.0, 3.0]] * 3)

    def lnp(q):
        inside = np.all((q > -3) & (q < 3), axis=1)
        out = np.full(len(q), -np.inf)
        if inside.any():
            out[inside] = o.predict(y, q[inside])
        return out

    p0 = np.random.RandomState(2).uniform(-2, 2, (20, 3))
    ref = so.run_ensemble(p0, 120, lnp, seed=9)[0]
    for stream in ("1", "0"):                                             # persistent and launch-per-half-step paths
        monkeypatch.setenv("ALABI_ENS_STREAM", stream)
        s = EnsembleSampler(20, 3, g, y, bounds, seed=9)
        s.run_mcmc(p0, 120)
        assert np.max(np.abs(s.get_chain() - ref)) < 1e-7


def test_surrogate_model_with_matern(tmp_path):
    from alabi_amd import SurrogateModel
    from alabi_amd.benchmarks import gaussian_2d
    for kernel in ("Matern52Kernel", "RationalQuadraticKernel"):
        sm = SurrogateModel(lnlike_fn=gaussian_2d["fn"], bounds=gaussian_2d["bounds"], savedir=str(tmp_path),
                            verbose=False, random_state=4, cache=False)
        sm.init_samples(ntrain=40)
        sm.init_gp(kernel=kernel, hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 5})
        assert sm.kernel_name == kernel
        sm.active_train(niter=3, algorithm="bape", optimizer_kwargs={"ncand": 2048})
        sm.run_emcee(nwalkers=12, nsteps=300, min_ess=50)
        assert sm.emcee_samples.shape[1] == 2
    with pytest.raises(ValueError):
        sm2 = SurrogateModel(lnlike_fn=gaussian_2d["fn"], bounds=gaussian_2d["bounds"], savedir=str(tmp_path), verbose=False)
        sm2.init_samples(ntrain=10)
        sm2.init_gp(kernel="PeriodicKernel")
