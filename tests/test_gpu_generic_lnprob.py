"""run_emcee as a drop-in beyond the fused defaults: arbitrary ``prior_fn`` / ``like_fn`` callables (the reference's
lnprob = like_fn(theta) + prior_fn(theta), alabi/core.py:2073-2100, :2253-2280), the reference's two non-affine
y scalers (alabi/utility.py:62-71), ``find_map`` / ``opt_init`` (core.py:2103, :2290-2294), and emcee's
reset() semantics.  Chains are compared step for step with oracle.stretch_oracle.run_ensemble driven by the same callable."""
from functools import partial

import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import torch
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    assert torch.cuda.is_available()
    X, y, h = make_problem(400, 4, 13)
    g = HipGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    return g, o, X, y, np.array([[-3.0, 3.0]] * 4)


def test_custom_prior_callable_matches_oracle(setup):
    """prior_fn = any Python callable: surrogate part on the device (no box gate), prior on the host per proposal."""
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    g, o, X, y, bounds = setup
    W, d, nsteps = 24, 4, 150
    calls = []

    def log_prior(q):                                   # batch of points in the sampler's coordinates -> [n]
        calls.append(len(q))
        return -0.5 * np.sum((q / 1.5) ** 2, axis=1)

    p0 = np.random.RandomState(3).uniform(-2, 2, (W, d))
    s = EnsembleSampler(W, d, g, y, bounds, seed=5, prior_fn=log_prior, gate_box=False)
    s.run_mcmc(p0, nsteps, thin_by=2)
    assert s.last_path == "host-callback" and len(calls) == 1 + 2 * nsteps
    lnp = lambda q: o.predict(y, q) + -0.5 * np.sum((q / 1.5) ** 2, axis=1)  # noqa: E731
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(p0, nsteps, lnp, seed=5, thin_by=2)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7
    assert np.max(np.abs(s.get_log_prob() - lp_o)) <= 1e-7
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)
    # walkers may leave the box: the Gaussian prior alone confines them
    with pytest.raises(ValueError):                      # emcee: "Probability function returned NaN"
        bad = EnsembleSampler(W, d, g, y, bounds, seed=5, prior_fn=lambda q: np.full(len(q), np.nan), gate_box=False)
        bad.run_mcmc(p0, 2)


def test_host_likelihood_under_box_prior_matches_oracle(setup):
    """like_fn on the host (e.g. the true likelihood) with the default box prior: only in-box proposals are evaluated."""
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    g, o, X, y, bounds = setup
    W, d, nsteps = 16, 4, 120
    seen = []

    def like(q):
        seen.append(q.copy())
        return -0.5 * np.sum(q ** 2, axis=1) - 0.1 * np.sum(q ** 4, axis=1)

    p0 = np.random.RandomState(4).uniform(-2.9, 2.9, (W, d))
    s = EnsembleSampler(W, d, g, y, bounds, seed=9, like_fn=like)
    s.run_mcmc(p0, nsteps)
    allq = np.vstack(seen)
    assert np.all(allq > bounds[:, 0]) and np.all(allq < bounds[:, 1])

    def lnp(q):
        out = so.box_lnprior_batch(q, bounds)
        ok = np.isfinite(out)
        out[ok] = -0.5 * np.sum(q[ok] ** 2, axis=1) - 0.1 * np.sum(q[ok] ** 4, axis=1)
        return out
    chain_o, _, nacc_o, _, _ = so.run_ensemble(p0, nsteps, lnp, seed=9)
    assert np.array_equal(s.get_chain(), chain_o)        # same NumPy arithmetic on both sides: identical bits
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)


@pytest.mark.parametrize("kind", ["nlog", "log"])
def test_nonaffine_y_scaler_fused(setup, kind):
    """logp = -10^mu (nlog_scaler) / 10^mu (log_scaler) of the GP mean, evaluated inside the half-step kernels."""
    from alabi_amd import EnsembleSampler, HipGP
    from oracle.gp_oracle import OracleGP
    from oracle import stretch_oracle as so
    _, _, X, y, bounds = setup
    ys = np.log10(-(y - 1.0)) if kind == "nlog" else np.log10(y - y.min() + 1.0)       # what the scaler's transform gives
    h = dict(mean=float(np.median(ys)), wn=-10.0, amp=float(np.log(np.var(ys))), log_M=np.log(np.full(4, 3.0)))
    g = HipGP(4, h["mean"], h["wn"], h["amp"], h["log_M"]); g.compute(X)
    o = OracleGP(4, h["mean"], h["wn"], h["amp"], h["log_M"]).compute(X)
    W, nsteps = 20, 200
    p0 = np.random.RandomState(6).uniform(-2, 2, (W, 4))
    s = EnsembleSampler(W, 4, g, ys, bounds, seed=21, logp_map=kind)
    s.run_mcmc(p0, nsteps)
    assert s.last_path == "launch-per-half-step"
    sign = -1.0 if kind == "nlog" else 1.0

    def lnp(q):
        out = so.box_lnprior_batch(q, bounds)
        ok = np.isfinite(out)
        out[ok] = sign * 10.0 ** o.predict(ys, q[ok])
        return out
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(p0, nsteps, lnp, seed=21)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7
    assert np.max(np.abs(s.get_log_prob() - lp_o) / (np.abs(lp_o) + 1)) <= 1e-9
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)


def test_reset_does_not_rewind_the_draws(setup):
    """burn-in, reset(), production: the production run continues the draw counter (emcee's reset keeps the RNG state)."""
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    g, o, X, y, bounds = setup
    W, d = 16, 4
    p0 = np.random.RandomState(8).uniform(-2, 2, (W, d))
    s = EnsembleSampler(W, d, g, y, bounds, seed=31)
    s.run_mcmc(p0, 40)
    last = s.get_chain()[-1].copy()
    s.reset()
    assert s.iteration == 0 and int(s._naccept.sum()) == 0
    s.run_mcmc(None, 30)
    lnp = lambda q: np.where(np.isfinite(so.box_lnprior_batch(q, bounds)), o.predict(y, q), -np.inf)  # noqa: E731
    cont, _, nacc, _, _ = so.run_ensemble(last, 30, lnp, seed=31, step0=40)
    replay, _, _, _, _ = so.run_ensemble(last, 30, lnp, seed=31, step0=0)
    assert np.max(np.abs(s.get_chain() - cont)) <= 1e-7
    assert np.max(np.abs(s.get_chain() - replay)) > 1e-3          # NOT the burn-in's draws again
    assert np.array_equal(s._naccept.cpu().numpy(), nacc)
    assert np.allclose(s.acceptance_fraction, nacc / 30.0)


def test_run_emcee_docstring_example_custom_prior_and_opt_init(tmp_path):
    """The reference's own example: sm.run_emcee(prior_fn=log_prior, opt_init=True) (core.py:2236-2239)."""
    from alabi_amd import SurrogateModel
    from alabi_amd.benchmarks import gaussian_2d
    sm = SurrogateModel(lnlike_fn=gaussian_2d["fn"], bounds=gaussian_2d["bounds"], savedir=str(tmp_path), verbose=False,
                        random_state=4, cache=False)
    sm.init_samples(ntrain=120)
    sm.init_gp(hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 10})

    def log_prior(theta):
        return -0.5 * np.sum(((theta - 0.3) / 0.2) ** 2)

    sm.run_emcee(prior_fn=log_prior, opt_init=True, nwalkers=12, nsteps=400, min_ess=50)
    assert sm.emcee_sampler.last_path == "host-callback"
    assert np.all(np.isfinite(sm.map_theta)) and np.isfinite(sm.map_lnprob)
    # the MAP beats every training point's posterior value
    post = np.array([float(sm.surrogate_log_likelihood(t)) + log_prior(t) for t in sm.theta()])
    assert sm.map_lnprob >= post.max() - 1e-6
    last = sm.emcee_samples_full[-1]
    lp = sm.emcee_sampler.get_log_prob()[-1]
    ref = np.array([float(sm.surrogate_log_likelihood(t)) + log_prior(t) for t in last])
    assert np.max(np.abs(lp - ref)) <= 1e-7 * (np.max(np.abs(ref)) + 1)
    assert sm.emcee_samples.shape[1] == 2 and sm.emcee_run
    # posterior = N(0.5, 0.1 I) x N(0.3, 0.04 I): mean (0.5/0.1 + 0.3/0.04) / (1/0.1 + 1/0.04) = 0.357
    assert np.all(np.abs(sm.emcee_samples.mean(axis=0) - 0.357) < 0.06)


def test_run_emcee_true_likelihood_and_nlog_scaler(tmp_path):
    from alabi_amd import SurrogateModel, utility as ut
    from alabi_amd.benchmarks import gaussian_2d
    fn = lambda th: float(gaussian_2d["fn"](th)) - 5.0          # strictly negative: nlog_scaler applies  # noqa: E731
    sm = SurrogateModel(lnlike_fn=fn, bounds=gaussian_2d["bounds"], savedir=str(tmp_path), verbose=False, random_state=6,
                        cache=False)
    sm.init_samples(ntrain=150)
    # like_fn="true" needs no GP at all (core.py:2073-2100 only requires one for the surrogate)
    sm.run_emcee(like_fn="true", nwalkers=10, nsteps=150, min_ess=20)
    assert sm.like_fn_name == "true" and sm.emcee_samples_true.shape[1] == 2
    lp = sm.emcee_sampler.get_log_prob()[-1]
    ref = np.array([fn(t) for t in sm.emcee_samples_full[-1]])
    assert np.array_equal(lp, ref)
    # the shipped non-affine scaler: fused through the device-side map
    sm.init_gp(hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 10}, y_scaler=ut.nlog_scaler)
    assert np.allclose(sm._y, np.log10(-sm.y_train.flatten()))
    sm.run_emcee(nwalkers=12, nsteps=300, min_ess=50)
    assert sm.emcee_sampler.last_path == "launch-per-half-step" and sm.emcee_sampler.logp_map == "nlog"
    last = sm.emcee_samples_full[-1]
    lp = sm.emcee_sampler.get_log_prob()[-1]
    ref = np.array([float(sm.surrogate_log_likelihood(t)) for t in last])
    assert np.max(np.abs(lp - ref)) <= 1e-9 * (np.max(np.abs(ref)) + 1)


@pytest.mark.parametrize("method", ["ml", "cv"])
def test_init_gp_without_amplitude(tmp_path, method):
    """fit_amp=False: the reference's model has no constant factor and no log_constant parameter (gp_utils.py:230,
    core.py:1057-1059); both hyper-parameter searches must run over the remaining vector."""
    from alabi_amd import SurrogateModel
    from alabi_amd.benchmarks import gaussian_2d
    sm = SurrogateModel(lnlike_fn=gaussian_2d["fn"], bounds=gaussian_2d["bounds"], savedir=str(tmp_path), verbose=False,
                        random_state=2, cache=False)
    sm.init_samples(ntrain=50)
    kw = dict(gp_nopt=2) if method == "ml" else dict(cv_n_candidates=8, cv_stage2_candidates=4, cv_stage3_candidates=3)
    sm.init_gp(fit_amp=False, hyperopt_method=method, **kw)
    names = sm.gp.get_parameter_names()
    assert not any("log_constant" in n for n in names) and len(names) == 2 + 2
    assert names[-1] == "kernel:metric:log_M_1_1" and sm.gp.log_constant == 0.0
    assert sm.hp_bounds.shape == (4, 2) and np.all(np.isfinite(sm.hp_bounds.astype(float)))
    assert sm.gp.grad_log_likelihood(sm._y).shape == (4,)
    p = sm.gp.get_parameter_vector()
    assert np.all(p >= sm.hp_bounds[:, 0].astype(float) - 1e-9) and np.all(p <= sm.hp_bounds[:, 1].astype(float) + 1e-9)
    sm.active_train(niter=2, algorithm="bape", gp_opt_freq=1000, optimizer_kwargs={"ncand": 2048})
    assert sm.ntrain == 52 and sm.gp.log_constant == 0.0
