"""BASELINE.json's five configurations on their own workloads (alabi_amd.workloads.make_config), each against the
CPU oracle: GP predict (mean + variance) AND an ensemble chain at the configuration's own walker count, step for step.

  C1  2-D Rosenbrock, N=50, W=64          C2  5-D Gaussian shells, N=500, W=128
  C3  10-D Gaussian, N=2000, W=256 (headline; also the two-sample KS test and the likelihood gradient at this size)
  C4  10-D Gaussian, N=5000, W=1024 (ensemble slice: 50 steps)
  C5  20-D ARD, N=10000: BAPE scan over 10^6 candidates (oracle at the arg-min and 200 random candidates) and three
      active_train iterations at N=10000 (the appended factor against an oracle factorised from scratch).

Reference call sites: alabi/core.py:2319-2325 (ensemble), :85 / :1601 (predict), :1587-1667 (next point), :1780 (refit).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(cfg):
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    h, d = cfg["hyper"], cfg["d"]
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(cfg["X"])
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"])
    return g, o


def _lnp_vectorised(cfg, o):
    """Oracle log-probability of a batch: box prior + K* alpha + m (the arithmetic of OracleGP.predict)."""
    from oracle.gp_oracle import sqexp_kernel
    h, X, b = cfg["hyper"], cfg["X"], cfg["bounds"]
    alpha = o._compute_alpha(cfg["y"])

    def lnp(q):
        inside = np.all((q > b[:, 0]) & (q < b[:, 1]), axis=1)
        out = np.full(len(q), -np.inf)
        if inside.any():
            out[inside] = sqexp_kernel(q[inside], X, h["log_amp"], h["log_M"]) @ alpha + h["mean"]
        return out
    return lnp


@pytest.mark.parametrize("name,nsteps", [("C1", 300), ("C2", 250), ("C3", 200)])
def test_config_predict_and_chain_vs_oracle(name, nsteps):
    import torch
    from alabi_amd import EnsembleSampler
    from alabi_amd.workloads import make_config
    from oracle import stretch_oracle as so
    assert torch.cuda.is_available()
    cfg = make_config(name)
    d, W, b = cfg["d"], cfg["W"], cfg["bounds"]
    g, o = _pair(cfg)
    amp = np.exp(cfg["hyper"]["log_amp"])
    Xs = np.random.RandomState(17).uniform(b[:, 0], b[:, 1], (1536, d))
    mu, var = g.predict(cfg["y"], Xs, return_var=True)
    mu_o, var_o = o.predict(cfg["y"], Xs, return_var=True)
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8, name
    assert np.max(np.abs(var - var_o)) <= 1e-6 * amp, name
    mu_only = g.predict(cfg["y"], Xs, return_cov=False)
    assert np.max(np.abs(mu_only - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    # the ensemble at the configuration's own walker count, counter-based draws, chain compared step for step
    s = EnsembleSampler(W, d, g, cfg["y"], b, seed=4242)
    s.run_mcmc(cfg["p0"], nsteps)
    chain = s.get_chain()
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(cfg["p0"], nsteps, _lnp_vectorised(cfg, o), seed=4242)
    assert chain.shape == chain_o.shape == (nsteps, W, d)
    assert np.max(np.abs(chain - chain_o)) <= 1e-7, (name, s.last_path)
    lp = s.get_log_prob()
    assert np.max(np.abs(lp - lp_o) / (np.abs(lp_o) + 1)) <= 1e-8
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)               # accept decisions: identical
    assert 0.05 < s.acceptance_fraction.mean() < 0.95


def test_C4_ensemble_slice_vs_oracle():
    """W=1024 walkers at N=5000 (the multi-proposal half-step kernel): 50 steps against the oracle."""
    from alabi_amd import EnsembleSampler
    from alabi_amd.workloads import make_config
    from oracle import stretch_oracle as so
    cfg = make_config("C4")
    g, o = _pair(cfg)
    s = EnsembleSampler(cfg["W"], cfg["d"], g, cfg["y"], cfg["bounds"], seed=99)
    s.run_mcmc(cfg["p0"], 50)
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(cfg["p0"], 50, _lnp_vectorised(cfg, o), seed=99)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7
    assert np.max(np.abs(s.get_log_prob() - lp_o) / (np.abs(lp_o) + 1)) <= 1e-8
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)


@pytest.fixture(scope="module")
def c5():
    from alabi_amd.workloads import make_config
    cfg = make_config("C5")
    g, o = _pair(cfg)          # CPU Cholesky of 10^4 x 10^4: a few seconds
    return cfg, g, o


def test_C5_scan_one_million_candidates(c5):
    """BAPE over M = 10^6 candidates at N=10000, d=20: the arg-min the device reports is the arg-min of the values it
    reports; the oracle's utility at that point and at 200 random candidates agrees with the device's."""
    import torch
    from alabi_amd.utility import utility_scan
    from oracle.utility_oracle import utility_batch
    cfg, g, o = c5
    b, d = cfg["bounds"], cfg["d"]
    gen = torch.Generator(device="cuda"); gen.manual_seed(6)
    lo = torch.as_tensor(b[:, 0], device="cuda"); hi = torch.as_tensor(b[:, 1], device="cuda")
    M = 1_000_000
    cand = lo + (hi - lo) * torch.rand((M, d), dtype=torch.float64, device="cuda", generator=gen)
    best, val, idx, u, mu, var = utility_scan(g, cfg["y"], cand, b, "bape", return_all=True)
    u_h = u.cpu().numpy()
    fin = np.isfinite(u_h)
    assert fin.sum() > 0.99 * M
    assert idx == int(np.flatnonzero(fin)[np.argmin(u_h[fin])]) and val == u_h[idx]      # index arithmetic: exact
    pick = np.concatenate([[idx], np.random.RandomState(3).choice(M, 200, replace=False)])
    sub = cand[torch.as_tensor(pick, device="cuda")].cpu().numpy()
    np.testing.assert_array_equal(best, sub[0])
    mu_o, var_o = o.predict(cfg["y"], sub, return_var=True)
    amp = np.exp(cfg["hyper"]["log_amp"])
    mu_d, var_d = mu.cpu().numpy()[pick], var.cpu().numpy()[pick]
    assert np.max(np.abs(mu_d - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert np.max(np.abs(var_d - var_o)) <= 1e-6 * amp
    # the epilogue on the device's own (mu, var) is the reference formula to rounding
    u_chk = utility_batch("bape", mu_d, var_d, sub, b)
    ok = np.isfinite(u_chk)
    assert np.all(np.abs(u_h[pick][ok] - u_chk[ok]) <= 1e-12 * np.abs(u_chk[ok]))
    # and against the oracle's (mu, var): d u = 2 d mu + (1 + 1/(e^var - 1)) d var, var is O(amp) away from the data
    u_o = utility_batch("bape", mu_o, var_o, sub, b)
    both = ok & np.isfinite(u_o) & (var_o > 1e-3 * amp)
    assert both.sum() > 150
    assert np.max(np.abs(u_h[pick][both] - u_o[both]) / (np.abs(u_o[both]) + 1)) <= 1e-6
    assert val <= np.min(u_h[pick][ok])


def test_C5_active_train_three_iterations(c5, tmp_path):
    """active_train on the N=10000, d=20 workload, 10^6 scan candidates per iteration: three points are added by
    appending to the factor; the resulting GP predicts like the oracle factorised from scratch on the 10003 points."""
    from alabi_amd import SurrogateModel
    from oracle.gp_oracle import OracleGP
    from oracle.utility_oracle import utility_batch
    cfg, g, o = c5
    d, N = cfg["d"], cfg["N"]
    f = tmp_path / "c5_train.npz"
    np.savez(f, theta=cfg["X"], y=cfg["y"].reshape(-1, 1))
    sm = SurrogateModel(lnlike_fn=cfg["fn"], bounds=cfg["bounds"], savedir=str(tmp_path), verbose=False, random_state=5,
                        cache=False)
    sm.init_samples(train_file=str(f))
    assert sm.ntrain == N
    sm.init_gp(hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 1})
    h = cfg["hyper"]
    sm.gp.set_parameter_vector(np.concatenate([[h["mean"], h["log_white_noise"], h["log_amp"]], h["log_M"]]))
    assert sm.gp.compute(sm._theta, quiet=True)
    sm.active_train(niter=3, algorithm="bape", gp_opt_freq=1000,
                    optimizer_kwargs={"ncand": 1_000_000, "refine": 1, "nrefine": 4096, "polish": 5})
    assert sm.ntrain == N + 3 and sm._theta.shape == (N + 3, d)
    assert getattr(sm.gp, "appended", 0) >= 2
    assert np.array_equal(sm._theta[:N], cfg["X"])
    p = sm.gp.get_parameter_vector()
    oo = OracleGP(d, p[0], p[1], p[2], p[3:]).compute(sm._theta)
    Xs = np.random.RandomState(8).uniform(-3, 3, (96, d))
    Xs[:3] = sm._theta[N:] + 1e-3                      # next to the new points, where they matter
    mu, var = sm.gp.predict(sm._y, Xs, return_var=True)
    mu_o, var_o = oo.predict(sm._y, Xs, return_var=True)
    amp = np.exp(p[2])
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert np.max(np.abs(var - var_o)) <= 1e-6 * amp
    # the new points were evaluated with the true function and lie inside the box
    assert np.allclose(sm._y[N:], [cfg["fn"](t) for t in sm._theta[N:]], rtol=1e-12)
    assert np.all(sm._theta[N:] > cfg["bounds"][:, 0]) and np.all(sm._theta[N:] < cfg["bounds"][:, 1])
    # the last chosen point minimised the acquisition function of the GP before it was added: the oracle's BAPE value
    # there (10002 points) is no larger than at 200 random candidates
    o2 = OracleGP(d, p[0], p[1], p[2], p[3:]).compute(sm._theta[:-1])
    probe = np.vstack([sm._theta[-1:], np.random.RandomState(9).uniform(-3, 3, (200, d))])
    m2, v2 = o2.predict(sm._y[:-1], probe, return_var=True)
    u2 = utility_batch("bape", m2, v2, probe, cfg["bounds"])
    assert np.isfinite(u2[0]) and u2[0] <= np.min(u2[1:][np.isfinite(u2[1:])])


def test_C3_grad_log_likelihood_vs_oracle():
    """d logL / dp at the headline size (N=2000, d=10, nugget e^-12) against the oracle's analytic gradient.  The
    gradient is a difference of two O(cond) terms (alpha alpha^T - K^-1): the agreement is relative to the largest entry."""
    from alabi_amd.workloads import make_config
    cfg = make_config("C3")
    g, o = _pair(cfg)
    ga, go = g.grad_log_likelihood(cfg["y"]), o.grad_log_likelihood(cfg["y"])
    assert ga.shape == go.shape == (13,)
    scale = np.max(np.abs(go))
    print("grad logL  device", ga, "\n           oracle", go, "\n  max |diff| / max|g|", np.max(np.abs(ga - go)) / scale)
    assert np.max(np.abs(ga - go)) <= 1e-6 * scale
    ll, llo = g.log_likelihood(cfg["y"]), o.log_likelihood(cfg["y"])
    assert abs(ll - llo) <= 1e-9 * abs(llo)


def test_C3_ks_distance_gpu_vs_cpu_chain():
    """North star: two-sample KS distance per marginal < 0.01 between the GPU chain and an independent CPU oracle
    chain at C3 (N=2000, d=10, W=256).  >= 10^6 kept samples per side (CPU: 6.1e6 after burn-in, GPU: 5e7)."""
    from scipy.stats import ks_2samp
    from alabi_amd import EnsembleSampler
    from alabi_amd.workloads import make_config
    from oracle import stretch_oracle as so
    cfg = make_config("C3")
    d, W, b = cfg["d"], cfg["W"], cfg["bounds"]
    g, o = _pair(cfg)
    burn = 2000
    s = EnsembleSampler(W, d, g, cfg["y"], b, seed=101)
    s.run_mcmc(cfg["p0"], burn, store=False)
    s.run_mcmc(None, 200_000, thin_by=4)
    gpu = s.get_chain(flat=True)                                # 5.0e7 x 10 doubles would be 4 GB on the host: thinned by 4
    tau = s.get_autocorr_time(tol=0) * 4
    n_cpu = 26_000
    chain, _, _, _, _ = so.run_ensemble(cfg["p0"], n_cpu, _lnp_vectorised(cfg, o), seed=202)   # independent draws
    cpu = chain[burn:].reshape(-1, d)
    assert cpu.shape[0] >= 1_000_000 and gpu.shape[0] >= 1_000_000
    n_eff = cpu.shape[0] / np.max(tau)
    ks = np.array([ks_2samp(gpu[::4, k], cpu[:, k]).statistic for k in range(d)])
    print(f"C3 KS per marginal {np.round(ks, 4)}, tau {np.round(tau, 1)}, CPU n_eff {n_eff:.3g}")
    assert np.all(ks < 0.01), ks
    assert np.all(np.abs(gpu.mean(0) - cpu.mean(0)) < 0.02 * (b[:, 1] - b[:, 0]))
    assert np.all(np.abs(gpu.std(0) / cpu.std(0) - 1) < 0.03)
