"""Posterior agreement: two-sample Kolmogorov-Smirnov distance per marginal between the GPU ensemble chain and
an independent CPU oracle chain on the same surrogate (north star: KS < 0.01)."""
import numpy as np
import pytest
from scipy.stats import ks_2samp

from conftest import make_problem

pytestmark = pytest.mark.gpu


def test_ks_distance_gpu_vs_cpu_chain():
    import torch
    from alabi_amd import EnsembleSampler, HipGP
    from oracle.gp_oracle import OracleGP, sqexp_kernel
    from oracle import stretch_oracle as so
    assert torch.cuda.is_available()
    d, N, W, nsteps, thin, burn = 5, 500, 64, 120_000, 10, 3000
    X, y, h = make_problem(N, d, 31)
    bounds = np.array([[-3.0, 3.0]] * d)
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    alpha = o._compute_alpha(y)
    p0 = np.random.RandomState(1).uniform(-1, 1, (W, d))

    s = EnsembleSampler(W, d, g, y, bounds, seed=101)
    s.run_mcmc(p0, nsteps, thin_by=thin)
    gpu = s.get_chain(discard=burn // thin, flat=True)

    def lnp(q):                                   # vectorised oracle log-probability (same arithmetic as OracleGP.predict)
        inside = np.all((q > bounds[:, 0]) & (q < bounds[:, 1]), axis=1)
        out = np.full(len(q), -np.inf)
        if inside.any():
            out[inside] = sqexp_kernel(q[inside], X, h["log_amp"], h["log_M"]) @ alpha + h["mean"]
        return out

    chain, _, _, _, _ = so.run_ensemble(p0, nsteps, lnp, seed=202, thin_by=thin)   # different seed: independent chain
    cpu = chain[burn // thin:].reshape(-1, d)
    tau = s.get_autocorr_time(discard=burn // thin, tol=0) * thin
    n_eff = gpu.shape[0] * thin / np.max(tau)
    ks = np.array([ks_2samp(gpu[:, k], cpu[:, k]).statistic for k in range(d)])
    print(f"KS per marginal {ks}, tau {tau}, n_eff {n_eff:.3g}")
    assert n_eff > 5e4
    assert np.all(ks < 0.01), ks
    # first two moments agree as well
    assert np.all(np.abs(gpu.mean(0) - cpu.mean(0)) < 0.02 * (bounds[:, 1] - bounds[:, 0]))
    assert np.all(np.abs(gpu.std(0) / cpu.std(0) - 1) < 0.03)
