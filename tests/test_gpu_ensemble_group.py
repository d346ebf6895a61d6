"""GPU parity of the group ensemble kernel (ens_group_kernel: training set partitioned over groups of workgroups,
kernel sums on the matrix cores) against oracle/stretch_oracle.py and against the launch-per-half-step path.

Reference call sites: alabi/core.py:2319-2325 (EnsembleSampler.run_mcmc), :2073-2100 (lnprob).  The group kernel sums in
another order than the other ensemble paths and forms -r^2/2 as one augmented dot product, so chains are compared to
rounding (<= 1e-7 absolute on coordinates, identical acceptance counts), not bit for bit."""
import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu


def _oracle_lnp(o, y, bounds):
    from oracle.stretch_oracle import box_lnprior_batch

    def f(q):
        lp = box_lnprior_batch(q, bounds)
        inside = np.isfinite(lp)
        out = np.full(len(q), -np.inf)
        if inside.any():
            out[inside] = o.predict(y, q[inside])
        return out
    return f


def _pair(N, d, seed, ell2=None, kernel="ExpSquaredKernel"):
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(N, d, seed, log_wn=-10.0, ell2=ell2)
    kw = {} if kernel == "ExpSquaredKernel" else {"kernel": kernel}
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], **kw); g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], **kw).compute(X)
    return g, o, y


@pytest.mark.parametrize("N,d,W,nsteps", [
    (300, 3, 32, 120),        # tiny: more members than point tiles allow, one proposal tile
    (777, 5, 130, 80),        # ragged: N not a multiple of 16, W/2 not a multiple of 16
    (2500, 10, 256, 60),      # does not fit ens_stream_kernel (N > 2048)
    (1500, 20, 96, 50),       # d = 20: 32 lanes per row, 6 MFMA k-steps
    (600, 30, 64, 40),        # largest supported dimension (d + 2 = 32)
    (900, 4, 33, 60),         # odd walker count: the two halves differ in size
])
def test_group_kernel_chain_vs_oracle(N, d, W, nsteps, monkeypatch):
    import torch
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    assert torch.cuda.is_available()
    monkeypatch.setenv("ALABI_ENS_GROUP", "1")
    g, o, y = _pair(N, d, 100 + d, ell2=2.0 * d)
    bounds = np.array([[-3.0, 3.0]] * d)
    p0 = np.random.RandomState(5).uniform(-2, 2, (W, d))
    s = EnsembleSampler(W, d, g, y, bounds, seed=777, live_dangerously=True)
    s.run_mcmc(p0, nsteps)
    assert s.last_path == "group", s.last_path
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(p0, nsteps, _oracle_lnp(o, y, bounds), seed=777)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7
    assert np.max(np.abs(s.get_log_prob() - lp_o) / (np.abs(lp_o) + 1)) <= 1e-8
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)
    assert 0.02 < s.acceptance_fraction.mean() < 0.98


def test_group_kernel_sixteen_member_blocking(monkeypatch):
    """Wide rows (d = 20) normally run with eight members per group and part of each wave's point tiles in registers;
    ALABI_ENS_GROUP_G16=1 selects the sixteen-member instantiation (what the planner falls back to when a slice does not fit):
    same chain to rounding, same acceptance counts."""
    from alabi_amd import EnsembleSampler
    g, o, y = _pair(1500, 20, 120, ell2=40.0)
    bounds = np.array([[-3.0, 3.0]] * 20)
    p0 = np.random.RandomState(5).uniform(-2, 2, (96, 20))
    monkeypatch.setenv("ALABI_ENS_GROUP", "1")
    runs = []
    for g16 in ("0", "1"):
        monkeypatch.setenv("ALABI_ENS_GROUP_G16", g16)
        s = EnsembleSampler(96, 20, g, y, bounds, seed=777)
        s.run_mcmc(p0, 60)
        assert s.last_path == "group"
        runs.append((s.get_chain(), s._naccept.cpu().numpy().copy()))
    assert np.max(np.abs(runs[0][0] - runs[1][0])) <= 1e-7
    assert np.array_equal(runs[0][1], runs[1][1])


def test_group_kernel_matches_half_step_path_and_continues(monkeypatch):
    """Two consecutive runs (the second continues the first), thinning, several ensembles per launch: the group kernel
    agrees with the launch-per-half-step kernels to rounding and the acceptance counters are identical."""
    from alabi_amd import EnsembleSampler
    g, o, y = _pair(1200, 6, 9, ell2=10.0)
    bounds = np.array([[-3.0, 3.0]] * 6)
    W, E = 48, 3
    p0 = np.random.RandomState(2).uniform(-2, 2, (W * E, 6))
    runs = {}
    for tag, env in (("half", {"ALABI_ENS_STREAM": "0"}), ("group", {"ALABI_ENS_STREAM": "1", "ALABI_ENS_GROUP": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = EnsembleSampler(W, 6, g, y, bounds, seed=31, n_ensembles=E)
        s.run_mcmc(p0, 90, thin_by=3)
        s.run_mcmc(None, 41)
        assert s.last_path == ("group" if tag == "group" else "launch-per-half-step")
        runs[tag] = (s.get_chain(), s.get_log_prob(), s._naccept.cpu().numpy().copy())
    assert runs["half"][0].shape == runs["group"][0].shape == (30 + 41, W * E, 6)
    assert np.max(np.abs(runs["half"][0] - runs["group"][0])) <= 1e-7
    assert np.max(np.abs(runs["half"][1] - runs["group"][1]) / (np.abs(runs["half"][1]) + 1)) <= 1e-8
    assert np.array_equal(runs["half"][2], runs["group"][2])


def test_group_kernel_across_chunks(monkeypatch):
    """More steps than one launch covers (1024): two full chunks and a remainder, the state carried from launch to launch
    through hist row 0; chain, log-probabilities and acceptance counters against the launch-per-half-step kernels."""
    from alabi_amd import EnsembleSampler
    g, o, y = _pair(700, 4, 19, ell2=7.0)
    bounds = np.array([[-3.0, 3.0]] * 4)
    W = 36
    p0 = np.random.RandomState(4).uniform(-2, 2, (W, 4))
    out = {}
    for tag, env in (("half", {"ALABI_ENS_STREAM": "0"}), ("group", {"ALABI_ENS_STREAM": "1", "ALABI_ENS_GROUP": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = EnsembleSampler(W, 4, g, y, bounds, seed=8)
        s.run_mcmc(p0, 2300, thin_by=5)
        out[tag] = (s.get_chain(), s.get_log_prob(), s._naccept.cpu().numpy().copy(), s.last_path)
    assert out["group"][3] == "group" and out["group"][0].shape == (460, W, 4)
    assert np.max(np.abs(out["half"][0] - out["group"][0])) <= 1e-7
    assert np.array_equal(out["half"][2], out["group"][2])


def test_group_kernel_normal_prior_and_affine_logp(monkeypatch):
    """lnprior_normal on two coordinates + an affine y scaler folded into amplitude and mean (alabi/utility.py:370,
    alabi/core.py:1483-1502), group kernel against the launch-per-half-step kernels."""
    from alabi_amd import EnsembleSampler
    g, o, y = _pair(1000, 5, 4, ell2=8.0)
    bounds = np.array([[-3.0, 3.0]] * 5)
    p0 = np.random.RandomState(8).uniform(-2, 2, (64, 5))
    prior = (np.array([0.3, np.nan, -0.2, np.nan, np.nan]), np.array([0.8, np.nan, 1.5, np.nan, np.nan]))
    out = {}
    for tag, env in (("half", {"ALABI_ENS_STREAM": "0"}), ("group", {"ALABI_ENS_STREAM": "1", "ALABI_ENS_GROUP": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = EnsembleSampler(64, 5, g, y, bounds, seed=12, logp_affine=(2.5, -0.75), normal_prior=prior)
        s.run_mcmc(p0, 100)
        out[tag] = (s.get_chain(), s.get_log_prob(), s._naccept.cpu().numpy().copy(), s.last_path)
    assert out["group"][3] == "group"
    assert np.max(np.abs(out["half"][0] - out["group"][0])) <= 1e-7
    assert np.max(np.abs(out["half"][1] - out["group"][1]) / (np.abs(out["half"][1]) + 1)) <= 1e-8
    assert np.array_equal(out["half"][2], out["group"][2])


@pytest.mark.parametrize("kernel", ["Matern32Kernel", "Matern52Kernel", "RationalQuadraticKernel"])
def test_group_kernel_other_kernel_families(kernel, monkeypatch):
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    monkeypatch.setenv("ALABI_ENS_GROUP", "1")
    g, o, y = _pair(500, 4, 21, ell2=6.0, kernel=kernel)
    bounds = np.array([[-3.0, 3.0]] * 4)
    p0 = np.random.RandomState(3).uniform(-2, 2, (40, 4))
    s = EnsembleSampler(40, 4, g, y, bounds, seed=5)
    s.run_mcmc(p0, 60)
    assert s.last_path == "group"
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(p0, 60, _oracle_lnp(o, y, bounds), seed=5)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)


def test_group_kernel_inputs_far_from_origin(monkeypatch):
    """Training inputs thousands of length scales from the origin: the augmented dot product is formed relative to the
    training mean, so the exponent keeps its digits (the same shift the matrix-core predict-mean kernel applies)."""
    from alabi_amd import EnsembleSampler, HipGP
    from oracle.gp_oracle import OracleGP
    from oracle import stretch_oracle as so
    monkeypatch.setenv("ALABI_ENS_GROUP", "1")
    X, y, h = make_problem(800, 4, 77, log_wn=-10.0, ell2=4.0)
    off = np.array([4000.0, -2500.0, 1000.0, 8000.0])
    Xo = X + off
    g = HipGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(Xo)
    o = OracleGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(Xo)
    bounds = np.stack([off - 3.0, off + 3.0], axis=1)
    p0 = off + np.random.RandomState(6).uniform(-2, 2, (32, 4))
    s = EnsembleSampler(32, 4, g, y, bounds, seed=2)
    s.run_mcmc(p0, 80)
    assert s.last_path == "group"
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(p0, 80, _oracle_lnp(o, y, bounds), seed=2)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7 * 8000
    assert np.max(np.abs(s.get_log_prob() - lp_o) / (np.abs(lp_o) + 1)) <= 1e-8
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)


def test_group_kernel_timeout_falls_back(monkeypatch):
    """A hand-off that cannot arrive within the spin limit: every workgroup leaves, the sampler restores the state and
    repeats the run on the launch-per-half-step path."""
    from alabi_amd import EnsembleSampler
    g, o, y = _pair(4000, 3, 11, ell2=5.0)     # enough kernel-sum time that the partials are never there at the second look
    bounds = np.array([[-3.0, 3.0]] * 3)
    p0 = np.random.RandomState(1).uniform(-2, 2, (16, 3))
    monkeypatch.setenv("ALABI_ENS_STREAM", "0")
    ref = EnsembleSampler(16, 3, g, y, bounds, seed=4); ref.run_mcmc(p0, 60)
    monkeypatch.setenv("ALABI_ENS_STREAM", "1")
    monkeypatch.setenv("ALABI_ENS_GROUP", "1")
    monkeypatch.setenv("ALABI_ENS_SPIN_LIMIT", "1")
    s = EnsembleSampler(16, 3, g, y, bounds, seed=4); s.run_mcmc(p0, 60)
    assert getattr(s, "stream_fallbacks", 0) == 1 and s.last_path == "launch-per-half-step"
    np.testing.assert_array_equal(s.get_chain(), ref.get_chain())
    np.testing.assert_array_equal(s.acceptance_fraction, ref.acceptance_fraction)


def test_C4_runs_on_the_group_kernel():
    """The default path of BASELINE's C4 (N = 5000, 1024 walkers) is the group kernel; 50 steps against the oracle."""
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.workloads import make_config
    from oracle.gp_oracle import OracleGP, sqexp_kernel
    from oracle import stretch_oracle as so
    cfg = make_config("C4")
    h, d, b = cfg["hyper"], cfg["d"], cfg["bounds"]
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(cfg["X"])
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"])
    alpha = o._compute_alpha(cfg["y"])

    def lnp(q):
        inside = np.all((q > b[:, 0]) & (q < b[:, 1]), axis=1)
        out = np.full(len(q), -np.inf)
        if inside.any():
            out[inside] = sqexp_kernel(q[inside], cfg["X"], h["log_amp"], h["log_M"]) @ alpha + h["mean"]
        return out
    s = EnsembleSampler(cfg["W"], d, g, cfg["y"], b, seed=99)
    s.run_mcmc(cfg["p0"], 50)
    assert s.last_path == "group"
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(cfg["p0"], 50, lnp, seed=99)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7
    assert np.max(np.abs(s.get_log_prob() - lp_o) / (np.abs(lp_o) + 1)) <= 1e-8
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)


def _config_lnp(cfg, o):
    from oracle.gp_oracle import sqexp_kernel
    h, b = cfg["hyper"], cfg["bounds"]
    alpha = o._compute_alpha(cfg["y"])

    def lnp(q):
        inside = np.all((q > b[:, 0]) & (q < b[:, 1]), axis=1)
        out = np.full(len(q), -np.inf)
        if inside.any():
            out[inside] = sqexp_kernel(q[inside], cfg["X"], h["log_amp"], h["log_M"]) @ alpha + h["mean"]
        return out
    return lnp


def test_C5_sized_group_kernel_vs_oracle():
    """The instantiation BASELINE's C5-sized ensemble selects (N = 10000, d = 20, 2048 walkers): wide rows (six MFMA k-steps),
    five point tiles per wave in registers AND five staged in LDS -- the mixed register + LDS loop of the kernel sums -- compared
    with the oracle step for step; the blocking is asserted so a planner change cannot move this test to another kernel."""
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.workloads import make_config
    from oracle.gp_oracle import OracleGP
    from oracle import stretch_oracle as so
    cfg = make_config("C5")
    h, d, b = cfg["hyper"], cfg["d"], cfg["bounds"]
    assert (cfg["N"], d, cfg["W"]) == (10000, 20, 2048)
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(cfg["X"])
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"])
    nsteps = 24
    s = EnsembleSampler(cfg["W"], d, g, cfg["y"], b, seed=4242)
    s.run_mcmc(cfg["p0"], nsteps)
    assert s.last_path == "group"
    pl = s.group_plan
    assert (pl["KS"], pl["G"], pl["Q"], pl["RT"]) == (6, 8, 2, 5) and pl["ltw"] == 5 and pl["NG"] == 32, pl
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(cfg["p0"], nsteps, _config_lnp(cfg, o), seed=4242)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7
    assert np.max(np.abs(s.get_log_prob() - lp_o) / (np.abs(lp_o) + 1)) <= 1e-8
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)
    assert 0.02 < s.acceptance_fraction.mean() < 0.98


@pytest.mark.parametrize("N,d,W,want", [
    (8000, 10, 512, dict(KS=3, G=8, Q=1, RT=5, ltw=3)),     # narrow rows, 8 point tiles per wave: 5 in registers + 3 in LDS
    (10000, 14, 256, dict(KS=4, G=8, Q=1, RT=5, ltw=5)),    # widest narrow row (16 words), 10 tiles per wave
])
def test_group_kernel_register_plus_lds_tiles_narrow_rows(N, d, W, want):
    """Narrow rows with more point tiles per wave than the five the registers hold (N > 5120): both parts of the kernel-sum
    loop run; chain against the oracle, blocking asserted."""
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    g, o, y = _pair(N, d, 300 + d, ell2=3.0 * d)
    bounds = np.array([[-3.0, 3.0]] * d)
    p0 = np.random.RandomState(15).uniform(-2, 2, (W, d))
    s = EnsembleSampler(W, d, g, y, bounds, seed=606)
    s.run_mcmc(p0, 30)
    assert s.last_path == "group"
    pl = s.group_plan
    assert {k: pl[k] for k in want} == want, pl
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(p0, 30, _oracle_lnp(o, y, bounds), seed=606)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7
    assert np.max(np.abs(s.get_log_prob() - lp_o) / (np.abs(lp_o) + 1)) <= 1e-8
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)


def test_C4_group_kernel_posterior_ks_vs_launch_per_half_step_path(monkeypatch):
    """Statistical agreement at C4 (N = 5000, 1024 walkers): two-sample KS distance per marginal between a long group-kernel
    chain and an INDEPENDENT (other seed) chain of the launch-per-half-step kernels, which equal the oracle step for step:
    < 0.01 on every marginal with n_eff > 5e4 on both sides."""
    from scipy.stats import ks_2samp
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.workloads import make_config
    cfg = make_config("C4")
    h, d, W = cfg["hyper"], cfg["d"], cfg["W"]
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(cfg["X"])
    flat, neff = {}, {}
    nsteps, thin, burn = 12000, 20, 2000
    for tag, stream, seed in (("group", "1", 11), ("half", "0", 12)):
        monkeypatch.setenv("ALABI_ENS_STREAM", stream)
        s = EnsembleSampler(W, d, g, cfg["y"], cfg["bounds"], seed=seed)
        s.run_mcmc(cfg["p0"], nsteps, thin_by=thin)
        assert s.last_path == ("group" if tag == "group" else "launch-per-half-step")
        tau = s.get_autocorr_time(discard=burn // thin, tol=0) * thin
        flat[tag] = s.get_chain(discard=burn // thin, flat=True)
        neff[tag] = flat[tag].shape[0] * thin / np.max(tau)
    ks = np.array([ks_2samp(flat["group"][:, k], flat["half"][:, k]).statistic for k in range(d)])
    print(f"C4 KS per marginal {ks}, n_eff {neff}")
    assert min(neff.values()) > 5e4
    assert np.all(ks < 0.01), ks
