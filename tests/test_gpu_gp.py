"""GPU parity: the HIP GP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64, stated per quantity):
  K, kernel.get_value : 1e-13 relative (elementwise, exp + d FMAs)
  L (factor)          : |L L^T - K| <= 1e-12 |K|   (L itself is not backward stable on ill-conditioned K)
  alpha               : residual |K alpha - r| <= 1e-8 |r|     (cond(K) up to ~1e9 at nugget e^-12)
  mu*                 : |dmu| / (|mu| + 1) <= 1e-8
  var*                : |dvar| <= 1e-6 amp  (two algebraically equal formulas, error ~ cond(K) eps amp)
  logdet / logL       : 1e-9 relative
"""
import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch


def _pair(X, y, h):
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    d = X.shape[1]
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    return g, o


@pytest.mark.parametrize("N,d,seed", [(50, 2, 0), (64, 1, 1), (65, 3, 2), (500, 5, 3), (777, 10, 4), (300, 20, 5), (130, 7, 6)])
def test_factor_alpha_predict(torch_gpu, N, d, seed):
    X, y, h = make_problem(N, d, seed)
    g, o = _pair(X, y, h)
    amp = np.exp(h["log_amp"])
    K = o.get_matrix(X)
    L = g.solver.get_factor().cpu().numpy()
    assert np.all(np.triu(L, 1) == 0)
    assert np.max(np.abs(L @ L.T - K)) <= 1e-12 * np.max(np.abs(K))
    Xs = np.random.RandomState(seed + 100).uniform(-3.2, 3.2, (333, d))
    mu, var = g.predict(y, Xs, return_var=True)
    mu_o, var_o = o.predict(y, Xs, return_var=True)
    alpha = g._alpha
    r = y - h["mean"]
    assert np.max(np.abs(K @ alpha - r)) <= 1e-8 * np.max(np.abs(r))
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert np.max(np.abs(var - var_o)) <= 1e-6 * amp
    mu_only = g.predict(y, Xs, return_cov=False)
    assert np.max(np.abs(mu_only - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert abs(g.solver.log_determinant - o.log_determinant) <= 1e-9 * abs(o.log_determinant) + 1e-9
    assert abs(g.log_likelihood(y) - o.log_likelihood(y)) <= 1e-9 * abs(o.log_likelihood(y)) + 1e-7


def test_headline_size_N2000_d10(torch_gpu):
    X, y, h = make_problem(2000, 10, 3)
    g, o = _pair(X, y, h)
    Xs = np.random.RandomState(9).uniform(-3, 3, (5000, 10))       # > 4096 -> tiled mean kernel
    mu_o, var_o = o.predict(y, Xs, return_var=True)
    mu = g.predict(y, Xs, return_cov=False)
    mu2, var = g.predict(y, Xs, return_var=True)
    amp = np.exp(h["log_amp"])
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert np.max(np.abs(mu2 - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert np.max(np.abs(var - var_o)) <= 1e-6 * amp
    small = g.predict(y, Xs[:100], return_cov=False)                # row-wise kernel
    assert np.max(np.abs(small - mu_o[:100]) / (np.abs(mu_o[:100]) + 1)) <= 1e-8


def test_kernel_get_value(torch_gpu):
    from oracle.gp_oracle import sqexp_kernel
    X, y, h = make_problem(150, 6, 7)
    g, o = _pair(X, y, h)
    X2 = np.random.RandomState(1).uniform(-3, 3, (70, 6))
    K = g.kernel.get_value(X, X2)
    np.testing.assert_allclose(K, sqexp_kernel(X, X2, h["log_amp"], h["log_M"]), rtol=1e-13, atol=1e-300)


def test_known_answers_and_properties(torch_gpu):
    X, y, h = make_problem(256, 3, 8, log_wn=-14.0)
    g, o = _pair(X, y, h)
    amp = np.exp(h["log_amp"])
    mu, var = g.predict(y, X, return_var=True)
    np.testing.assert_allclose(mu, y, atol=1e-3)                    # interpolation
    assert np.all(np.abs(var) < 1e-3 * amp)                          # variance collapses at the data
    far = np.full((3, 3), 400.0)
    mu_f, var_f = g.predict(y, far, return_var=True)
    assert np.allclose(mu_f, h["mean"], atol=1e-12) and np.allclose(var_f, amp, rtol=1e-14)
    # linearity of the mean in y (same factor): mu(a y1 + b y2) - m = a (mu(y1)-m) + b (mu(y2)-m)
    rng = np.random.RandomState(0)
    y2 = rng.randn(len(y))
    Xs = rng.uniform(-3, 3, (64, 3))
    m1 = g.predict(y, Xs, return_cov=False) - h["mean"]
    m2 = g.predict(y2, Xs, return_cov=False) - h["mean"]
    m3 = g.predict(2.0 * y - 3.0 * y2 + h["mean"] * 2.0, Xs, return_cov=False) - h["mean"]
    np.testing.assert_allclose(m3, 2.0 * m1 - 3.0 * m2, rtol=1e-7, atol=1e-7)


def test_not_positive_definite_status(torch_gpu):
    from alabi_amd import HipGP
    X = np.zeros((70, 2)); X[:, 0] = np.arange(70) * 1e-9            # numerically identical points
    g = HipGP(2, 0.0, -80.0, 0.0, [0.0, 0.0])
    with pytest.raises(np.linalg.LinAlgError):
        g.compute(X)
    assert g.compute(X, quiet=True) is False
    assert g.log_likelihood(np.zeros(70), quiet=True) == -np.inf
    g.set_parameter_vector([0.0, -2.0, 0.0, 0.0, 0.0])               # a real nugget repairs it
    assert g.compute(X) is True


def test_parameter_protocol_pickle_and_growth(torch_gpu):
    import copy
    import pickle
    X, y, h = make_problem(90, 2, 11)
    g, o = _pair(X, y, h)
    assert g.get_parameter_names() == o.get_parameter_names()
    np.testing.assert_array_equal(g.get_parameter_vector(), o.get_parameter_vector())
    Xs = np.random.RandomState(2).uniform(-3, 3, (17, 2))
    ref = g.predict(y, Xs, return_cov=False)
    g2 = pickle.loads(pickle.dumps(g))
    np.testing.assert_allclose(g2.predict(y, Xs, return_cov=False), ref, rtol=1e-12)
    g3 = copy.deepcopy(g)
    p = g3.get_parameter_vector(); p[-1] += 0.3
    g3.set_parameter_vector(p)
    o.set_parameter_vector(p); o.recompute()
    np.testing.assert_allclose(g3.predict(y, Xs, return_cov=False), o.predict(y, Xs), rtol=1e-8)
    np.testing.assert_allclose(g.predict(y, Xs, return_cov=False), ref, rtol=1e-12)   # the copy did not alias
    # capacity growth: refit on more points than the first handle held
    X2, y2, _ = make_problem(400, 2, 12)
    g.compute(X2)
    o2 = type(o)(2, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X2)
    np.testing.assert_allclose(g.predict(y2, Xs, return_cov=False), o2.predict(y2, Xs), rtol=1e-7, atol=1e-7)


def _jones_atol(mu, var, y_best, zeta=0.01):
    """EI = imp*cdf + sd*pdf cancels in the lower tail (by a factor ~z^2), and each term is exp(-z^2/2)-like,
    so a 1-ulp rounding of its argument is a relative error of ~z^2/2 * eps in the term.  Below 1e-300 the
    reference's cdf has already underflowed to 0 while erfc on the device still returns subnormals."""
    from scipy.stats import norm
    sd = np.sqrt(var); imp = mu - y_best - zeta
    with np.errstate(over="ignore", invalid="ignore"):
        z = imp / sd
        z2 = np.minimum(z * z, 1e6)
        return (2e-14 + 4.4e-16 * z2) * (np.abs(imp) * norm.cdf(z) + sd * norm.pdf(z)) + 1e-300


def test_utility_epilogue_against_golden(torch_gpu, golden):
    from alabi_amd.utility import utility_eval_device
    g = golden
    for algo, key in (("bape", "util_bape"), ("agp", "util_agp"), ("jones", "util_jones")):
        u = utility_eval_device(algo, g["util_theta"], g["util_bounds"], g["util_mu"], g["util_var"],
                                float(g["util_y_best"])).cpu().numpy()
        ref = g[key]
        assert np.array_equal(np.isnan(u), np.isnan(ref))
        inf = np.isinf(ref) & ~((g["util_var"] > 0) & (g["util_var"] < 1e-15) & (algo == "bape"))
        assert np.array_equal(u[inf], ref[inf])
        fin = np.isfinite(ref)
        # bape: the reference's literal log(1 - exp(-var)) amplifies the 1-ulp rounding of exp by 1/(1-e^-var);
        # that conditioning (not a looser kernel) is the absolute tolerance for small var.
        atol = np.full(len(ref), 1e-300)
        v = g["util_var"]
        pos = v > 0
        if algo == "bape":
            atol[pos] = 4.5e-16 / (-np.expm1(-v[pos]))
        if algo == "jones":
            atol[pos] = _jones_atol(g["util_mu"][pos], v[pos], float(g["util_y_best"]))
        if algo == "bape":
            # var within a few ulps of 0: 1 - exp(-var) is 0, 1 or 2 ulps depending on the exp implementation
            # (NumPy's SIMD exp is not correctly rounded there), so the reference value itself is a coin flip
            # between +inf and log(k eps).  Accept +inf or the same value to within log(4).
            tiny = pos & (v < 1e-15)
            ok_tiny = np.isposinf(u[tiny]) | (np.abs(u[tiny] - ref[tiny]) <= 1.4)
            assert ok_tiny.all()
            fin = fin & ~tiny
        bad = np.abs(u[fin] - ref[fin]) > 1e-12 * np.abs(ref[fin]) + atol[fin]
        assert not bad.any(), (algo, u[fin][bad][:5], ref[fin][bad][:5])


@pytest.mark.parametrize("algo", ["bape", "agp", "jones"])
def test_utility_scan_argmin(torch_gpu, algo):
    from alabi_amd.utility import utility_scan
    from oracle.utility_oracle import utility_batch
    X, y, h = make_problem(400, 4, 21)
    g, o = _pair(X, y, h)
    bounds = np.array([[-3.0, 3.0]] * 4)
    cand = np.random.RandomState(5).uniform(-3.3, 3.3, (20000, 4))    # some outside the box
    best, val, idx, u, mu, var = utility_scan(g, y, cand, bounds, algo, y_best=float(y.max()), return_all=True)
    mu_o, var_o = o.predict(y, cand, return_var=True)
    u_dev = u.cpu().numpy()
    # epilogue applied to the DEVICE (mu, var) must match the oracle formula exactly; arg-min over finite values
    u_chk = utility_batch(algo, mu.cpu().numpy(), var.cpu().numpy(), cand, bounds, float(y.max()))
    fin = np.isfinite(u_chk)
    assert np.array_equal(np.isfinite(u_dev), fin)
    atol = _jones_atol(mu.cpu().numpy()[fin], np.maximum(var.cpu().numpy()[fin], 1e-300), float(y.max())) if algo == "jones" else 0.0
    assert np.all(np.abs(u_dev[fin] - u_chk[fin]) <= 1e-12 * np.abs(u_chk[fin]) + atol)
    assert idx == int(np.flatnonzero(fin)[np.argmin(u_dev[fin])]) and val == u_dev[idx]
    np.testing.assert_array_equal(best, cand[idx])
    # and the whole scan agrees with the oracle's own (mu, var) to conditioning
    u_o = utility_batch(algo, mu_o, var_o, cand, bounds, float(y.max()))
    both = fin & np.isfinite(u_o)
    assert both.sum() > 0.5 * len(cand)
    assert abs(u_dev[idx] - np.min(u_o[both])) < 1e-5 * (1 + abs(u_dev[idx]))


def test_full_size_configs_properties(torch_gpu):
    """BASELINE.json's largest sizes through size-independent properties (the oracle is only run on a slice)."""
    from alabi_amd import HipGP
    from alabi_amd.utility import utility_scan
    from alabi_amd.workloads import make_config
    from oracle.gp_oracle import OracleGP
    # C4: N=5000, d=10 -- oracle on 200 query points
    cfg = make_config("C4"); h = cfg["hyper"]
    g = HipGP(10, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(cfg["X"])
    o = OracleGP(10, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"])
    Xs = np.random.RandomState(0).uniform(-3, 3, (200, 10))
    mu, var = g.predict(cfg["y"], Xs, return_var=True)
    mu_o, var_o = o.predict(cfg["y"], Xs, return_var=True)
    amp = np.exp(h["log_amp"])
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8 and np.max(np.abs(var - var_o)) <= 1e-6 * amp
    del g, o
    # C5: N=10000, d=20 ARD -- properties only
    cfg = make_config("C5"); h = cfg["hyper"]; X, y = cfg["X"], cfg["y"]; amp = np.exp(h["log_amp"])
    g = HipGP(20, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    idx = np.random.RandomState(1).choice(len(X), 500, replace=False)
    mu, var = g.predict(y, X[idx], return_var=True)
    assert np.max(np.abs(mu - y[idx])) < 1e-3 * (1 + np.max(np.abs(y)))          # interpolates its training data
    assert np.all(np.abs(var) < 1e-3 * amp)                                        # variance collapses there
    far = np.full((4, 20), 300.0)
    mu_f, var_f = g.predict(y, far, return_var=True)
    assert np.allclose(mu_f, h["mean"], atol=1e-9) and np.allclose(var_f, amp, rtol=1e-13)   # prior far away
    Xq = np.random.RandomState(2).uniform(-3, 3, (4096 + 64, 20))
    m1 = g.predict(y, Xq, return_cov=False); m2, v2 = g.predict(y, Xq, return_var=True)
    assert np.max(np.abs(m1 - m2)) <= 1e-9 * (1 + np.max(np.abs(m1)))             # tile-mean == variance-path mean
    assert np.all(v2 <= amp * (1 + 1e-12)) and np.all(v2 > -1e-6 * amp)            # 0 <= var <= amp up to rounding
    m3, v3 = g.predict(y, Xq, return_var=True)
    assert np.array_equal(m2, m3) and np.array_equal(v2, v3)                       # deterministic
    best, val, i = utility_scan(g, y, Xq, cfg["bounds"], "bape")
    u_i = -((2 * m2[i] + v2[i]) + (v2[i] + np.log(1 - np.exp(-v2[i]))))
    assert i >= 0 and abs(val - u_i) <= 1e-9 * abs(u_i)
    L = g.solver.get_factor()
    assert float(L.diagonal().min()) > 0 and abs(g.solver.log_determinant - 2 * float(L.diagonal().log().sum())) < 1e-6


def test_grad_log_likelihood_analytic():
    """alabi_gp_grad_log_likelihood vs the oracle's analytic gradient (sq-exp) and vs central differences of the device
    likelihood (all four kernels); N not a multiple of 64 so the padded rows are exercised."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(333, 5, 21, log_wn=-6.0)
    g = HipGP(5, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(5, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    ga, go = g.grad_log_likelihood(y), o.grad_log_likelihood(y)
    assert ga.shape == go.shape == (8,)
    np.testing.assert_allclose(ga, go, rtol=1e-8, atol=1e-8 * np.max(np.abs(go)))
    for kernel in ("Matern32Kernel", "Matern52Kernel", "RationalQuadraticKernel"):
        gk = HipGP(5, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], kernel=kernel, log_alpha=0.3)
        gk.compute(X)
        an, fd = gk.grad_log_likelihood(y), gk.grad_log_likelihood_fd(y, h=1e-5)
        assert an.shape == fd.shape
        np.testing.assert_allclose(an, fd, rtol=2e-5, atol=2e-5 * np.max(np.abs(fd)))
    # frozen parameters drop out of the vector (george protocol)
    gf = HipGP(5, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], fit_mean=False, fit_white_noise=False)
    gf.compute(X)
    np.testing.assert_allclose(gf.grad_log_likelihood(y), ga[2:], rtol=1e-12)


@pytest.mark.parametrize("w_path", ["1", "0"])
@pytest.mark.parametrize("N", [1, 63, 64, 65, 128, 200, 256, 257, 320, 400, 700])
def test_predict_variance_block_row_boundaries(N, w_path, monkeypatch):
    """Both wave-specialised variance kernels -- the product with the cached L^-1 (block rows split over workgroups) and the
    substitution kernel (block rows 0..3 one stage at a time, later rows two stages ahead) -- at training-set sizes on both
    sides of every boundary (1, 2, 4, 5, 7, 11 block rows, ragged last block)."""
    monkeypatch.setenv("ALABI_PV_W", w_path)
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(N, 3, 100 + N, log_wn=-8.0)
    if N == 1:
        h["log_amp"] = 0.3
    g = HipGP(3, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(3, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    Xs = np.random.RandomState(N).uniform(-3.2, 3.2, (1000, 3))          # 16 tiles, the last one ragged
    mu, var = g.predict(y, Xs, return_var=True)
    mu_o, var_o = o.predict(y, Xs, return_var=True)
    amp = np.exp(h["log_amp"])
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-9
    assert np.max(np.abs(var - var_o)) <= 1e-7 * amp


def test_predict_variance_paths_agree(monkeypatch):
    """Chunked launches (several rounds of tiles), the legacy kernel (ALABI_PV_LEGACY=1, also used for d > 16) and the
    default path give the same variances to rounding; d = 20 (cached-inverse path with a wide K* pre-pass) and d = 40 (legacy
    dispatch on its own) against the oracle."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(500, 6, 8, log_wn=-8.0)
    Xs = np.random.RandomState(3).uniform(-3, 3, (5000, 6))
    out = {}
    for tag, env in (("default", {}), ("chunked", {"ALABI_PV_CHUNK_TILES": "7"}), ("legacy", {"ALABI_PV_LEGACY": "1"}),
                     ("substitution", {"ALABI_PV_W": "0"}), ("substitution_chunked", {"ALABI_PV_W": "0", "ALABI_PV_CHUNK_TILES": "7"})):
        for k in ("ALABI_PV_CHUNK_TILES", "ALABI_PV_LEGACY", "ALABI_PV_W"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = HipGP(6, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
        out[tag] = g.predict(y, Xs, return_var=True)
    amp = np.exp(h["log_amp"])
    # a chunk of 7 tiles splits each tile's block rows over more workgroups than the default launch: same sums, other order
    assert np.max(np.abs(out["default"][1] - out["chunked"][1])) <= 1e-12 * amp
    # (the K* pre-pass splits the training points over more workgroups when a chunk has few tiles: same sum, other grouping)
    assert np.max(np.abs(out["default"][0] - out["chunked"][0])) <= 1e-13 * (np.max(np.abs(out["default"][0])) + 1)
    assert np.max(np.abs(out["default"][1] - out["legacy"][1])) <= 1e-9 * amp
    assert np.max(np.abs(out["default"][0] - out["legacy"][0])) <= 1e-9 * (np.max(np.abs(out["legacy"][0])) + 1)
    assert np.max(np.abs(out["default"][1] - out["substitution"][1])) <= 1e-9 * amp
    np.testing.assert_array_equal(out["substitution"][1], out["substitution_chunked"][1])
    for k in ("ALABI_PV_CHUNK_TILES", "ALABI_PV_LEGACY", "ALABI_PV_W"):
        monkeypatch.delenv(k, raising=False)
    for dd in (20, 40):
        X2, y2, h2 = make_problem(300, dd, 9, log_wn=-8.0, ell2=float(dd))
        g2 = HipGP(dd, h2["mean"], h2["log_white_noise"], h2["log_amp"], h2["log_M"]); g2.compute(X2)
        o2 = OracleGP(dd, h2["mean"], h2["log_white_noise"], h2["log_amp"], h2["log_M"]).compute(X2)
        for Mq in (300, 9):
            Xs2 = np.random.RandomState(4).uniform(-3, 3, (Mq, dd))
            mu2, var2 = g2.predict(y2, Xs2, return_var=True)
            mu_o, var_o = o2.predict(y2, Xs2, return_var=True)
            assert np.max(np.abs(mu2 - mu_o) / (np.abs(mu_o) + 1)) <= 1e-9
            assert np.max(np.abs(var2 - var_o)) <= 1e-7 * np.exp(h2["log_amp"])


@pytest.mark.parametrize("N,d", [(300, 4), (1000, 10), (257, 1)])
def test_small_batch_variance_path(N, d, monkeypatch):
    """At most 16 queries take the cached-L^-1 path (one multiply spread over the block rows); it must agree with the oracle
    and with the substitution kernel, survive a refit (the cache follows the factor) and a different training-set size."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(N, d, 40 + N, log_wn=-8.0)
    amp = np.exp(h["log_amp"])
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    rs = np.random.RandomState(N)
    for M in (1, 5, 16):
        Xs = rs.uniform(-3.1, 3.1, (M, d))
        mu, var = g.predict(y, Xs, return_var=True)
        mu_o, var_o = o.predict(y, Xs, return_var=True)
        assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-9
        assert np.max(np.abs(var - var_o)) <= 1e-7 * amp
        monkeypatch.setenv("ALABI_PV_SMALL", "0")
        mu_s, var_s = g.predict(y, Xs, return_var=True)
        monkeypatch.delenv("ALABI_PV_SMALL")
        assert np.max(np.abs(var - var_s)) <= 1e-8 * amp and np.max(np.abs(mu - mu_s)) <= 1e-9 * (np.max(np.abs(mu_s)) + 1)
    # refit with other hyper-parameters and fewer points: the cached inverse must be rebuilt
    p = g.get_parameter_vector(); p[-1] += 0.3
    g.set_parameter_vector(p); o.set_parameter_vector(p)
    g.compute(X[: N - 70]); o.compute(X[: N - 70])
    Xs = rs.uniform(-3.1, 3.1, (7, d))
    mu, var = g.predict(y[: N - 70], Xs, return_var=True)
    mu_o, var_o = o.predict(y[: N - 70], Xs, return_var=True)
    assert np.max(np.abs(var - var_o)) <= 1e-7 * amp and np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-9


def test_append_point_matches_full_factorisation():
    """HipGP.compute_from takes over the previous factor and appends one row (alabi_gp_append): factor, log-determinant,
    predictions (all three variance paths), likelihood and gradient must match a full factorisation / the oracle; the append
    is refused (full refit) when the padding rows are used up, and a point that breaks positive definiteness leaves the old
    factor usable."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    d = 3
    X, y, h = make_problem(200, d, 55, log_wn=-8.0)
    amp = np.exp(h["log_amp"])
    mk = lambda: HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])  # noqa: E731
    g = mk(); g.compute(X[:120])
    appended = 0
    for n in range(121, 200):
        g2 = mk()
        g2.compute_from(g, X[:n])
        if getattr(g2, "appended", 0) > appended:
            appended = g2.appended
        g = g2
        if n in (121, 128, 129, 150, 192, 193, 199):
            o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X[:n])
            Lf = g.solver.get_factor().cpu().numpy()
            assert np.max(np.abs(Lf - o._L)) <= 1e-9 * np.max(np.abs(o._L))
            for M in (7, 300, 5000):
                Xs = np.random.RandomState(n + M).uniform(-3, 3, (M, d))
                mu, var = g.predict(y[:n], Xs, return_var=True)
                mu_o, var_o = o.predict(y[:n], Xs, return_var=True)
                assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
                assert np.max(np.abs(var - var_o)) <= 1e-7 * amp
            assert abs(g.log_likelihood(y[:n]) - o.log_likelihood(y[:n])) <= 1e-8 * abs(o.log_likelihood(y[:n]))
            np.testing.assert_allclose(g.grad_log_likelihood(y[:n]), o.grad_log_likelihood(y[:n]), rtol=1e-6,
                                       atol=1e-7 * np.max(np.abs(o.grad_log_likelihood(y[:n]))))
    # 79 steps, refits at the two 64-row boundaries (128 -> 129, 192 -> 193) only... counted through the chain of objects
    assert appended >= 60
    # a duplicate of an existing point with a negligible nugget: whatever the extended factor does (refused append + failed or
    # barely passing refit), the previous object must stay usable (it kept its factor or rebuilds it)
    gd = HipGP(d, h["mean"], -40.0, h["log_amp"], h["log_M"]); gd.compute(X[:100])
    gbad = HipGP(d, h["mean"], -40.0, h["log_amp"], h["log_M"])
    try:
        gbad.compute_from(gd, np.vstack([X[:100], X[:1]]))
    except np.linalg.LinAlgError:
        pass
    mu = gd.predict(y[:100], X[:5], return_cov=False)
    assert np.all(np.isfinite(mu))


@pytest.mark.parametrize("N,d,M,kernel", [(40, 3, 24, "ExpSquaredKernel"), (500, 5, 7, "ExpSquaredKernel"),
                                          (2000, 10, 33, "ExpSquaredKernel"), (300, 4, 16, "Matern52Kernel"),
                                          (300, 4, 5, "RationalQuadraticKernel"), (130, 20, 3, "ExpSquaredKernel")])
def test_predict_grad_against_oracle(torch_gpu, N, d, M, kernel):
    """alabi_gp_predict_grad (closed-form d mu/dx, d var/dx through the cached L^-1) against the oracle: the closed form
    for the squared exponential (1e-9 of the gradient scale) and the reference-shaped finite differences
    (utility.py:511-623, step 1e-6) for every kernel family (1e-5)."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    from oracle import utility_oracle as uo
    X, y, h = make_problem(N, d, 11 + d)
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], kernel=kernel)
    g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], kernel=kernel).compute(X)
    Xs = np.random.RandomState(5).uniform(-2.5, 2.5, (M, d))
    mu, var, dmu, dvar = [t.cpu().numpy() for t in g.predict_grad_device(y, Xs)]
    mu_o, var_o = o.predict(y, Xs, return_var=True)
    amp = np.exp(h["log_amp"])
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8
    assert np.max(np.abs(var - var_o)) <= 1e-6 * amp
    mu2, var2 = g.predict(y, Xs, return_var=True)                 # same numbers as the predict entry point
    assert np.max(np.abs(mu - mu2)) <= 1e-10 * (np.max(np.abs(mu2)) + 1) and np.max(np.abs(var - var2)) <= 1e-10 * amp
    o._compute_alpha(y)
    nq = min(M, 6)
    fd_mu = np.array([uo.grad_gp_mean_prediction(t, o) for t in Xs[:nq]])
    fd_var = np.array([uo.grad_gp_var_prediction(t, o) for t in Xs[:nq]])
    assert np.max(np.abs(dmu[:nq] - fd_mu)) <= 1e-5 * np.max(np.abs(fd_mu))
    assert np.max(np.abs(dvar[:nq] - fd_var)) <= 1e-5 * np.max(np.abs(fd_var)) + 1e-7 * amp
    if kernel == "ExpSquaredKernel":
        _, _, dmu_o, dvar_o = uo.analytic_predict_grad(o, y, Xs)
        assert np.max(np.abs(dmu - dmu_o)) <= 1e-9 * np.max(np.abs(dmu_o))
        assert np.max(np.abs(dvar - dvar_o)) <= 1e-9 * np.max(np.abs(dvar_o)) + 1e-9 * amp


def test_grad_utilities_against_reference_vectors(torch_gpu, golden_grad):
    """grad_gp_mean_prediction / grad_gp_var_prediction / grad_agp_utility / grad_bape_utility of the product (GPU, closed
    form) against the arrays the reference's own functions returned (finite differences, step 1e-6)."""
    from alabi_amd import HipGP
    from alabi_amd import utility as ut
    g = golden_grad
    gp = HipGP(g["grad_X"].shape[1], float(g["grad_mean"]), float(g["grad_log_wn"]), float(g["grad_log_amp"]), g["grad_log_M"])
    gp.compute(g["grad_X"])
    gp.predict(g["grad_y"], g["grad_X"][:1], return_cov=False)      # sets gp._y as the reference's callers do
    th, b = g["grad_theta"], g["grad_bounds"]
    dmu = np.array([ut.grad_gp_mean_prediction(t, gp) for t in th])
    dvar = np.array([ut.grad_gp_var_prediction(t, gp) for t in th])
    assert np.max(np.abs(dmu - g["grad_dmu"])) <= 1e-6 * np.max(np.abs(g["grad_dmu"]))
    assert np.max(np.abs(dvar - g["grad_dvar"])) <= 1e-6 * np.max(np.abs(g["grad_dvar"]))
    agp = np.array([ut.grad_agp_utility(t, gp, b) for t in th])
    bape = np.array([ut.grad_bape_utility(t, gp, b) for t in th])
    assert np.array_equal(np.isinf(agp), np.isinf(g["grad_agp"])) and np.array_equal(np.isinf(bape), np.isinf(g["grad_bape"]))
    fin = np.isfinite(g["grad_agp"][:, 0])
    assert np.max(np.abs(agp[fin] - g["grad_agp"][fin])) <= 1e-6 * np.max(np.abs(g["grad_agp"][fin]))
    _, var = gp.predict(g["grad_y"], th, return_var=True)
    ok = fin & (var > 1e-9 * np.exp(float(g["grad_log_amp"])))
    assert ok.sum() >= 15
    assert np.max(np.abs(bape[ok] - g["grad_bape"][ok]) / (np.abs(g["grad_bape"][ok]) + 1.0)) <= 1e-5


@pytest.mark.parametrize("N,d,log_wn", [(130, 3, -12.0), (1000, 6, -10.0), (64, 2, -8.0), (1999, 10, -12.0)])
def test_get_inverse_is_native_and_matches_cho_solve(torch_gpu, N, d, log_wn):
    """gp.solver.get_inverse() (reference: alabi/utility.py:610) = alabi_gp_get_inverse (K^-1 = W^T W on the matrix cores):
    symmetric, K K^-1 = I to rounding x condition, and equal to scipy's cho_solve(L, I) of the oracle within the same bound."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(N, d, 40 + d, log_wn=log_wn)
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    Kinv = g.solver.get_inverse()
    assert Kinv.shape == (N, N) and np.array_equal(Kinv, Kinv.T)
    K = o.get_matrix(X)
    Kinv_o = o.get_inverse()
    # both inverses carry an error of order eps * cond(K) * |K^-1|: compare residuals, and each other within that bound
    eps_cond = np.finfo(float).eps * np.linalg.cond(K)
    res = np.max(np.abs(K @ Kinv - np.eye(N)))
    res_o = np.max(np.abs(K @ Kinv_o - np.eye(N)))
    assert res <= max(20.0 * res_o, 1e-9), (res, res_o)
    assert np.max(np.abs(Kinv - Kinv_o)) <= 50.0 * eps_cond * np.max(np.abs(Kinv_o)), (eps_cond,)
    # quadratic forms (what the reference's variance gradient takes): k^T K^-1 k against the variance of the predict path
    Xs = np.random.RandomState(3).uniform(-2.5, 2.5, (5, d))
    ks = g.kernel.get_value(Xs, X)
    _, var = g.predict(y, Xs, return_var=True)
    amp = np.exp(h["log_amp"])
    assert np.max(np.abs((amp - np.einsum("qi,ij,qj->q", ks, Kinv, ks)) - var)) <= 1e-6 * amp


def test_reference_finite_difference_gradients_through_native_get_inverse(torch_gpu, golden_grad):
    """The reference's finite-difference route for d var / dx (utility.py:586-623: numerical_kernel_gradient + solver.get_inverse(),
    restated in oracle/utility_oracle.py) run on the HipGP protocol members -- kernel.get_value, solver.get_inverse, _x, _alpha --
    against the arrays the reference's own functions returned on george-free inputs (tests/golden/make_golden_grad.py)."""
    from alabi_amd import HipGP
    from oracle import utility_oracle as uo
    g = golden_grad
    gp = HipGP(g["grad_X"].shape[1], float(g["grad_mean"]), float(g["grad_log_wn"]), float(g["grad_log_amp"]), g["grad_log_M"])
    gp.compute(g["grad_X"])
    gp.predict(g["grad_y"], g["grad_X"][:1], return_cov=False)

    class Protocol:                                          # the members the oracle's restatement touches, served by HipGP
        _x = gp._x
        _alpha = gp._alpha
        _k = staticmethod(lambda a, b: gp.kernel.get_value(a, b))
        get_inverse = staticmethod(gp.solver.get_inverse)
    th = g["grad_theta"][:8]
    dvar = np.array([uo.grad_gp_var_prediction(t, Protocol) for t in th])
    dmu = np.array([uo.grad_gp_mean_prediction(t, Protocol) for t in th])
    assert np.max(np.abs(dmu - g["grad_dmu"][:8])) <= 1e-6 * np.max(np.abs(g["grad_dmu"]))
    assert np.max(np.abs(dvar - g["grad_dvar"][:8])) <= 1e-5 * np.max(np.abs(g["grad_dvar"]))


@pytest.mark.parametrize("N,d,M", [(200, 3, 2048 + 37), (1000, 10, 4096), (777, 5, 20001), (2000, 10, 40000)])
def test_two_wave_variance_kernel(N, d, M, monkeypatch):
    """predict_var_w2_kernel (128 queries per workgroup, two MFMA waves per SIMD; taken from 2048 queries on) against the
    oracle and against the 64-query kernel (ALABI_PV_W2=0): odd tile counts (the last tile is paired with itself), block
    rows split over several workgroups (few groups), several rounds per workgroup, chunked launches."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(N, d, 21 + d, log_wn=-9.0)
    Xs = np.random.RandomState(7).uniform(-3, 3, (M, d))
    amp = np.exp(h["log_amp"])
    out = {}
    for tag, env in (("w2", {"ALABI_PV_W": "1"}), ("w1", {"ALABI_PV_W": "1", "ALABI_PV_W2": "0"}),
                     ("w2_chunked", {"ALABI_PV_W": "1", "ALABI_PV_CHUNK_TILES": "66"})):
        for k in ("ALABI_PV_W", "ALABI_PV_W2", "ALABI_PV_CHUNK_TILES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
        out[tag] = g.predict(y, Xs, return_var=True)
    assert np.max(np.abs(out["w2"][1] - out["w1"][1])) <= 1e-12 * amp          # same sums, possibly another split of the rows
    assert np.max(np.abs(out["w2"][1] - out["w2_chunked"][1])) <= 1e-12 * amp
    np.testing.assert_array_equal(out["w2"][0], out["w1"][0])
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    sel = np.random.RandomState(1).choice(M, 1500, replace=False)
    sel[:3] = [M - 1, M - 64, 0]
    mu_o, var_o = o.predict(y, Xs[sel], return_var=True)
    assert np.max(np.abs(out["w2"][0][sel] - mu_o) / (np.abs(mu_o) + 1)) <= 1e-9
    assert np.max(np.abs(out["w2"][1][sel] - var_o)) <= 1e-7 * amp


@pytest.mark.parametrize("N", [130, 257, 520, 777, 1000])
def test_block_recursive_inverse_matches_substitution_chains(N, monkeypatch):
    """W = L^-1 from the recursive block inversion (gp_inverse.hip; block counts that are not powers of two: 3, 5, 9, 13, 16)
    gives the same variances as W from the substitution chains (ALABI_WINV_DNC=0) and as the oracle."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    d = 4
    X, y, h = make_problem(N, d, 31, log_wn=-9.0)
    Xs = np.random.RandomState(2).uniform(-3, 3, (333, d))
    amp = np.exp(h["log_amp"])
    out = {}
    monkeypatch.setenv("ALABI_PV_W", "1")
    for tag in ("1", "0"):
        monkeypatch.setenv("ALABI_WINV_DNC", tag)
        g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
        out[tag] = g.predict(y, Xs, return_var=True)[1]
        out[tag + "s"] = g.predict(y, Xs[:7], return_var=True)[1]                   # small-batch kernel on the same W
    _, var_o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X).predict(y, Xs, return_var=True)
    assert np.max(np.abs(out["1"] - out["0"])) <= 1e-11 * amp
    assert np.max(np.abs(out["1s"] - out["0s"])) <= 1e-11 * amp
    assert np.max(np.abs(out["1"] - var_o)) <= 1e-7 * amp


@pytest.mark.parametrize("N,d", [(300, 3), (1000, 10)])
def test_medium_batches_groups_and_split_prepass(N, d):
    """17 ... 128 queries go through predict_var_small_kernel in groups of 16 (last group partial); 129 ... 2047 through the tile
    kernels with the K* pre-pass split over the training points; mean-only batches above 4096 queries through the split
    predict_mean_tile_kernel + mean_combine_kernel.  All against the oracle."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(N, d, 41, log_wn=-9.0)
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    amp = np.exp(h["log_amp"])
    rng = np.random.RandomState(9)
    for M in (17, 40, 128, 129, 700, 1999):
        Xs = rng.uniform(-3, 3, (M, d))
        mu, var = g.predict(y, Xs, return_var=True)
        mu_o, var_o = o.predict(y, Xs, return_var=True)
        assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-9, M
        assert np.max(np.abs(var - var_o)) <= 1e-7 * amp, M
    Xs = rng.uniform(-3, 3, (5003, d))
    mu = g.predict(y, Xs, return_cov=False)
    mu_o = o.predict(y, Xs)
    assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) <= 1e-9


@pytest.mark.parametrize("M", [2048 + 37, 10000, 32768 + 4 * 64 + 37, 800000])
@pytest.mark.parametrize("d,N,kernel", [(10, 2000, "ExpSquaredKernel"), (2, 130, "ExpSquaredKernel"), (20, 700, "ExpSquaredKernel"),
                                        (5, 500, "Matern52Kernel"), (30, 300, "RationalQuadraticKernel")])
def test_predict_mean_matrix_core_path(torch_gpu, monkeypatch, d, N, kernel, M):
    """Batches of >= 2048 queries form q.x on the matrix cores (predict_mean_mfma_kernel, the exponent as ONE augmented dot
    product): against the vector kernel on every point and against the oracle on a slice; ragged last workgroup included.
    Medium batches split the training points over several workgroups per query block (partial sums added by
    mean_combine_kernel): 2085 and 10^4 queries; 800 000 run unsplit."""
    if M > 100000 and N != 130:
        pytest.skip("the unsplit launch is exercised once")
    import torch
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(N, d, 40 + d)
    kw = dict(kernel=kernel, log_alpha=0.4)
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], **kw); g.compute(X)
    Xs = np.random.RandomState(d).uniform(-3.2, 3.2, (M, d))
    Xs[:N] = X                                                   # queries ON training points: r2 = 0 exactly in exact arithmetic
    mu_m = g.predict(y, Xs, return_cov=False)
    monkeypatch.setenv("ALABI_PM_MFMA", "0")
    mu_v = g.predict(y, Xs, return_cov=False)
    monkeypatch.delenv("ALABI_PM_MFMA")
    assert mu_m.shape == mu_v.shape == (M,)
    assert np.max(np.abs(mu_m - mu_v) / (np.abs(mu_v) + 1)) <= 1e-9
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"], **kw).compute(X)
    pick = np.concatenate([np.arange(64), np.random.RandomState(1).choice(M, 400, replace=False), [M - 1]])
    mu_o = o.predict(y, Xs[pick])
    assert np.max(np.abs(mu_m[pick] - mu_o) / (np.abs(mu_o) + 1)) <= 1e-8


def test_predict_grad_point_host_buffers(torch_gpu):
    """alabi_gp_predict_grad_point (one point, host buffers on both sides: what the polish step of find_next_point calls per
    evaluation, alabi/utility.py:1030-1163) returns exactly what the batched device entry returns for that point; several points
    take the device route."""
    import torch
    from alabi_amd import HipGP
    for N, d in ((90, 2), (700, 5), (2000, 10)):
        X, y, h = make_problem(N, d, 300 + N)
        g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
        rng = np.random.RandomState(N)
        pts = rng.uniform(-1.0, 1.0, (5, d))
        ref = [t.cpu().numpy() for t in g.predict_grad_device(y, pts)]
        for q in range(5):
            for x in (pts[q], pts[q:q + 1], list(pts[q])):
                mu, var, dmu, dvar = g.predict_grad_host(y, x)
                assert mu.shape == (1,) and var.shape == (1,) and dmu.shape == (1, d) and dvar.shape == (1, d)
                assert mu[0] == ref[0][q] and var[0] == ref[1][q]
                np.testing.assert_array_equal(dmu[0], ref[2][q]); np.testing.assert_array_equal(dvar[0], ref[3][q])
        many = g.predict_grad_host(y, pts)
        for a, b in zip(many, ref):
            np.testing.assert_array_equal(a, b)
        y2 = y + 0.5                                          # a new y: alpha is recomputed before the point is evaluated
        mu2 = g.predict_grad_host(y2, pts[0])[0][0]
        assert mu2 == g.predict_grad_device(y2, pts[:1])[0].cpu().numpy()[0] and mu2 != ref[0][0]


@pytest.mark.parametrize("N,panel", [(705, "4"), (1500, "4"), (1100, "2"), (3200, "4"), (2900, "8"), (1700, "6")])
def test_cholesky_panel_path(torch_gpu, monkeypatch, N, panel):
    """Panels of 2 / 4 block columns with one rank-128 / rank-256 trailing update (syrk_panel_kernel, 128 x 128 tiles) and
    look-ahead on a second stream: the factor reproduces K and equals the rank-64 factorisation to rounding (the path is the
    default from 128 block columns on; test_C5_* in test_gpu_configs.py run it at N = 10000)."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(N, 6, 50 + N)
    o = OracleGP(6, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    K = o.get_matrix(X)
    facs = {}
    monkeypatch.setenv("ALABI_CHOL_TASKS", "0")                  # these sizes would take the task-queue kernel by default
    for tag, env, la in (("panel", panel, "1"), ("panel-serial", panel, "0"), ("rank64", "0", "1")):
        if env is None:
            monkeypatch.delenv("ALABI_CHOL_PANEL", raising=False)
        else:
            monkeypatch.setenv("ALABI_CHOL_PANEL", env)
        monkeypatch.setenv("ALABI_CHOL_LOOKAHEAD", la)
        g = HipGP(6, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
        assert g.solver.factor_path == "steps"
        L = g.solver.get_factor().cpu().numpy()
        assert np.allclose(np.triu(L, 1), 0.0)
        assert np.max(np.abs(L @ L.T - K)) <= 1e-12 * np.max(np.abs(K)), tag
        facs[tag] = (L, g.log_likelihood(y), g.predict(y, X[:64] + 0.01, return_var=True))
    assert np.array_equal(facs["panel"][0], facs["panel-serial"][0])             # the look-ahead changes no bit
    assert np.max(np.abs(facs["panel"][0] - facs["rank64"][0])) <= 1e-9 * np.max(np.abs(facs["rank64"][0]))
    assert abs(facs["panel"][1] - o.compute(X).log_likelihood(y)) <= 1e-9 * abs(facs["rank64"][1])
    assert np.max(np.abs(facs["panel"][2][1] - facs["rank64"][2][1])) <= 1e-8 * np.exp(h["log_amp"])


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("N", [130, 200, 705, 2000, 4096, 5200])
def test_cholesky_task_queue(torch_gpu, monkeypatch, N, waves):
    """Up to 256 block columns the factorisation is ONE launch: persistent workgroups draw tile tasks (chain / panel solve /
    update) from a queue and hand tiles over through versioned write-through stores.  Same factor as the launch-per-step path
    to rounding, K reproduced, a non-positive-definite matrix reported with LAPACK's pivot index, and a wait that runs out
    falls back to the step-by-step path with the same result.  Both kernels at every size: four waves per workgroup
    (chol_tasks_kernel, the reference) and eight (chol_tasks8_kernel, the default: four helper waves in the updates)."""
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    monkeypatch.setenv("ALABI_CHOL_W8", "1" if waves == 8 else "0")
    X, y, h = make_problem(N, 6, 70 + N)
    o = OracleGP(6, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    K = o.get_matrix(X)
    facs = {}
    for tag, env in (("queue", "1"), ("steps", "0")):
        monkeypatch.setenv("ALABI_CHOL_TASKS", env)
        g = HipGP(6, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
        assert g.solver.factor_path == tag                    # (an earlier time-out's penalty does not apply when the queue is forced)
        L = g.solver.get_factor().cpu().numpy()
        assert np.allclose(np.triu(L, 1), 0.0)
        assert np.max(np.abs(L @ L.T - K)) <= 1e-12 * np.max(np.abs(K)), tag
        facs[tag] = (L, g.log_likelihood(y), g.predict(y, X[:64] + 0.01, return_var=True))
    assert np.max(np.abs(facs["queue"][0] - facs["steps"][0])) <= 1e-10 * np.max(np.abs(facs["steps"][0]))
    assert abs(facs["queue"][1] - facs["steps"][1]) <= 1e-10 * abs(facs["steps"][1])
    assert np.max(np.abs(facs["queue"][2][1] - facs["steps"][2][1])) <= 1e-9 * np.exp(h["log_amp"])
    monkeypatch.setenv("ALABI_CHOL_TASKS", "1")
    if N == 705:
        # not positive definite: numerically identical points without a nugget -> reported, not hung
        Xd = np.zeros((N, 6)); Xd[:, 0] = np.arange(N) * 1e-9
        for env in ("1", "0"):
            monkeypatch.setenv("ALABI_CHOL_TASKS", env)
            gd = HipGP(6, 0.0, -80.0, 0.0, np.zeros(6))
            with pytest.raises(np.linalg.LinAlgError):
                gd.compute(Xd)
            assert gd.compute(Xd, quiet=True) is False
        # a wait that runs out: fall back, same factor
        monkeypatch.setenv("ALABI_CHOL_TASKS", "1")
        monkeypatch.setenv("ALABI_CHOL_SPIN_LIMIT", "1")
        g2 = HipGP(6, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g2.compute(X)
        assert g2.solver.factor_path == "queue-timeout"
        assert np.array_equal(g2.solver.get_factor().cpu().numpy(), facs["steps"][0])
        monkeypatch.delenv("ALABI_CHOL_SPIN_LIMIT")


def test_cholesky_task_queue_four_and_eight_waves_same_bits(torch_gpu, monkeypatch):
    """chol_tasks8_kernel splits a 64 x 64 update tile over eight waves (16 rows x 32 columns each) instead of four (16 x 64):
    every output element still receives its k-steps in the same order, so the factors are bit-identical (N = 3000: 47 block
    columns, grouped updates over 8 block columns and a near band of one-column updates)."""
    from alabi_amd import HipGP
    X, y, h = make_problem(3000, 5, 11)
    monkeypatch.setenv("ALABI_CHOL_TASKS", "1")
    facs = []
    # four waves; eight waves with one / two (UPDATE2, the default at this size) / 2 x 2 (UPDATE4, the default from 100 block columns
    # on) tiles per grouped update: every output element receives its k-steps in the same order in all of them
    for w8, two, four in (("0", "1", "0"), ("1", "0", "0"), ("1", "1", "0"), ("1", "1", "1")):
        monkeypatch.setenv("ALABI_CHOL_W8", w8)
        monkeypatch.setenv("ALABI_CHOL_UPDATE2", two)
        monkeypatch.setenv("ALABI_CHOL_UPDATE4", four)
        g = HipGP(5, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
        facs.append(g.solver.get_factor().cpu().numpy())
    for f in facs[1:]:
        assert np.array_equal(facs[0], f)
