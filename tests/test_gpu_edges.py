"""Edge cases of the C ABI and the host mirror on the GPU: tiny / ragged / maximal sizes, empty batches,
bad arguments, NaN inputs, candidates that all fail."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu


def _pair(X, y, h):
    from alabi_amd import HipGP
    from oracle.gp_oracle import OracleGP
    d = X.shape[1]
    g = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    return g, o


@pytest.mark.parametrize("N,d", [(1, 1), (2, 3), (63, 2), (64, 64), (129, 33), (200, 48)])
def test_tiny_ragged_and_max_dimension(N, d):
    X, y, h = make_problem(N, d, N + d, log_wn=-6.0, ell2=4.0 * d)
    if not np.isfinite(h["log_amp"]):        # a single training point has var(y) = 0
        h["log_amp"] = 0.3
    g, o = _pair(X, y, h)
    for M in (1, 63, 65):
        Xs = np.random.RandomState(M).uniform(-3, 3, (M, d))
        mu, var = g.predict(y, Xs, return_var=True)
        mu_o, var_o = o.predict(y, Xs, return_var=True)
        assert np.max(np.abs(mu - mu_o) / (np.abs(mu_o) + 1)) < 1e-9
        assert np.max(np.abs(var - var_o)) < 1e-8 * np.exp(h["log_amp"])
        np.testing.assert_allclose(g.predict(y, Xs, return_cov=False), mu_o, rtol=1e-9, atol=1e-9)


def test_ragged_last_tile_of_the_tiled_mean_kernel():
    X, y, h = make_problem(100, 3, 5)
    g, o = _pair(X, y, h)
    Xs = np.random.RandomState(0).uniform(-3, 3, (4097 + 37, 3))     # > 4096 -> tile kernel, last tile partial
    np.testing.assert_allclose(g.predict(y, Xs, return_cov=False), o.predict(y, Xs), rtol=1e-9, atol=1e-9)


def test_empty_batch_and_protocol_errors():
    import torch
    from alabi_amd import HipGP, _lib
    X, y, h = make_problem(40, 2, 1)
    g = HipGP(2, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    with pytest.raises(RuntimeError):
        g.predict(y, X[:2], return_cov=False)                         # george: "You need to compute the model first"
    g.compute(X)
    assert g.predict(y, np.zeros((0, 2)), return_cov=False).shape == (0,)
    mu, var = g.predict(y, np.zeros((0, 2)), return_var=True)
    assert mu.shape == (0,) and var.shape == (0,)
    with pytest.raises(ValueError):
        g.predict(y[:-1], X[:2], return_cov=False)                    # y of the wrong length
    with pytest.raises(ValueError):
        g.predict(y, np.zeros((3, 5)), return_cov=False)              # wrong number of columns
    with pytest.raises(NotImplementedError):
        g.predict(y, X[:2])                                           # return_cov=True is off alabi's path
    with pytest.raises(ValueError):
        g.set_parameter_vector([0.0, 1.0])
    # raw C ABI: status codes, no exceptions across the boundary
    lib = _lib.lib()
    h_ = C.c_void_p()
    assert lib.alabi_gp_create(0, 2, C.byref(h_)) == _lib.BAD_ARG
    assert lib.alabi_gp_create(10, 65, C.byref(h_)) == _lib.BAD_ARG
    assert lib.alabi_gp_create(10, 2, C.byref(h_)) == _lib.OK
    xs = torch.zeros((4, 2), dtype=torch.float64, device="cuda"); mu_d = torch.zeros(4, dtype=torch.float64, device="cuda")
    assert lib.alabi_gp_predict(h_, _lib.ptr(xs), 4, _lib.ptr(mu_d), None, None) == _lib.NOT_COMPUTED
    assert lib.alabi_gp_set_y(h_, _lib.ptr(mu_d), None) == _lib.NOT_COMPUTED
    big = torch.zeros((11 * 64, 2), dtype=torch.float64, device="cuda")
    assert lib.alabi_gp_compute(h_, _lib.ptr(big), 11 * 64, None) == _lib.BAD_ARG      # beyond capacity
    assert lib.alabi_gp_set_hyper(h_, float("nan"), -12.0, 0.0, _lib.host_doubles([0.0, 0.0])) == _lib.BAD_ARG
    e = C.c_void_p()
    assert lib.alabi_ens_create(h_, 1, 2, 1, _lib.host_doubles([0, 1, 0, 1]), 0, C.byref(e)) == _lib.BAD_ARG
    assert lib.alabi_ens_create(h_, 9000, 2, 1, _lib.host_doubles([0, 1, 0, 1]), 0, C.byref(e)) == _lib.BAD_ARG
    assert lib.alabi_gp_destroy(h_) == _lib.OK
    assert lib.alabi_gp_destroy(None) == _lib.OK


def test_nan_training_input_is_reported_not_propagated():
    from alabi_amd import HipGP
    X, y, h = make_problem(70, 2, 2)
    X[13, 1] = np.nan
    g = HipGP(2, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    assert g.compute(X, quiet=True) is False
    with pytest.raises(np.linalg.LinAlgError):
        g.compute(X)


def test_scan_with_no_valid_candidate():
    from alabi_amd.utility import utility_scan
    X, y, h = make_problem(80, 2, 3)
    g, o = _pair(X, y, h)
    bounds = np.array([[-3.0, 3.0]] * 2)
    outside = np.random.RandomState(1).uniform(3.5, 9.0, (500, 2))
    best, val, idx = utility_scan(g, y, outside, bounds, "bape")
    assert idx == -1 and np.isnan(val) and np.all(np.isnan(best))
    on_the_wall = np.array([[3.0, 0.0], [-3.0, 1.0], [0.0, 3.0]])      # the open box excludes its faces
    assert utility_scan(g, y, on_the_wall, bounds, "agp")[2] == -1


def test_sampler_small_ensembles_and_thinning():
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    X, y, h = make_problem(90, 2, 4)
    g, o = _pair(X, y, h)
    bounds = np.array([[-3.0, 3.0]] * 2)
    lnp = lambda q: np.where(np.all((q > -3) & (q < 3), axis=1), o.predict(y, np.clip(q, -2.999, 2.999)), -np.inf)  # noqa: E731
    for W in (4, 5):
        p0 = np.random.RandomState(W).uniform(-1, 1, (W, 2))
        s = EnsembleSampler(W, 2, g, y, bounds, seed=8)
        s.run_mcmc(p0, 30, thin_by=7)
        ref = so.run_ensemble(p0, 30, lnp, seed=8, thin_by=7)[0]
        assert s.get_chain().shape == (4, W, 2) and np.max(np.abs(s.get_chain() - ref)) < 1e-7
    with pytest.raises(RuntimeError):
        EnsembleSampler(3, 2, g, y, bounds)                            # fewer than 2*ndim walkers (emcee's check)
    s = EnsembleSampler(4, 2, g, y, bounds, seed=1)
    s.run_mcmc(np.random.RandomState(0).uniform(-1, 1, (4, 2)), 5, thin_by=10)   # nothing reaches the store
    with pytest.raises(AttributeError):
        s.get_chain()
    with pytest.raises(ValueError):
        s.run_mcmc(np.zeros((4, 2)), 5)                                # degenerate initial ensemble
    with pytest.raises(ValueError):
        s.run_mcmc(np.zeros((3, 2)), 5)                                # wrong shape
    # walkers that start outside the box have logp = -inf and move in once a proposal lands inside
    p0 = np.random.RandomState(3).uniform(-1, 1, (8, 2)); p0[0] = [5.0, 5.0]
    s = EnsembleSampler(8, 2, g, y, bounds, seed=2)
    st = s.run_mcmc(p0, 200)
    assert np.all(np.isfinite(st.log_prob)) and np.all(np.abs(st.coords) < 3.0)


def test_ensemble_with_zero_alpha_points_and_short_length_scales():
    """A training set whose targets equal the mean (alpha identically 0) and inputs many length scales from their centre: the
    squared-exponential half-step kernels (se_pair_terms: exp(q.x - |q|^2/2 - h)) must return exactly the mean inside the box --
    a point without weight contributes nothing, wherever it lies (round-3 advisor: with its real coordinates the padded exponent
    could overflow)."""
    from alabi_amd import EnsembleSampler, HipGP
    rng = np.random.RandomState(0)
    d, N, W = 3, 300, 16
    X = rng.uniform(-40.0, 40.0, (N, d))                     # ~ +-90 length scales
    y = np.full(N, 1.25)
    g = HipGP(d, 1.25, -8.0, 0.0, np.log(np.full(d, 0.2)))
    g.compute(X)
    bounds = np.array([[-40.0, 40.0]] * d)
    p0 = rng.uniform(-35, 35, (W, d))
    for stream in ("1", "0"):
        os.environ["ALABI_ENS_STREAM"] = stream
        try:
            s = EnsembleSampler(W, d, g, y, bounds, seed=3)
            s.run_mcmc(p0, 40)
        finally:
            os.environ.pop("ALABI_ENS_STREAM", None)
        lp = s.get_log_prob()
        assert np.all(np.isfinite(lp)) and np.all(lp == 1.25), (stream, lp.min(), lp.max())
        assert s.acceptance_fraction.mean() > 0.2
    # half the targets off the mean: the live half decides, against the oracle
    from oracle.gp_oracle import OracleGP
    y2 = y.copy(); y2[::2] += rng.normal(0, 0.3, len(y2[::2]))
    o = OracleGP(d, 1.25, -8.0, 0.0, np.log(np.full(d, 0.2))).compute(X)
    s = EnsembleSampler(W, d, g, y2, bounds, seed=3)
    q = np.vstack([X[:W // 2] + 0.05, p0[:W // 2]])
    np.testing.assert_allclose(s.compute_log_prob(q).cpu().numpy(), o.predict(y2, q), rtol=1e-9, atol=1e-12)
