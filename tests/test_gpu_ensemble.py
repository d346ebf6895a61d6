"""GPU parity of the ensemble sampler (through the C ABI) against oracle/stretch_oracle.py."""
import ctypes as C

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
    torch.cuda.set_device(0)
    X, y, h = make_problem(500, 5, 31)
    g = HipGP(5, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(5, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    bounds = np.array([[-3.0, 3.0]] * 5)
    return torch, g, o, y, bounds


def _lnp(o, y, bounds):
    from oracle.stretch_oracle import box_lnprior_batch

    def f(q):
        lp = box_lnprior_batch(q, bounds)
        inside = np.isfinite(lp)
        out = np.full(len(q), -np.inf)
        if inside.any():
            out[inside] = o.predict(y, q[inside])
        return out
    return f


def test_device_draws_match_oracle_bit_for_bit(setup):
    torch, g, o, y, bounds = setup
    from alabi_amd import EnsembleSampler, _lib
    from oracle.stretch_oracle import draw_step_randoms
    seed = 0xDEADBEEFCAFE1234
    for W, E in ((10, 1), (64, 1), (257, 1), (1024, 1), (32, 3)):
        s = EnsembleSampler(W, 5, g, y, bounds, seed=seed, live_dangerously=True, n_ensembles=E)
        s._ensure_ens()
        WT = W * E
        for step in (0, 1, 12345678901):
            order = torch.empty(WT, dtype=torch.int32, device="cuda"); partner = torch.empty_like(order)
            cw = torch.empty_like(order)
            u_z = torch.empty(WT, dtype=torch.float64, device="cuda"); u_acc = torch.empty_like(u_z)
            zz = torch.empty_like(u_z)
            n0 = C.c_int(0)
            st = _lib.lib().alabi_ens_export_draws(s._ens, step, 2.0, _lib.ptr(order), C.byref(n0), _lib.ptr(u_z),
                                                   _lib.ptr(partner), _lib.ptr(u_acc), _lib.ptr(cw), _lib.ptr(zz),
                                                   _lib.current_stream())
            _lib.check(st, "export")
            torch.cuda.synchronize()
            for e in range(E):      # the device exports in LIST order with global ids; the oracle keys by walker id
                ro, rn0, ruz, rp, rua = draw_step_randoms(seed, step, W, id0=e * W)
                sl = slice(e * W, (e + 1) * W)
                assert n0.value == rn0
                assert np.array_equal(order.cpu().numpy()[sl], ro + e * W)   # integer index arithmetic: bit exact
                assert np.array_equal(partner.cpu().numpy()[sl], rp[ro])
                assert np.array_equal(u_z.cpu().numpy()[sl], ruz[ro])        # 53-bit uniforms: bit exact
                assert np.array_equal(u_acc.cpu().numpy()[sl], rua[ro])
                ref_cw = np.where(np.arange(W) < rn0, ro[rn0:][np.minimum(rp[ro], W - rn0 - 1)],
                                  ro[:rn0][np.minimum(rp[ro], rn0 - 1)])
                assert np.array_equal(cw.cpu().numpy()[sl], ref_cw + e * W)
                assert np.array_equal(zz.cpu().numpy()[sl], ((2.0 - 1.0) * ruz[ro] + 1.0) ** 2.0 / 2.0)


def test_step_with_injected_randoms(setup):
    """Same (order, u_z, partner, u_acc) -> same partner indices, proposals, accept mask."""
    torch, g, o, y, bounds = setup
    from alabi_amd import EnsembleSampler, _lib
    from oracle import stretch_oracle as so
    W, d = 48, 5
    rng = np.random.RandomState(4)
    s = EnsembleSampler(W, d, g, y, bounds, seed=1)
    coords = rng.uniform(-2.9, 2.9, (W, d))             # close to the walls: some proposals leave the box
    lnp = _lnp(o, y, bounds)
    logp_o = lnp(coords)
    c_dev = torch.as_tensor(coords, device="cuda").clone()
    lp_dev = s.compute_log_prob(c_dev)
    assert np.max(np.abs(lp_dev.cpu().numpy() - logp_o)) < 1e-8 * (1 + np.max(np.abs(logp_o)))
    nacc = torch.zeros(W, dtype=torch.int64, device="cuda")
    n_out = 0
    for it in range(30):
        rs = np.random.RandomState(100 + it)
        inds = np.arange(W) % 2; rs.shuffle(inds)
        ids = np.arange(W)
        order = np.concatenate([ids[inds == 0], ids[inds == 1]]).astype(np.int32); n0 = int((inds == 0).sum())
        u_z = rs.rand(W); u_acc = rs.rand(W); partner = rs.randint(W // 2, size=W).astype(np.int32)
        c_o, l_o, a_o = so.stretch_step_arrays(coords, logp_o, order, n0, u_z, partner, u_acc, lnp)
        before = c_dev.cpu().numpy().copy()
        # keep the device copies alive until the kernels have run (the call is asynchronous)
        d_order, d_uz, d_partner, d_uacc = (torch.as_tensor(a, device="cuda") for a in (order, u_z, partner, u_acc))
        st = _lib.lib().alabi_ens_step_with_randoms(s._ens, _lib.ptr(c_dev), _lib.ptr(lp_dev), _lib.ptr(d_order), n0,
                                                    _lib.ptr(d_uz), _lib.ptr(d_partner), _lib.ptr(d_uacc), 2.0,
                                                    _lib.ptr(nacc), _lib.current_stream())
        _lib.check(st, "step_with_randoms")
        torch.cuda.synchronize()
        c_g = c_dev.cpu().numpy(); l_g = lp_dev.cpu().numpy()
        a_g = np.any(c_g != before, axis=1)
        # knife-edge accepts (|lnpdiff - ln u| below the fp64 agreement of the two lnprobs) may differ; none expected
        assert np.array_equal(a_g, a_o), f"accept mask differs at iteration {it}"
        assert np.array_equal(c_g, c_o)                  # proposals are bit-identical (no FMA contraction)
        assert np.max(np.abs(l_g - l_o)) < 1e-8 * (1 + np.max(np.abs(l_o[np.isfinite(l_o)])))
        n_out += int(np.sum(~a_o))
        coords, logp_o = c_o, l_o
        lp_dev.copy_(torch.as_tensor(l_o, device="cuda"))   # keep the two states identical for the next step
    assert int(nacc.sum()) > 0 and n_out > 0


@pytest.mark.parametrize("W,nsteps,thin", [(32, 300, 1), (64, 257, 3), (33, 64, 1)])
def test_production_run_matches_oracle_chain(setup, W, nsteps, thin):
    """Counter-based draws + kernel sequence (graph replay + eager tail) == oracle run, step for step."""
    torch, g, o, y, bounds = setup
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    rng = np.random.RandomState(W)
    p0 = rng.uniform(-2, 2, (W, 5))
    s = EnsembleSampler(W, 5, g, y, bounds, seed=77)
    s.run_mcmc(p0, nsteps, thin_by=thin)
    chain = s.get_chain()
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(p0, nsteps, _lnp(o, y, bounds), seed=77, thin_by=thin)
    assert chain.shape == chain_o.shape
    assert np.max(np.abs(chain - chain_o)) < 1e-7
    assert np.max(np.abs(s.get_log_prob() - lp_o)) < 1e-7
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)
    # continuing the run continues the counter (emcee: run_mcmc(None, n) resumes)
    s.run_mcmc(None, 10 * thin, thin_by=thin)
    chain_o2, _, _, _, _ = so.run_ensemble(chain_o[-1] if nsteps % thin == 0 else None, 10 * thin, _lnp(o, y, bounds),
                                           seed=77, thin_by=thin, step0=nsteps) if nsteps % thin == 0 else (None,) * 5
    if chain_o2 is not None:
        assert np.max(np.abs(s.get_chain()[-10:] - chain_o2)) < 1e-7


def test_graph_and_eager_paths_agree(setup, monkeypatch):
    torch, g, o, y, bounds = setup
    from alabi_amd import EnsembleSampler
    p0 = np.random.RandomState(2).uniform(-2, 2, (40, 5))
    monkeypatch.setenv("ALABI_ENS_GRAPH_STEPS", "64")
    a = EnsembleSampler(40, 5, g, y, bounds, seed=5); a.run_mcmc(p0, 200)
    monkeypatch.setenv("ALABI_ENS_GRAPH", "0")
    b = EnsembleSampler(40, 5, g, y, bounds, seed=5); b.run_mcmc(p0, 200)
    assert np.array_equal(a.get_chain(), b.get_chain())
    assert np.array_equal(a.acceptance_fraction, b.acceptance_fraction)


def test_sampler_statistics_on_gaussian_surrogate(setup):
    """Size-independent property: the chain samples exp(surrogate) -- compare moments with the oracle chain."""
    torch, g, o, y, bounds = setup
    from alabi_amd import EnsembleSampler
    W = 64
    p0 = np.random.RandomState(8).uniform(-1, 1, (W, 5))
    s = EnsembleSampler(W, 5, g, y, bounds, seed=3)
    s.run_mcmc(p0, 4000)
    flat = s.get_chain(discard=500, flat=True)
    assert np.all(np.abs(flat.mean(axis=0)) < 0.5)
    assert 0.15 < s.acceptance_fraction.mean() < 0.9
    tau = s.get_autocorr_time(tol=0)
    assert tau.shape == (5,) and np.all(np.isfinite(tau))
    assert np.all(flat > -3.0) and np.all(flat < 3.0)      # the box prior is never violated


def test_independent_ensembles_share_launches(setup):
    """n_ensembles=E: rows [eW,(e+1)W) evolve exactly like a stand-alone ensemble whose walker ids start at eW."""
    torch, g, o, y, bounds = setup
    from alabi_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    W, E, nsteps = 24, 4, 150
    p0 = np.random.RandomState(21).uniform(-2, 2, (W * E, 5))
    s = EnsembleSampler(W, 5, g, y, bounds, seed=1234, n_ensembles=E)
    s.run_mcmc(p0, nsteps)
    chain = s.get_chain()
    assert chain.shape == (nsteps, W * E, 5)
    lnp = _lnp(o, y, bounds)
    for e in range(E):
        ref, _, nacc, _, _ = so.run_ensemble(p0[e * W:(e + 1) * W], nsteps, lnp, seed=1234, id0=e * W)
        assert np.max(np.abs(chain[:, e * W:(e + 1) * W] - ref)) < 1e-7
        assert np.array_equal(s._naccept.cpu().numpy()[e * W:(e + 1) * W], nacc)
    # ensembles are independent: different draws, different chains
    assert not np.allclose(chain[-1, :W], chain[-1, W:2 * W])


def test_stream_kernel_equals_launch_per_half_step(setup, monkeypatch):
    """The persistent dataflow kernel (default when the ensemble fits one workgroup per CU) and the
    launch-per-half-step path produce bit-identical chains, acceptance counts and final states."""
    torch, g, o, y, bounds = setup
    from alabi_amd import EnsembleSampler
    # the last two cases have more list positions than CUs per ensemble: each workgroup strides over several
    for W, E, nsteps, thin in ((40, 1, 300, 1), (33, 1, 130, 2), (24, 3, 90, 1), (600, 1, 40, 1), (200, 3, 40, 1)):
        p0 = np.random.RandomState(W).uniform(-2, 2, (W * E, 5))
        monkeypatch.setenv("ALABI_ENS_STREAM", "1")
        a = EnsembleSampler(W, 5, g, y, bounds, seed=5, n_ensembles=E); sa = a.run_mcmc(p0, nsteps, thin_by=thin)
        assert getattr(a, "stream_fallbacks", 0) == 0
        monkeypatch.setenv("ALABI_ENS_STREAM", "0")
        b = EnsembleSampler(W, 5, g, y, bounds, seed=5, n_ensembles=E); sb = b.run_mcmc(p0, nsteps, thin_by=thin)
        assert np.array_equal(a.get_chain(), b.get_chain())
        assert np.array_equal(a.get_log_prob(), b.get_log_prob())
        assert np.array_equal(a.acceptance_fraction, b.acceptance_fraction)
        assert np.array_equal(sa.coords, sb.coords) and np.array_equal(sa.log_prob, sb.log_prob)
        # and a second run continues identically
        a.run_mcmc(None, 50); b.run_mcmc(None, 50)
        assert np.array_equal(a.get_chain(), b.get_chain())


def test_persistent_kernel_equals_the_launch_per_half_step_path(setup, monkeypatch):
    """ens_stream_kernel against the launch-per-half-step kernels at several ensemble shapes (one or several ensembles per
    launch, odd walker counts, thinning): bit-identical chains, log-probabilities and acceptance counters."""
    torch, g, o, y, bounds = setup
    from alabi_amd import EnsembleSampler
    for W, E, nsteps, thin in ((256, 1, 1500, 1), (64, 4, 400, 3), (37, 2, 257, 1), (10, 1, 1100, 1)):
        p0 = np.random.RandomState(W + E).uniform(-2.5, 2.5, (W * E, 5))
        runs = {}
        for tag, stream in (("stream", "1"), ("half", "0")):
            monkeypatch.setenv("ALABI_ENS_STREAM", stream)
            s = EnsembleSampler(W, 5, g, y, bounds, seed=17, n_ensembles=E)
            s.run_mcmc(p0, nsteps, thin_by=thin)
            assert getattr(s, "stream_fallbacks", 0) == 0
            assert s.last_stream_kernel == {"stream": "ens_stream_kernel", "half": None}[tag]
            runs[tag] = (s.get_chain(), s.get_log_prob(), s._naccept.cpu().numpy().copy())
        assert np.array_equal(runs["stream"][0], runs["half"][0]), (W, E)
        assert np.array_equal(runs["stream"][1], runs["half"][1])
        assert np.array_equal(runs["stream"][2], runs["half"][2])


def test_persistent_kernel_timeout_falls_back(monkeypatch):
    """A hand-off that does not arrive within the spin limit makes every workgroup of the persistent kernel leave; the
    sampler restores the state and repeats the run on the launch-per-half-step path with the identical chain."""
    from alabi_amd import EnsembleSampler, HipGP
    X, y, h = make_problem(150, 3, 11, log_wn=-9.0)
    g = HipGP(3, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    bounds = np.array([[-3.0, 3.0]] * 3)
    p0 = np.random.RandomState(1).uniform(-2, 2, (16, 3))
    monkeypatch.setenv("ALABI_ENS_STREAM", "0")
    ref = EnsembleSampler(16, 3, g, y, bounds, seed=4); ref.run_mcmc(p0, 60)
    monkeypatch.setenv("ALABI_ENS_STREAM", "1")
    monkeypatch.setenv("ALABI_ENS_SPIN_LIMIT", "1")          # the second half step can never be ready after one poll
    s = EnsembleSampler(16, 3, g, y, bounds, seed=4); s.run_mcmc(p0, 60)
    assert getattr(s, "stream_fallbacks", 0) == 1 and s.last_path == "launch-per-half-step"
    np.testing.assert_array_equal(s.get_chain(), ref.get_chain())
    np.testing.assert_array_equal(s.get_log_prob(), ref.get_log_prob())
    np.testing.assert_array_equal(s.acceptance_fraction, ref.acceptance_fraction)
    monkeypatch.delenv("ALABI_ENS_SPIN_LIMIT")
    s.run_mcmc(None, 40); ref.run_mcmc(None, 40)              # the sampler stays on the fallback path and keeps going
    np.testing.assert_array_equal(s.get_chain(), ref.get_chain())


@pytest.mark.parametrize("W,E", [(40, 16), (36, 32), (700, 1)])
def test_multi_proposal_half_step_bit_identical(W, E, monkeypatch):
    """More proposals than CUs per half step: ens_half_multi_kernel (2 or 4 proposals per workgroup sharing one pass over the
    training set) must reproduce ens_half_kernel bit for bit (ragged last workgroup included)."""
    from alabi_amd import EnsembleSampler, HipGP
    X, y, h = make_problem(260, 4, 21, log_wn=-9.0)
    g = HipGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    bounds = np.array([[-3.0, 3.0]] * 4)
    p0 = np.random.RandomState(5).uniform(-2, 2, (W * E, 4))
    monkeypatch.setenv("ALABI_ENS_STREAM", "0")
    out = {}
    for multi in ("0", "1"):
        monkeypatch.setenv("ALABI_ENS_MULTI", multi)
        s = EnsembleSampler(W, 4, g, y, bounds, seed=8, n_ensembles=E)
        s.run_mcmc(p0, 40)
        out[multi] = (s.get_chain(), s.get_log_prob(), s.acceptance_fraction)
    for a, b in zip(out["0"], out["1"]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("stream", ["1", "0"])
def test_normal_prior_and_logp_affine_against_oracle(stream, monkeypatch):
    """lnprior_normal on two of four coordinates plus an affine log-probability map: both ensemble paths (and the multi-proposal
    kernel through E = 12) must follow the oracle run step for step."""
    from scipy.stats import norm
    from alabi_amd import EnsembleSampler, HipGP
    from oracle import stretch_oracle as so
    from oracle.gp_oracle import OracleGP
    X, y, h = make_problem(220, 4, 77, log_wn=-9.0)
    g = HipGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(X)
    o = OracleGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
    bounds = np.array([[-3.0, 3.0]] * 4)
    pm = np.array([0.4, np.nan, -0.8, np.nan]); ps = np.array([0.7, np.nan, 1.3, np.nan])
    c, e0 = 2.5, -3.0

    def lnp(q):
        inside = np.all((q > -3) & (q < 3), axis=1)
        out = np.full(len(q), -np.inf)
        if inside.any():
            v = c * o.predict(y, q[inside]) + e0
            for k in (0, 2):
                v = v + norm.logpdf(q[inside][:, k], pm[k], ps[k])
            out[inside] = v
        return out

    monkeypatch.setenv("ALABI_ENS_STREAM", stream)
    W = 24
    p0 = np.random.RandomState(3).uniform(-2, 2, (W, 4))
    ref = so.run_ensemble(p0, 150, lnp, seed=13)
    s = EnsembleSampler(W, 4, g, y, bounds, seed=13, logp_affine=(c, e0), normal_prior=(pm, ps))
    s.run_mcmc(p0, 150)
    assert np.max(np.abs(s.get_chain() - ref[0])) < 1e-7
    assert np.max(np.abs(s.get_log_prob() - ref[1])) < 1e-7 * (np.max(np.abs(ref[1])) + 1)
    np.testing.assert_array_equal(np.rint(s.acceptance_fraction * 150).astype(np.int64), ref[2])
    if stream == "0":       # many small ensembles -> several proposals per workgroup
        E = 12
        p0e = np.random.RandomState(4).uniform(-2, 2, (W * E, 4))
        se = EnsembleSampler(W, 4, g, y, bounds, seed=13, n_ensembles=E, logp_affine=(c, e0), normal_prior=(pm, ps))
        se.run_mcmc(p0e, 40)
        for k in (0, E - 1):
            refk = so.run_ensemble(p0e[k * W:(k + 1) * W], 40, lnp, seed=13, id0=k * W)
            assert np.max(np.abs(se.get_chain()[:, k * W:(k + 1) * W] - refk[0])) < 1e-7


@pytest.mark.parametrize("stream", ["1", "0"])
def test_half_step_kernels_inputs_far_from_origin(monkeypatch, stream):
    """The squared-exponential half-step kernels (ens_stream_kernel / ens_half_kernel) form q.x - |x|^2/2 - |q|^2/2 instead of
    -|x - q|^2/2 (se_pair_terms: one fma per point and coordinate, ln|alpha| folded into the resident norm); all coordinates are
    taken relative to the mean of the training inputs, so inputs thousands of length scales from the origin keep their digits:
    chain, log-probabilities and acceptance counts against the oracle (alabi/core.py:2319-2325 -> emcee stretch move)."""
    from alabi_amd import EnsembleSampler, HipGP
    from oracle.gp_oracle import OracleGP
    from oracle import stretch_oracle as so
    from oracle.stretch_oracle import box_lnprior_batch
    monkeypatch.setenv("ALABI_ENS_STREAM", stream)
    monkeypatch.setenv("ALABI_ENS_GROUP", "0")
    X, y, h = make_problem(800, 4, 77, log_wn=-10.0, ell2=4.0)
    off = np.array([4000.0, -2500.0, 1000.0, 8000.0])
    Xo = X + off
    g = HipGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(Xo)
    o = OracleGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(Xo)
    bounds = np.stack([off - 3.0, off + 3.0], axis=1)
    p0 = off + np.random.RandomState(6).uniform(-2, 2, (32, 4))

    def lnp(q):
        lp = box_lnprior_batch(q, bounds)
        out = np.full(len(q), -np.inf)
        inside = np.isfinite(lp)
        if inside.any():
            out[inside] = o.predict(y, q[inside])
        return out
    s = EnsembleSampler(32, 4, g, y, bounds, seed=2)
    s.run_mcmc(p0, 80)
    assert s.last_path == ("stream" if stream == "1" else "launch-per-half-step")
    chain_o, lp_o, nacc_o, _, _ = so.run_ensemble(p0, 80, lnp, seed=2)
    assert np.max(np.abs(s.get_chain() - chain_o)) <= 1e-7 * 8000
    assert np.max(np.abs(s.get_log_prob() - lp_o) / (np.abs(lp_o) + 1)) <= 1e-8
    assert np.array_equal(s._naccept.cpu().numpy(), nacc_o)
