"""Chain post-processing on the device: the DEVICE branch of integrated_time -- the library's own
four-step transforms in LDS (alabi_chain_autocorr, alabi_amd/csrc/chain_acf.hip), which every chain that lives on the GPU takes --
against oracle/autocorr_oracle.py and against NumPy's FFT on the walker-averaged autocorrelation function itself, and the burn-in /
thinning rule of alabi/mcmc_utils.py:45-72 (iburn = int(2 max tau), ithin = max(int(min tau / 2), 1)) on top of it.
Reference call sites: alabi/mcmc_utils.py:45 (sampler.get_autocorr_time(tol=0)), alabi/core.py:2335-2345."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ar1(n, w, d, phi, seed):
    from scipy.signal import lfilter
    rng = np.random.RandomState(seed)
    e = rng.randn(n, w, d)
    return lfilter([1.0], [1.0, -phi], e, axis=0)


def test_device_integrated_time_vs_oracle():
    import torch
    from alabi_amd.mcmc_utils import integrated_time
    from oracle.autocorr_oracle import integrated_time as oracle_tau
    assert torch.cuda.is_available()
    n, w, d = 70000, 64, 2                                 # 8.96e6 elements: above the host / device switch
    x = np.stack([_ar1(n, w, 1, 0.9, 1)[..., 0], _ar1(n, w, 1, 0.6, 2)[..., 0]], axis=-1)
    xd = torch.as_tensor(x, device="cuda")
    assert xd.numel() > 8_000_000
    tau_dev = integrated_time(xd, tol=0)                   # alabi_chain_autocorr on the device
    tau_ora = oracle_tau(x)
    np.testing.assert_allclose(tau_dev, tau_ora, rtol=1e-8)
    assert abs(tau_ora[0] - 19.0) < 2.0 and abs(tau_ora[1] - 4.0) < 0.5      # (1 + phi) / (1 - phi)
    # the host branch (what short chains take) returns the same numbers
    tau_host = integrated_time(torch.as_tensor(x), tol=0)
    np.testing.assert_allclose(tau_host, tau_ora, rtol=1e-10)


def test_sampler_burnin_and_thinning_on_a_device_chain(golden):
    """A real ensemble chain that stays on the device (8.3e6 elements): get_autocorr_time(tol=0) through the device
    branch equals the oracle on the host copy, and estimate_burnin applies the reference's rule to it."""
    import torch
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.mcmc_utils import estimate_burnin
    from alabi_amd.workloads import make_config
    from oracle.autocorr_oracle import integrated_time as oracle_tau
    cfg = make_config("C2")
    h = cfg["hyper"]
    g = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); g.compute(cfg["X"])
    s = EnsembleSampler(cfg["W"], cfg["d"], g, cfg["y"], cfg["bounds"], seed=3)
    nsteps = 13000                                         # 13000 x 128 x 5 = 8.32e6 elements
    s.run_mcmc(cfg["p0"], nsteps)
    chain_dev = s.get_chain_device()
    assert chain_dev.is_cuda and chain_dev.numel() > 8_000_000
    tau = s.get_autocorr_time(tol=0)
    tau_o = oracle_tau(s.get_chain())
    np.testing.assert_allclose(tau, tau_o, rtol=1e-8)
    iburn, ithin = estimate_burnin(s)
    assert iburn == int(2.0 * np.max(tau_o)) and ithin == max(int(0.5 * np.min(tau_o)), 1)
    # the rule itself is pinned by the reference's own estimate_burnin (tests/golden/make_golden.py); here on the device taus
    for row, n, ib, it in zip(golden["burn_tau"], golden["burn_ntau"], golden["burn_iburn"], golden["burn_ithin"]):
        class Stub:
            def get_autocorr_time(self, tol=0, _t=row[:n]):
                return _t
        assert tuple(int(v) for v in estimate_burnin(Stub())) == (int(ib), int(it))


@pytest.mark.parametrize("n_t,n_w,n_d", [(1, 3, 2), (2, 2, 1), (7, 5, 3), (100, 33, 2), (1024, 8, 4), (1025, 40, 10), (4097, 130, 5),
                                         (50000, 64, 3), (300001, 6, 2), (2 ** 21, 1, 1)])
def test_native_autocorrelation_function_vs_numpy(n_t, n_w, n_d):
    """alabi_chain_autocorr (walker-averaged, normalised autocorrelation function per dimension) against NumPy's FFT for lengths
    below / at / above powers of two, series counts that are no multiple of the tile sizes, and the largest supported length."""
    import torch
    from alabi_amd import _lib
    rng = np.random.RandomState(n_t % 1000 + n_w)
    x = _ar1(n_t, n_w, n_d, 0.8, 5) + 3.0 + rng.randn(1, n_w, n_d)          # offsets: the mean removal matters
    xd = torch.as_tensor(x, device="cuda")
    out = torch.empty((n_d, n_t), dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().alabi_chain_autocorr(_lib.ptr(xd), n_t, n_w, n_d, _lib.ptr(out), _lib.current_stream()), "alabi_chain_autocorr")
    n = 1
    while n < n_t:
        n *= 2
    xh = np.moveaxis(x, 0, -1) - x.mean(axis=0)[..., None]                   # [n_w, n_d, n_t]
    f = np.fft.rfft(xh, n=2 * n, axis=-1)
    acf = np.fft.irfft(f * np.conjugate(f), n=2 * n, axis=-1)[..., :n_t]
    with np.errstate(all="ignore"):
        ref = (acf / acf[..., :1]).mean(axis=0)                              # [n_d, n_t]
    got = out.cpu().numpy()
    if n_t == 1:                                                             # x - mean = 0: 0 / 0 in emcee as well
        assert np.all(np.isnan(got)) and np.all(np.isnan(ref))
        return
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-11)
    assert np.all(np.abs(got[:, 0] - 1.0) <= 1e-14)


def test_native_and_rocfft_branches_agree_and_bad_arguments(monkeypatch):
    import torch
    from alabi_amd import _lib
    from alabi_amd.mcmc_utils import integrated_time
    x = torch.as_tensor(_ar1(30000, 16, 3, 0.7, 9), device="cuda")
    tau_native = integrated_time(x, tol=0)
    monkeypatch.setenv("ALABI_ACF_NATIVE", "0")
    tau_rocfft = integrated_time(x, tol=0)
    np.testing.assert_allclose(tau_native, tau_rocfft, rtol=1e-9)
    out = torch.empty((3, 8), dtype=torch.float64, device="cuda")
    assert _lib.lib().alabi_chain_autocorr(_lib.ptr(x), 2 ** 21 + 1, 1, 1, _lib.ptr(out), _lib.current_stream()) == _lib.BAD_ARG
    assert _lib.lib().alabi_chain_autocorr(_lib.ptr(x), 0, 1, 1, _lib.ptr(out), _lib.current_stream()) == _lib.BAD_ARG
