"""Chain post-processing: integrated autocorrelation time and burn-in / thin estimates.

``estimate_burnin`` mirrors alabi/mcmc_utils.py:15-72 (iburn = int(2 max tau),
ithin = max(int(0.5 min tau), 1), tau from ``sampler.get_autocorr_time(tol=0)``).
``integrated_time`` restates emcee 3 ``autocorr.integrated_time`` (Sokal window, ACF averaged
over walkers); chains above 64 MB are transformed on the device that holds them, so that only
``ndim`` numbers come back; shorter ones on the host, where the FFT has no
run-time compilation cost.
"""
from __future__ import annotations

import os

import numpy as np
import torch

__all__ = ["estimate_burnin", "integrated_time", "AutocorrError"]


class AutocorrError(Exception):
    def __init__(self, tau, *args, **kwargs):
        self.tau = tau
        super().__init__(*args, **kwargs)


def _next_pow_two(n):
    i = 1
    while i < n:
        i = i << 1
    return i


def _auto_window(taus, c):
    m = np.arange(len(taus)) < c * taus
    if np.any(m):
        return int(np.argmin(m))
    return len(taus) - 1


def integrated_time(x, c=5, tol=50, quiet=False, has_walkers=True):
    """tau[ndim] for a chain x[n_t, n_w, n_d] (torch tensor on any device, or array-like)."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.atleast_1d(np.asarray(x, dtype=np.float64)))
    x = x.to(torch.float64)
    if x.dim() == 1:
        x = x[:, None, None]
    if x.dim() == 2:
        x = x[:, :, None]
    n_t, n_w, n_d = x.shape
    n = _next_pow_two(n_t)
    # The first FFT of a process on the GPU pays rocFFT's run-time kernel compilation (2.9 s on MI355X, then 1 ms per call);
    # chains of up to 64 MB are therefore transformed on the host with NumPy's real FFT along a contiguous time axis
    # (5 MB: about 10 ms), longer ones stay on the device (ALABI_FFT_HOST_MAX, in elements, moves the switch).
    if (not x.is_cuda) or x.numel() <= int(os.environ.get("ALABI_FFT_HOST_MAX", 8_000_000)):
        xh = np.ascontiguousarray(np.moveaxis(x.cpu().numpy(), 0, -1))           # [n_w, n_d, n_t]
        xh = xh - xh.mean(axis=-1, keepdims=True)
        f = np.fft.rfft(xh, n=2 * n, axis=-1)
        acf = np.fft.irfft(f * np.conjugate(f), n=2 * n, axis=-1)[..., :n_t]
        acf = acf / acf[..., :1]
        fmean = acf.mean(axis=0)                                                 # average over walkers -> [n_d, n_t]
        taus = (2.0 * np.cumsum(fmean, axis=-1) - 1.0).T                         # [n_t, n_d]
    else:
        xc = x - x.mean(dim=0, keepdim=True)
        f = torch.fft.fft(xc, n=2 * n, dim=0)
        acf = torch.fft.ifft(f * torch.conj(f), dim=0)[:n_t].real
        acf = acf / acf[0:1]
        fmean = acf.mean(dim=1)                       # average over walkers -> [n_t, n_d]
        taus = (2.0 * torch.cumsum(fmean, dim=0) - 1.0).cpu().numpy()
    tau_est = np.empty(n_d)
    for d in range(n_d):
        tau_est[d] = taus[_auto_window(taus[:, d], c), d]
    flag = tol * tau_est > n_t
    if np.any(flag) and tol > 0:
        msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} parameter(s). "
               "Use this estimate with caution and run a longer chain!\nN/{0} = {2:.0f};\ntau: {3}"
               ).format(tol, int(np.sum(flag)), n_t / tol, tau_est)
        if not quiet:
            raise AutocorrError(tau_est, msg)
    return tau_est


def estimate_burnin(sampler, est_burnin=True, thin_chains=True, verbose=False):
    tau = sampler.get_autocorr_time(tol=0)
    if np.any(~np.isfinite(tau)):
        tau = tau[np.isfinite(np.array(tau))]
        if len(tau) < 1:
            if verbose:
                print("Failed to compute integrated autocorrelation length, tau.")
                print("Setting tau = 1")
            tau = 1
    iburn = int(2.0 * np.max(tau)) if est_burnin else 0
    ithin = np.max((int(0.5 * np.min(tau)), 1)) if thin_chains else 1
    if verbose:
        print("burn-in estimate: %d" % iburn)
        print("thin estimate: %d\n" % ithin)
    return iburn, ithin
