"""Chain post-processing: integrated autocorrelation time and burn-in / thin estimates.

``estimate_burnin`` mirrors alabi/mcmc_utils.py:15-72 (iburn = int(2 max tau),
ithin = max(int(0.5 min tau), 1), tau from ``sampler.get_autocorr_time(tol=0)``).
``integrated_time`` restates emcee 3 ``autocorr.integrated_time`` (Sokal window, ACF averaged
over walkers).  A chain that lives on the GPU is transformed there by the library's own kernels
(``alabi_chain_autocorr``: four-step transforms in LDS, alabi_amd/csrc/chain_acf.hip) and only the
walker-averaged autocorrelation function [ndim, nsteps] comes back -- no FFT
library: rocFFT compiles kernels at run time for every new transform length (1-2 s each on MI355X), and
the length follows the number of steps of the run.  Host arrays use NumPy's real FFT.
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
    # Device chains: the library's transforms (any length up to 2^21 steps, no run-time compilation).  Host chains, and device chains
    # of at most ALABI_FFT_HOST_MAX elements (default 0: none), use NumPy's real FFT along a contiguous time axis.
    native = x.is_cuda and n_t <= (1 << 21) and x.numel() > int(os.environ.get("ALABI_FFT_HOST_MAX", 0)) \
        and os.environ.get("ALABI_ACF_NATIVE", "1") != "0"
    if native:
        from . import _lib
        xc = x.contiguous()
        acf = torch.empty((n_d, n_t), dtype=torch.float64, device=x.device)
        _lib.check(_lib.lib().alabi_chain_autocorr(_lib.ptr(xc), n_t, n_w, n_d, _lib.ptr(acf), _lib.current_stream()),
                   "alabi_chain_autocorr")
        taus = (2.0 * np.cumsum(acf.cpu().numpy(), axis=-1) - 1.0).T                # [n_t, n_d]
    elif (not x.is_cuda) or x.numel() <= int(os.environ.get("ALABI_FFT_HOST_MAX", 0)):
        xh = np.ascontiguousarray(np.moveaxis(x.cpu().numpy(), 0, -1))           # [n_w, n_d, n_t]
        xh = xh - xh.mean(axis=-1, keepdims=True)
        f = np.fft.rfft(xh, n=2 * n, axis=-1)
        acf = np.fft.irfft(f * np.conjugate(f), n=2 * n, axis=-1)[..., :n_t]
        acf = acf / acf[..., :1]
        fmean = acf.mean(axis=0)                                                 # average over walkers -> [n_d, n_t]
        taus = (2.0 * np.cumsum(fmean, axis=-1) - 1.0).T                         # [n_t, n_d]
    else:                                             # more than 2^21 steps on the device (or ALABI_ACF_NATIVE=0): torch.fft = rocFFT
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
