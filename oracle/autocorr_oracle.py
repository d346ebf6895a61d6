"""Integrated autocorrelation time, restated from emcee 3.x ``autocorr.py``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (emcee is not
installed); the reference reaches it through ``sampler.get_autocorr_time(tol=0)``
at alabi/mcmc_utils.py:45 and alabi/core.py:2387.
"""
from __future__ import annotations

import numpy as np

__all__ = ["function_1d", "auto_window", "integrated_time"]


def _next_pow_two(n):
    i = 1
    while i < n:
        i = i << 1
    return i


def function_1d(x):
    x = np.atleast_1d(x)
    n = _next_pow_two(len(x))
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[: len(x)].real
    acf /= acf[0]
    return acf


def auto_window(taus, c):
    m = np.arange(len(taus)) < c * taus
    if np.any(m):
        return int(np.argmin(m))
    return len(taus) - 1


def integrated_time(x, c=5):
    """x: [n_t, n_w, n_d] -> tau[n_d] (Sokal window, averaged ACF over walkers)."""
    x = np.atleast_1d(x)
    if x.ndim == 1:
        x = x[:, None, None]
    if x.ndim == 2:
        x = x[:, :, None]
    n_t, n_w, n_d = x.shape
    tau = np.empty(n_d)
    for d in range(n_d):
        f = np.zeros(n_t)
        for k in range(n_w):
            f += function_1d(x[:, k, d])
        f /= n_w
        taus = 2.0 * np.cumsum(f) - 1.0
        tau[d] = taus[auto_window(taus, c)]
    return tau
