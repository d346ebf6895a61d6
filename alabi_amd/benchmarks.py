"""Synthetic log-likelihoods used by the benchmarks and tests.

Same functions as alabi/benchmarks.py (values pinned by tests/golden/reference_vectors.npz),
plus the N-dimensional generalisations BASELINE.json's configs name:
``gaussian_shells_nd`` (benchmarks.py:100-116 is hard-coded 2-D) and ``gaussian_nd`` built on
the ``random_gaussian_covariance`` recipe (benchmarks.py:195-206).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.optimize import rosen
from scipy.stats import multivariate_normal

__all__ = ["test1d", "rosenbrock", "gaussian_shells", "eggbox", "gaussian_2d", "multimodal",
           "random_gaussian_covariance", "rosenbrock_nd", "gaussian_shells_nd", "gaussian_nd"]


def test1d_fn(theta):
    theta = np.asarray(theta)
    return -np.sin(3 * theta) - theta ** 2 + 0.7 * theta


test1d = {"fn": test1d_fn, "bounds": [(-2, 1)]}


def rosenbrock_fn(x):
    return -rosen(x) / 100.0


rosenbrock = {"fn": rosenbrock_fn, "bounds": [(-5, 5), (-5, 5)]}


def rosenbrock_nd(x, a, b):
    """Hybrid N-d Rosenbrock of Pagani et al. (2020), as parameterised in benchmarks.py:59-93."""
    x = np.asarray(x, dtype=np.float64)
    n1, n2 = b.shape
    ndim = (n1 - 1) * n2 + 1
    single = x.ndim == 1
    if single:
        x = x.reshape(1, -1)
    ll = -a * (x[:, 0] - 1) ** 2
    cnorm = np.sqrt(a / np.pi) * np.pi ** ndim
    ll = ll - ((x[:, 2:n1] - x[:, 1:n1 - 1] ** 2) ** 2 * b[:, 2:].sum(axis=0)).sum(axis=1)
    cnorm *= np.sqrt(np.prod(b[:, 2:]))
    ll = ll - np.log(cnorm)
    return ll[0] if single else ll


def _logcirc(theta, c, r=2.0, w=0.1):
    const = math.log(1.0 / math.sqrt(2.0 * math.pi * w ** 2))
    dist = np.sqrt(np.sum((theta - c) ** 2, axis=-1))
    return const - (dist - r) ** 2 / (2.0 * w ** 2)


def gaussian_shells_fn(theta):
    theta = np.asarray(theta).flatten()
    return np.logaddexp(_logcirc(theta, np.array([-3.5, 0.0])), _logcirc(theta, np.array([3.5, 0.0])))


gaussian_shells = {"fn": gaussian_shells_fn, "bounds": [(-6, 6), (-6, 6)]}


def gaussian_shells_nd(ndim, r=2.0, w=0.1, sep=3.5):
    """N-d double shell: centres (+-sep, 0, ..., 0), bounds [-6,6]^ndim."""
    c1 = np.zeros(ndim); c1[0] = -sep
    c2 = np.zeros(ndim); c2[0] = sep

    def fn(theta):
        theta = np.asarray(theta, dtype=np.float64)
        return np.logaddexp(_logcirc(theta, c1, r, w), _logcirc(theta, c2, r, w))

    return {"fn": fn, "bounds": [(-6, 6)] * ndim}


def eggbox_fn(x):
    x = np.asarray(x).flatten()
    tmax = 5.0 * np.pi
    t = 2.0 * tmax * x - tmax
    return -(2.0 + np.cos(t[0] / 2.0) * np.cos(t[1] / 2.0)) ** 5.0


eggbox = {"fn": eggbox_fn, "bounds": [(0, 1), (0, 1)]}


def multimodal_fn(x):
    x = np.asarray(x).flatten()
    return -(np.sin(x[0]) ** 10 + np.cos(10 + x[1] * x[0]) * np.cos(x[0]))


multimodal = {"fn": multimodal_fn, "bounds": [(0, 5), (0, 5)]}


def gaussian_2d_fn(theta):
    theta = np.asarray(theta).flatten()
    return multivariate_normal.logpdf(theta, mean=np.array([0.5, 0.5]), cov=np.array([[0.1, 0.0], [0.0, 0.1]]))


gaussian_2d = {"fn": gaussian_2d_fn, "bounds": [(0, 1), (0, 1)]}


def random_gaussian_covariance(n_dims):
    """Exponential(1) eigenvalues in a QR-orthogonal basis, drawn from NumPy's GLOBAL RNG in the
    same order as benchmarks.py:195-206 (so np.random.seed(s) reproduces the reference's matrix)."""
    eigenvals = np.random.exponential(scale=1.0, size=n_dims)
    q, _ = np.linalg.qr(np.random.randn(n_dims, n_dims))
    return q @ np.diag(eigenvals) @ q.T


def gaussian_nd(ndim, seed=2, half_width=3.0):
    """Zero-mean N-d Gaussian log-pdf with a random covariance."""
    state = np.random.get_state()
    np.random.seed(seed)
    cov = random_gaussian_covariance(ndim)
    np.random.set_state(state)
    prec = np.linalg.inv(cov)
    _, logdet = np.linalg.slogdet(cov)
    const = -0.5 * (ndim * np.log(2 * np.pi) + logdet)

    def fn(theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            return const - 0.5 * float(theta @ prec @ theta)
        return const - 0.5 * np.einsum("ni,ij,nj->n", theta, prec, theta)

    return {"fn": fn, "bounds": [(-half_width, half_width)] * ndim, "cov": cov}
