import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
    return dict(np.load(path))


@pytest.fixture(scope="session")
def golden_grad():
    """Acquisition-gradient vectors from the reference's utility.py (tests/golden/make_golden_grad.py)."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_grad_vectors.npz")))


def make_problem(N, d, seed, log_wn=-12.0, ell2=None):
    """Synthetic training set on a smooth target (SURVEY.md section 8(d) recipe, scaled down)."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(-3.0, 3.0, (N, d))
    A = rng.randn(d, d)
    prec = A @ A.T / d + 0.5 * np.eye(d)
    y = -0.5 * np.einsum("ni,ij,nj->n", X, prec, X) / d
    log_M = np.log(np.full(d, 2.0 if ell2 is None else ell2) * rng.uniform(0.7, 1.4, d))
    hyper = dict(mean=float(np.median(y)), log_white_noise=log_wn, log_amp=float(np.log(np.var(y))), log_M=log_M)
    return X, y, hyper
