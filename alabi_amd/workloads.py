"""Synthetic workloads of BASELINE.json's configs: seeded inputs, fixed
hyper-parameters, built only from this repo's own code so they exist on the GPU box."""
from __future__ import annotations

import numpy as np

from . import benchmarks as bm

__all__ = ["make_config", "CONFIGS"]

CONFIGS = {
    # name: (description, d, N, W)
    "C1": ("2-D Rosenbrock, N_train=50, 64 walkers (reference CPU case)", 2, 50, 64),
    "C2": ("5-D Gaussian shells, N_train=500, 128 walkers", 5, 500, 128),
    "C3": ("10-D Gaussian, N_train=2000, 256 walkers (headline)", 10, 2000, 256),
    "C4": ("10-D Gaussian, N_train=5000, 1024 walkers", 10, 5000, 1024),
    "C5": ("20-D Gaussian ARD, N_train=10000, BAPE over 1e6 candidates", 20, 10000, 2048),
}


def make_config(name, N=None, W=None):
    desc, d, N0, W0 = CONFIGS[name]
    N = N0 if N is None else int(N)
    W = W0 if W is None else int(W)
    if name == "C1":
        fn, bounds = bm.rosenbrock_fn, np.array(bm.rosenbrock["bounds"], dtype=np.float64)
        X = np.random.RandomState(0).uniform(bounds[:, 0], bounds[:, 1], (N, d))
        y = np.array([fn(x) for x in X])
        log_M = np.log(np.array([4.0, 6.0]))
    elif name == "C2":
        sh = bm.gaussian_shells_nd(d)
        fn, bounds = sh["fn"], np.array(sh["bounds"], dtype=np.float64)
        X = np.random.RandomState(1).uniform(bounds[:, 0], bounds[:, 1], (N, d))
        y = fn(X)
        log_M = np.log(np.full(d, 3.0))
    else:
        seeds = {"C3": (2, 3), "C4": (2, 4), "C5": (2, 5)}[name]
        g = bm.gaussian_nd(d, seed=seeds[0])
        fn, bounds = g["fn"], np.array(g["bounds"], dtype=np.float64)
        rng = np.random.RandomState(seeds[1])
        X = rng.uniform(bounds[:, 0], bounds[:, 1], (N, d))
        y = fn(X)
        base = 30.0 if d == 10 else 60.0
        log_M = np.log(base * rng.uniform(0.8, 1.25, d)) if name != "C5" else np.log(base) + rng.uniform(-0.5, 0.5, d)
    hyper = dict(mean=float(np.median(y)), log_white_noise=-12.0, log_amp=float(np.log(np.var(y))), log_M=log_M)
    p0 = np.random.RandomState(1000 + d).uniform(bounds[:, 0] * 0.5, bounds[:, 1] * 0.5, (W, d))
    return dict(name=name, description=desc, d=d, N=N, W=W, X=X, y=y, hyper=hyper, bounds=bounds, p0=p0, fn=fn)
