#!/usr/bin/env python3
"""Generate tests/golden/reference_grad_vectors.npz by EXECUTING the reference's own acquisition-gradient code.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden_grad.py

The reference's ``grad_gp_mean_prediction`` / ``grad_gp_var_prediction`` / ``grad_agp_utility`` /
``grad_bape_utility`` (alabi/utility.py:558-850) touch the GP only through ``gp.kernel.get_value``, ``gp._x``,
``gp._alpha``, ``gp._y``, ``gp.solver.get_inverse()`` and ``gp.predict``.  george is not installed here
(SURVEY.md section 8c), so the GP arithmetic behind those members is oracle.OracleGP (an adapter exposes the
member names); everything else -- the finite differencing, the contractions, the utility chain rule and the
bounds gate -- is the reference's code, loaded by file path exactly as in make_golden.py.  Stored: the
inputs (training set, hyper-parameters, query points, bounds) and the arrays the reference functions returned.
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden import REF, _load, _placeholders  # noqa: E402
from oracle.gp_oracle import OracleGP  # noqa: E402

OUT = os.path.join(HERE, "reference_grad_vectors.npz")


class _Adapter:
    """The members alabi/utility.py:558-850 touch, backed by an OracleGP."""

    def __init__(self, gp, y):
        self._gp = gp
        self._x = gp._x
        self._y = np.asarray(y, dtype=np.float64)
        self._alpha = gp._compute_alpha(y)
        self.kernel = types.SimpleNamespace(get_value=lambda a, b: gp._k(a, b))
        self.solver = types.SimpleNamespace(get_inverse=gp.get_inverse)

    def predict(self, y, t, return_var=False, return_cov=False):
        return self._gp.predict(y, t, return_var=return_var, return_cov=return_cov)


def main():
    warnings.filterwarnings("ignore")
    _placeholders()
    ut = _load("ref_utility", f"{REF}/utility.py")
    rng = np.random.RandomState(20261004)
    d, n, m = 3, 40, 24
    bounds = np.array([[0.0, 1.0], [-2.0, 2.0], [10.0, 11.0]])
    X = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * rng.rand(n, d)
    y = -0.5 * np.sum(((X - bounds.mean(axis=1)) / (0.3 * (bounds[:, 1] - bounds[:, 0]))) ** 2, axis=1)
    log_M = np.log(np.array([0.09, 1.3, 0.2]))
    hyper = dict(mean=float(np.median(y)), log_white_noise=-10.0, log_amp=float(np.log(np.var(y))), log_M=log_M)
    gp = OracleGP(d, **hyper)
    gp.compute(X)
    ad = _Adapter(gp, y)
    theta = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * rng.rand(m, d)
    theta[-3:, 0] = bounds[0, 1] + 0.1 * rng.rand(3)            # outside the box: gradient is inf[d]
    theta[-4] = X[5]                                             # on a training point
    out = dict(grad_X=X, grad_y=y, grad_bounds=bounds, grad_mean=np.array(hyper["mean"]),
               grad_log_wn=np.array(hyper["log_white_noise"]), grad_log_amp=np.array(hyper["log_amp"]), grad_log_M=log_M,
               grad_theta=theta,
               grad_dmu=np.array([ut.grad_gp_mean_prediction(t, ad) for t in theta]),
               grad_dvar=np.array([ut.grad_gp_var_prediction(t, ad) for t in theta]),
               grad_agp=np.array([ut.grad_agp_utility(t, ad, bounds) for t in theta]),
               grad_bape=np.array([np.asarray(ut.grad_bape_utility(t, ad, bounds)).flatten() for t in theta]))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
