"""CPU restatement of the scale -> predict -> un-scale composite alabi wraps around the GP.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates, with OracleGP in george's place,

* ``SurrogateModel.surrogate_log_likelihood`` (alabi/core.py:1446-1508): theta -> theta_scaler.transform -> gp.predict ->
  y_scaler.inverse_transform; a 1-D input returns scalars; with ``return_var`` the VARIANCE is pushed through
  ``inverse_transform`` as well (the reference's own line, core.py:1502 -- kept as is);
* ``CachedSurrogateLikelihood.__call__`` (alabi/core.py:53-122): the same mean; the variance times ``scale_[0] ** 2`` when
  the scaler has a ``scale_`` attribute, otherwise times the squared numerical derivative of ``inverse_transform`` between
  0 and 1e-6 (core.py:100-116).

The scalers are whatever object the caller passes (sklearn transformers, the reference's FunctionTransformer scalers): only
``transform`` / ``inverse_transform`` / ``scale_`` are touched, as in the reference.
"""
from __future__ import annotations

import numpy as np

__all__ = ["surrogate_log_likelihood", "cached_surrogate_call"]


def _shape(theta_xs):
    theta_xs = np.asarray(theta_xs)
    one = theta_xs.ndim == 1
    if one:
        theta_xs = theta_xs.reshape(1, -1)
    elif theta_xs.ndim != 2:
        raise ValueError(f"theta_xs must be 1D or 2D array, got {theta_xs.ndim}D")
    return theta_xs, one


def surrogate_log_likelihood(gp, _y, theta_scaler, y_scaler, theta_xs, return_var=False):
    """core.py:1446-1508 with ``gp`` an OracleGP computed on the scaled training inputs and ``_y`` the scaled targets."""
    theta_xs, one = _shape(theta_xs)
    _t = theta_scaler.transform(theta_xs)
    if not return_var:
        _yp = gp.predict(_y, _t, return_var=False, return_cov=False)
        yp = y_scaler.inverse_transform(np.asarray(_yp).reshape(-1, 1)).flatten()
        return yp[0] if one else yp
    _yp, _vp = gp.predict(_y, _t, return_var=True)
    yp = y_scaler.inverse_transform(np.asarray(_yp).reshape(-1, 1)).flatten()
    vp = y_scaler.inverse_transform(np.asarray(_vp).reshape(-1, 1)).flatten()      # core.py:1502
    return (yp[0], vp[0]) if one else (yp, vp)


def cached_surrogate_call(gp, _y, theta_scaler, y_scaler, ndim, theta_xs, return_var=False):
    """core.py:53-122."""
    theta_xs, one = _shape(theta_xs)
    _t = np.atleast_2d(theta_scaler.transform(theta_xs))
    if _t.shape[0] == 1 and _t.shape[1] != ndim:
        if _t.shape[1] == 1 and _t.shape[0] == ndim:
            _t = _t.T
        elif _t.size == ndim:
            _t = _t.reshape(1, -1)
    if not return_var:
        _yp = gp.predict(_y, _t, return_var=False, return_cov=False)
        yp = y_scaler.inverse_transform(np.asarray(_yp).reshape(-1, 1)).flatten()
        return yp[0] if one else yp
    _yp, _vp = gp.predict(_y, _t, return_var=True)
    yp = y_scaler.inverse_transform(np.asarray(_yp).reshape(-1, 1)).flatten()
    if getattr(y_scaler, "scale_", None) is not None:
        vp = _vp * y_scaler.scale_[0] ** 2
    else:
        eps = 1e-6
        tr = y_scaler.inverse_transform(np.array([[0.0], [eps]]))
        vp = _vp * ((tr[1] - tr[0]) / eps) ** 2
    return (yp[0], vp[0]) if one else (yp, vp)
