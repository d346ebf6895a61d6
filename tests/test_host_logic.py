"""Host-side helpers that need no GPU."""
import numpy as np


def test_inverse_map_affine_and_fallback():
    """gp_utils._inverse_map: an affine y scaler becomes a * y + b (same numbers as inverse_transform), a non-affine one
    (the reference's nlog_scaler, utility.py:64-70) keeps calling the scaler."""
    from sklearn.preprocessing import MinMaxScaler, StandardScaler
    from alabi_amd import gp_utils
    from alabi_amd.utility import nlog_scaler, no_scaler
    y = np.random.RandomState(0).normal(-40.0, 7.0, (200, 1))
    v = np.random.RandomState(1).normal(0.0, 1.0, 57)
    for sc in (StandardScaler().fit(y), MinMaxScaler().fit(y), no_scaler):
        inv = gp_utils._inverse_map(sc)
        ref = sc.inverse_transform(v.reshape(-1, 1)).flatten()
        np.testing.assert_allclose(inv(v), ref, rtol=1e-13, atol=1e-12)
        assert inv.__defaults__ is not None and len(inv.__defaults__) == 2          # the a, b of the affine form
    inv = gp_utils._inverse_map(nlog_scaler)
    np.testing.assert_array_equal(inv(v), nlog_scaler.inverse_transform(v.reshape(-1, 1)).flatten())
    assert not inv.__defaults__


def test_utility_value_and_grad_matches_finite_differences():
    """utility.utility_value_and_grad: the chain rule for bape / agp / jones against central differences in (mu, var)."""
    from alabi_amd import utility as ut
    rng = np.random.RandomState(3)
    d = 4
    for algo in ("bape", "agp", "jones"):
        for _ in range(20):
            mu, var = rng.normal(-3.0, 2.0), float(np.exp(rng.uniform(-6.0, 2.5)))
            dmu, dvar = rng.normal(size=d), rng.normal(size=d) * var
            u, g = ut.utility_value_and_grad(algo, mu, var, dmu, dvar, y_best=-2.0)
            h = 1e-6
            for k in range(d):          # move along coordinate k: (mu, var) change by (dmu_k, dvar_k) per unit step
                up = ut.utility_value_and_grad(algo, mu + h * dmu[k], var + h * dvar[k], dmu, dvar, y_best=-2.0)[0]
                dn = ut.utility_value_and_grad(algo, mu - h * dmu[k], var - h * dvar[k], dmu, dvar, y_best=-2.0)[0]
                assert abs((up - dn) / (2 * h) - g[k]) <= 1e-5 * (abs(g[k]) + 1.0)
    assert ut.utility_value_and_grad("bape", 0.0, -1e-9, np.zeros(d), np.zeros(d))[0] == np.inf
