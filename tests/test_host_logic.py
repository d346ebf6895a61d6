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


def _cv_golden():
    import os
    from conftest import ROOT
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_cv_vectors.npz")))


def test_kfold_splits_are_sklearns_shuffled_kfold():
    """gp_utils.kfold_splits on RandomState(seed) = the validation sets sklearn's KFold(shuffle=True, random_state=None) drew
    inside the reference's worker under np.random.seed(seed) (gp_utils.py:538; folds recorded by tests/golden/make_golden_cv.py)."""
    from alabi_amd import gp_utils
    g = _cv_golden()
    n, k = len(g["cv_theta"]), int(g["cv_k"])
    for ci, seed in enumerate(g["cv_seeds"]):
        got = gp_utils.kfold_splits(n, k, np.random.RandomState(int(seed)))
        want = [v[v >= 0] for v in g[f"cv_val_{ci}"]]
        assert len(got) == k
        for a, b in zip(got, want):
            np.testing.assert_array_equal(np.sort(a), np.sort(b))


def test_fold_scoring_rules_vs_reference_worker():
    """gp_utils._score + _inverse_map on OracleGP predictions = the fold scores the reference's _evaluate_candidate_worker
    returned (gp_utils.py:603-619: mse, mae, -r2, weighted mse; y un-scaled through no_scaler, nlog_scaler and a MinMaxScaler),
    and weighted_mse_by_probability for every weighting method (gp_utils.py:449-508)."""
    from sklearn.preprocessing import MinMaxScaler
    from alabi_amd import gp_utils
    from alabi_amd import utility as ut
    from oracle.gp_oracle import OracleGP
    g = _cv_golden()
    theta = g["cv_theta"]
    d = theta.shape[1]
    for scaler in ("none", "nlog", "minmax"):
        _y = g[f"cv_y_{scaler}"]
        ys = {"none": ut.no_scaler, "nlog": ut.nlog_scaler, "minmax": MinMaxScaler().fit(g["cv_lnlike"].reshape(-1, 1))}[scaler]
        inv = gp_utils._inverse_map(ys)
        for ci, hp in enumerate(g["cv_cands"]):
            folds = [v[v >= 0] for v in g[f"cv_val_{ci}"]]
            for kf, val in enumerate(folds):
                train = np.setdiff1d(np.arange(len(theta)), val)
                o = OracleGP(d, hp[0], hp[1], hp[2], hp[3:]).compute(theta[train])
                pred = o.predict(_y[train], theta[val])
                for scoring in ("mse", "mae", "r2", "weighted_mse"):
                    got = gp_utils._score(inv(_y[val]), inv(pred), scoring)
                    np.testing.assert_allclose(got, g[f"cv_{scaler}_{scoring}_{ci}"][kf], rtol=1e-9)
    for method in ("exponential", "linear", "softmax", "rank"):
        for temp in (1.0, 2.5):
            got = gp_utils.weighted_mse_by_probability(g["wmse_true"], g["wmse_pred"], weight_method=method, temperature=temp)
            np.testing.assert_allclose(got, g[f"wmse_{method}_{temp}"], rtol=1e-13)
