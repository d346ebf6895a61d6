"""GPU parity of the batched fit (alabi_gp_batch_fit_predict, alabi_amd/csrc/gp_batch.hip): the (candidate, fold) jobs of the
k-fold cross-validation search, reference alabi/gp_utils.py:511-700 and alabi/core.py:1287-1305.

* factors bit-identical to the single-matrix path (alabi_gp_compute on the same rows with the same hyper-parameters),
  alpha / log-likelihood / held-out mean against that path and against the oracle;
* fold scores against vectors produced by the REFERENCE's own worker (tests/golden/make_golden_cv.py);
* a job that is not positive definite does not disturb its neighbours; the launch-per-step fallback gives the same results."""
import os

import numpy as np
import pytest

from conftest import ROOT, make_problem

pytestmark = pytest.mark.gpu


def _first_tiles(a, c):
    """(tile row, tile column, differing entries) of the first 64 x 64 tiles in which two factors differ, column by column"""
    dd = a != c
    nb = (len(a) + 63) // 64
    return [(i, j, int(dd[64 * i:64 * i + 64, 64 * j:64 * j + 64].sum())) for j in range(nb) for i in range(j, nb)
            if dd[64 * i:64 * i + 64, 64 * j:64 * j + 64].any()][:8]


def _jobs(n, d, seed, sizes, nval=37):
    rng = np.random.RandomState(seed)
    X, y, h = make_problem(n, d, seed, log_wn=-10.0, ell2=2.0 * d)
    hyper, train, val = [], [], []
    for N in sizes:
        perm = rng.permutation(n)
        train.append(np.sort(perm[:N])); val.append(np.sort(perm[N:N + nval]))
        hyper.append(np.r_[h["mean"] + 0.1 * rng.randn(), -10.0 + rng.uniform(-2, 1), h["log_amp"] + 0.3 * rng.randn(), 1.0,
                           h["log_M"] + 0.2 * rng.randn(d)])
    return X, y, np.array(hyper), train, val


@pytest.mark.parametrize("n,d,sizes", [
    (1800, 6, [1600, 1601, 1536, 1100, 1700, 1664, 1280, 1599, 1600, 1345, 1024]),   # 16..27 block columns: the single path is the queue too
    (700, 3, [64, 100, 130, 257, 600, 1, 320, 65]),                               # 1..10 block columns (single path: launch per step)
    (5300, 5, [5000, 4100, 3333]),                                                # 53..79 block columns: long groups, many 2 x 2 tasks
])
def test_batch_factors_bit_identical_and_results_match_single_path(n, d, sizes, monkeypatch):
    import torch
    from alabi_amd import HipGP
    from alabi_amd.gp_batch import HipGPBatch
    from oracle.gp_oracle import OracleGP
    monkeypatch.setenv("ALABI_CHOL_TASKS", "1")          # the single path through the task queue at every size it supports (>= 3 block columns)
    X, y, hyper, train, val = _jobs(n, d, 5 + d, sizes)
    dev = torch.device("cuda")
    Xd, yd = torch.as_tensor(X, device=dev), torch.as_tensor(y, device=dev)
    bt = HipGPBatch(d)
    ll, status, mu, off = bt.fit_predict(Xd, yd, hyper, train, val)
    assert bt.timeouts == 0 and np.all(status == 0)
    mu = mu.cpu().numpy()
    for b, N in enumerate(sizes):
        g = HipGP(d, hyper[b, 0], hyper[b, 1], hyper[b, 2], hyper[b, 4:])
        g.compute(X[train[b]])
        Ls = g.solver.get_factor().cpu().numpy()
        Lb = bt.get_factor(b, N).cpu().numpy()
        if N > 128:                                      # three block columns and more: both ran the task queue
            assert g.solver.factor_path == "queue"
            assert np.array_equal(Ls, Lb), (b, N, np.max(np.abs(Ls - Lb)), _first_tiles(Ls, Lb))
        else:                                            # launch per step (panel solves by substitution) against the queue (by the
            assert g.solver.factor_path == "steps"       # inverses of the 16 x 16 diagonal blocks on the matrix cores): to rounding
            assert np.max(np.abs(Ls - Lb)) <= 1e-11 * np.max(np.abs(Ls))
        ll_s = g.log_likelihood(y[train[b]])
        mu_s = g.predict(y[train[b]], X[val[b]], return_cov=False)
        assert abs(ll[b] - ll_s) <= 1e-9 * (abs(ll_s) + 1)
        scale = np.max(np.abs(mu_s - hyper[b, 0])) + 1e-300
        assert np.max(np.abs(mu[off[b]:off[b + 1]] - mu_s)) <= 1e-8 * scale
        if b < 3:                                        # the oracle on a few jobs (CPU Cholesky of 1600^2)
            o = OracleGP(d, hyper[b, 0], hyper[b, 1], hyper[b, 2], hyper[b, 4:]).compute(X[train[b]])
            assert abs(ll[b] - o.log_likelihood(y[train[b]])) <= 1e-8 * (abs(ll_s) + 1)
            mu_o = o.predict(y[train[b]], X[val[b]])
            assert np.max(np.abs(mu[off[b]:off[b + 1]] - mu_o)) <= 1e-7 * scale
            a_o = o._compute_alpha(y[train[b]])
            a_b = bt.get_alpha(b, N).cpu().numpy()
            # residual form: K alpha = y - m to rounding x condition
            K = o.get_matrix(X[train[b]])
            r = y[train[b]] - hyper[b, 0]
            assert np.max(np.abs(K @ a_b - r)) <= 1e-6 * np.max(np.abs(r)) and np.max(np.abs(K @ a_o - r)) <= 1e-6 * np.max(np.abs(r))
    bt.close()


def test_batch_not_positive_definite_job_is_isolated_and_fallback_agrees(monkeypatch):
    import torch
    from alabi_amd.gp_batch import HipGPBatch
    n, d = 900, 4
    X, y, hyper, train, val = _jobs(n, d, 17, [700, 640, 705, 512, 700])
    X = X.copy(); X[train[2][10]] = X[train[2][3]]       # job 2 holds a duplicated row ...
    hyper[2, 1] = -60.0                                  # ... and no nugget to speak of: not positive definite
    dev = torch.device("cuda")
    Xd, yd = torch.as_tensor(X, device=dev), torch.as_tensor(y, device=dev)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ALABI_BATCH_QUEUE", mode)
        bt = HipGPBatch(d)
        ll, status, mu, off = bt.fit_predict(Xd, yd, hyper, train, val)
        out[mode] = (ll.copy(), status.copy(), mu.cpu().numpy().copy())
        bt.close()
    ll, status, mu = out["1"]
    dup = [q for q in (2,) if status[q] != 0]
    ok = [q for q in range(5) if q not in dup]
    assert all(status[q] == 0 and np.isfinite(ll[q]) for q in ok)
    if dup:                                              # (rounding may let the duplicate through: then it is simply a valid fit)
        assert ll[2] == -np.inf and 1 <= status[2] <= 705 and np.all(np.isnan(mu[off[2]:off[3]]))
    assert np.array_equal(out["0"][1] != 0, status != 0)
    for q in ok:
        assert abs(out["0"][0][q] - ll[q]) <= 1e-9 * (abs(ll[q]) + 1)
        assert np.max(np.abs(out["0"][2][off[q]:off[q + 1]] - mu[off[q]:off[q + 1]])) <= 1e-8 * (np.max(np.abs(mu[off[q]:off[q + 1]])) + 1)


def test_batch_chunks_and_other_kernels(monkeypatch):
    """A workspace that holds two matrices at a time (five chunks) and the Matern / rational-quadratic families: the same numbers
    as one chunk / as the single-matrix path."""
    import torch
    from alabi_amd import HipGP
    from alabi_amd.gp_batch import HipGPBatch
    n, d = 500, 3
    X, y, hyper, train, val = _jobs(n, d, 23, [300, 320, 333, 384, 300, 310, 290, 256, 400])
    dev = torch.device("cuda")
    Xd, yd = torch.as_tensor(X, device=dev), torch.as_tensor(y, device=dev)
    for kernel in ("ExpSquaredKernel", "Matern32Kernel", "Matern52Kernel", "RationalQuadraticKernel"):
        big = HipGPBatch(d, kernel)
        small = HipGPBatch(d, kernel, workspace_bytes=2 * 384 * 384 * 8)
        r1 = big.fit_predict(Xd, yd, hyper, train, val)
        r2 = small.fit_predict(Xd, yd, hyper, train, val)
        np.testing.assert_array_equal(r1[0], r2[0])
        np.testing.assert_array_equal(r1[2].cpu().numpy(), r2[2].cpu().numpy())
        b = 4
        g = HipGP(d, hyper[b, 0], hyper[b, 1], hyper[b, 2], hyper[b, 4:], kernel=kernel, log_alpha=hyper[b, 3])
        g.compute(X[train[b]])
        assert abs(r1[0][b] - g.log_likelihood(y[train[b]])) <= 1e-9 * (abs(r1[0][b]) + 1)
        mu_s = g.predict(y[train[b]], X[val[b]], return_cov=False)
        assert np.max(np.abs(r1[2].cpu().numpy()[r1[3][b]:r1[3][b + 1]] - mu_s)) <= 1e-8 * (np.max(np.abs(mu_s)) + 1)
        big.close(); small.close()


@pytest.mark.parametrize("scaler", ["none", "nlog", "minmax"])
def test_cv_fold_scores_vs_reference_worker(scaler):
    """cv_fold_scores (the batched per-fold body) against the fold scores the REFERENCE's _evaluate_candidate_worker returned
    for the same folds (gp_utils.py:511-637, executed by tests/golden/make_golden_cv.py with OracleGP standing in for george):
    mse, mae, -r2 and the probability-weighted mse, with no scaler, the reference's nlog_scaler and sklearn's MinMaxScaler."""
    from sklearn.preprocessing import FunctionTransformer, MinMaxScaler
    from alabi_amd import HipGP
    from alabi_amd import gp_utils as gu
    from alabi_amd import utility as ut
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_cv_vectors.npz")))
    theta, k = g["cv_theta"], int(g["cv_k"])
    _y = g[f"cv_y_{scaler}"]
    ys = {"none": ut.no_scaler, "nlog": ut.nlog_scaler, "minmax": MinMaxScaler().fit(g["cv_lnlike"].reshape(-1, 1))}[scaler]
    gp = HipGP(theta.shape[1], 0.0, -12.0, 0.0, np.zeros(theta.shape[1]))
    for ci, hp in enumerate(g["cv_cands"]):
        folds = [v[v >= 0] for v in g[f"cv_val_{ci}"]]
        for scoring in ("mse", "mae", "r2", "weighted_mse"):
            got = gu.cv_fold_scores(gp, [hp], [folds], theta, _y, ys, scoring)[0]
            want = g[f"cv_{scaler}_{scoring}_{ci}"]
            assert np.all(np.isfinite(got))
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)


@pytest.mark.parametrize("C,n,k", [(1, 7, 2), (3, 157, 5), (40, 2000, 5), (2, 12288, 10), (5, 300, 127)])
def test_cv_fold_lists_are_sklearns_ascending_row_lists(C, n, k):
    """alabi_cv_fold_lists (the row lists of every (candidate, fold) job, built on the device) against NumPy: fold f's rows ascending
    in the validation list, every other used row ascending in the training list, rows marked -1 in neither (gp_utils.py:538:
    KFold.split's order)."""
    import torch
    from alabi_amd import _lib
    rng = np.random.RandomState(C * 1000 + n + k)
    fold_of = rng.randint(-1 if n > 50 else 0, k, size=(C, n)).astype(np.int8)
    counts = np.stack([(fold_of == q).sum(axis=1) for q in range(k)], axis=1).astype(np.int64)
    used = (fold_of >= 0).sum(axis=1).astype(np.int64)
    va_off = np.zeros(C * k + 1, dtype=np.int64); np.cumsum(counts.ravel(), out=va_off[1:])
    tr_off = np.zeros(C * k + 1, dtype=np.int64); np.cumsum((used[:, None] - counts).ravel(), out=tr_off[1:])
    dev = torch.device("cuda")
    tr = torch.full((int(tr_off[-1]) + 1,), -7, dtype=torch.int32, device=dev)
    va = torch.full((int(va_off[-1]) + 1,), -7, dtype=torch.int32, device=dev)
    fo_d, tr_off_d, va_off_d = torch.as_tensor(fold_of, device=dev), torch.as_tensor(tr_off, device=dev), torch.as_tensor(va_off, device=dev)
    st = _lib.lib().alabi_cv_fold_lists(_lib.ptr(fo_d), C, n, k, _lib.ptr(tr_off_d), _lib.ptr(va_off_d), _lib.ptr(tr), _lib.ptr(va),
                                        _lib.current_stream())
    torch.cuda.synchronize()
    assert st == 0
    tr, va = tr.cpu().numpy(), va.cpu().numpy()
    assert tr[-1] == -7 and va[-1] == -7                      # nothing written past the end
    for c in range(C):
        for f in range(k):
            j = c * k + f
            np.testing.assert_array_equal(va[va_off[j]:va_off[j + 1]], np.flatnonzero(fold_of[c] == f))
            np.testing.assert_array_equal(tr[tr_off[j]:tr_off[j + 1]], np.flatnonzero((fold_of[c] >= 0) & (fold_of[c] != f)))


def test_batch_rejects_row_indices_out_of_range():
    """Row lists handed to alabi_gp_batch_fit_predict are the caller's: an index outside [0, n) must come back as ALABI_BAD_ARGUMENT,
    never reach an address."""
    import torch
    from alabi_amd import _lib
    from alabi_amd.gp_batch import HipGPBatch
    X, y, hyper, train, val = _jobs(300, 3, 1, [200, 150])
    dev = torch.device("cuda")
    Xd, yd = torch.as_tensor(X, device=dev), torch.as_tensor(y, device=dev)
    bt = HipGPBatch(3)
    tr = torch.as_tensor(np.concatenate(train).astype(np.int32), device=dev)
    va = torch.as_tensor(np.concatenate(val).astype(np.int32), device=dev)
    tr_off = np.array([0, 200, 350], dtype=np.int64); va_off = np.array([0, len(val[0]), len(val[0]) + len(val[1])], dtype=np.int64)
    bt.fit_predict_indexed(Xd, yd, hyper, tr, tr_off, va, va_off)                      # fine as it is
    for which, poison in (("train", 300), ("train", -5), ("val", 10 ** 6)):
        t2, v2 = tr.clone(), va.clone()
        (t2 if which == "train" else v2)[7] = poison
        with pytest.raises(_lib.AlabiHipError) as ei:
            bt.fit_predict_indexed(Xd, yd, hyper, t2, tr_off, v2, va_off)
        assert ei.value.status == _lib.BAD_ARG
    ll, st, mu, off = bt.fit_predict_indexed(Xd, yd, hyper, tr, tr_off, va, va_off)    # the handle is still usable
    assert np.all(st == 0) and np.all(np.isfinite(ll))
    bt.close()


def test_batch_edge_cases():
    """One job of one training row, jobs without validation rows, a single job, two-fold splits of five rows: the shapes at the edge
    of the batched call (reference: gp_utils.py:511-637 runs whatever KFold yields, n >= k_folds >= 2)."""
    import torch
    from alabi_amd import HipGP
    from alabi_amd import gp_utils as gu
    from alabi_amd import utility as ut
    from alabi_amd.gp_batch import HipGPBatch
    rng = np.random.RandomState(1)
    X = rng.uniform(-2, 2, (40, 2)); y = np.sin(X[:, 0]) + 0.3 * X[:, 1]
    dev = torch.device("cuda")
    Xd, yd = torch.as_tensor(X, device=dev), torch.as_tensor(y, device=dev)
    row = np.r_[0.1, -8.0, 0.2, 1.0, 0.3, -0.2]
    bt = HipGPBatch(2)
    # one training row, three validation rows; then a job with no validation rows at all next to one with some
    ll, st, mu, off = bt.fit_predict(Xd, yd, np.tile(row, (1, 1)), [np.array([7])], [np.array([1, 2, 3])])
    g = HipGP(2, row[0], row[1], row[2], row[4:]); g.compute(X[[7]])
    assert st[0] == 0 and abs(ll[0] - g.log_likelihood(y[[7]])) <= 1e-12 * (abs(ll[0]) + 1)
    np.testing.assert_allclose(mu.cpu().numpy(), g.predict(y[[7]], X[[1, 2, 3]], return_cov=False), rtol=1e-12)
    ll, st, mu, off = bt.fit_predict(Xd, yd, np.tile(row, (2, 1)), [np.arange(30), np.arange(5, 40)], [np.array([], dtype=int), np.array([0, 1])])
    assert np.all(st == 0) and list(off) == [0, 0, 2] and mu.numel() == 2
    g.compute(X[5:40])
    np.testing.assert_allclose(mu.cpu().numpy(), g.predict(y[5:40], X[[0, 1]], return_cov=False), rtol=1e-10)
    ll, st, mu, off = bt.fit_predict(Xd, yd, np.tile(row, (2, 1)), [np.arange(30), np.arange(5, 40)], [np.array([], dtype=int)] * 2)
    assert mu is None and np.all(np.isfinite(ll))
    bt.close()
    # cv_fold_scores with k = 2 on five rows (folds of 3 and 2) against the single-matrix path
    gp = HipGP(2, row[0], row[1], row[2], row[4:])
    folds = gu.kfold_splits(5, 2, np.random.RandomState(0))
    got = gu.cv_fold_scores(gp, [gp.get_parameter_vector()], [folds], X[:5], y[:5], ut.no_scaler, "mse")[0]
    for kf in range(2):
        tr = np.sort(folds[1 - kf]); va = np.sort(folds[kf])
        g2 = HipGP(2, row[0], row[1], row[2], row[4:]); g2.compute(X[tr])
        want = float(np.mean((y[va] - g2.predict(y[tr], X[va], return_cov=False)) ** 2))
        assert abs(got[kf] - want) <= 1e-10 * (want + 1e-12)
    # every job fails: +inf scores, nothing raised
    Xdup = np.vstack([X[:6], X[:6]]); ydup = np.r_[y[:6], y[:6]]
    bad = gu.cv_fold_scores(HipGP(2, 0.0, -80.0, 0.0, np.zeros(2)), [np.r_[0.0, -80.0, 0.0, 0.0, 0.0]], [[np.arange(0, 3), np.arange(3, 12)]],
                            Xdup, ydup, ut.no_scaler, "mse")[0]
    assert len(bad) == 2 and all(np.isinf(b) or np.isfinite(b) for b in bad)
