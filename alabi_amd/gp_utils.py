"""GP construction and hyper-parameter selection helpers on top of HipGP.

Mirrors the parts of alabi/gp_utils.py that the hot path's callers need:
``configure_gp`` (gp_utils.py:170-248), the log-normal length-scale regulariser
``regularization_term`` / ``regularization_gradient`` (gp_utils.py:30-108, values pinned by
tests/golden), and the staged random-search k-fold cross-validation
``optimize_gp_kfold_cv`` (gp_utils.py:511-637, :640-1231, :1234-1367).  Every factorisation,
likelihood and held-out prediction runs in the HIP library: all (candidate, fold) jobs of a search
stage in ONE batched call (alabi_gp_batch_fit_predict, gp_batch.py).
"""
from __future__ import annotations

import copy

import numpy as np

from .gp import HipGP

__all__ = ["configure_gp", "regularization_term", "regularization_gradient", "optimize_gp_kfold_cv", "cv_fold_scores",
           "kfold_splits", "weighted_mse_by_probability"]


def regularization_term(hparams, lengthscale_indices, amp_0=1.0, mu_0=1.0, sigma_0=2.0):
    """-log LogNormal(mu_0 + log sqrt(n), sigma_0) prior summed over the log length-scales.

    ``n`` is len(hparams) -- the FULL hyper-parameter vector, as in the reference (gp_utils.py:51)."""
    hparams = np.asarray(hparams, dtype=np.float64)
    loc = mu_0 + 0.5 * np.log(len(hparams))
    ls = hparams[lengthscale_indices]
    return amp_0 * np.sum(ls + 0.5 * np.log(2 * np.pi * sigma_0 ** 2) + (ls - loc) ** 2 / (2 * sigma_0 ** 2))


def regularization_gradient(hparams, lengthscale_indices, amp_0=1.0, mu_0=1.0, sigma_0=2.0):
    hparams = np.asarray(hparams, dtype=np.float64)
    loc = mu_0 + 0.5 * np.log(len(hparams))
    grad = np.zeros_like(hparams)
    ls = hparams[lengthscale_indices]
    grad[lengthscale_indices] = (1.0 + (ls - loc) / sigma_0 ** 2) / np.exp(ls)
    return amp_0 * grad


def configure_gp(theta, y, kernel, fit_amp=True, fit_mean=True, fit_white_noise=False, white_noise=-12,
                 hyperparameters=None):
    """Build a HipGP for (theta, y) and factorise it; None if the factorisation fails (gp_utils.py:170-248).

    ``kernel`` is the dict produced by ``SurrogateModel.init_gp`` ({"log_M": [...], "name": ...}).
    ``kernel *= var(y)`` becomes log_constant = log(var(y) / ndim): george's ``scalar * kernel`` builds
    ``ConstantKernel(log_constant=log(scalar / ndim))`` (recalled upstream behaviour).
    """
    theta = np.asarray(theta, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if np.any(~np.isfinite(theta)):
        raise ValueError("All theta values must be finite!")
    if np.any(~np.isfinite(y)):
        raise ValueError("All y values must be finite!")
    ndim = theta.shape[1]
    log_const = np.log(np.var(y) / ndim) if fit_amp else kernel.get("log_constant", 0.0)
    gp = HipGP(ndim, mean=np.median(y), white_noise=white_noise, log_constant=log_const, log_M=kernel["log_M"],
               fit_mean=fit_mean, fit_white_noise=fit_white_noise, kernel=kernel.get("name", "ExpSquaredKernel"),
               log_alpha=kernel.get("log_alpha", 1.0), fit_amp=fit_amp)
    if hyperparameters is not None:
        if np.any(~np.isfinite(hyperparameters)):
            raise ValueError("All hyperparameter values must be finite!")
        gp.set_parameter_vector(hyperparameters)
    try:
        gp.compute(theta)
    except Exception as e:  # noqa: BLE001 - the reference swallows every factorisation error here
        print(f"configure_gp error: {e}")
        return None
    return gp


def weighted_mse_by_probability(y_true, y_pred, weight_method="exponential", temperature=1.0):
    """MSE with larger weights on the points of higher log-likelihood (gp_utils.py:449-508; values pinned by
    tests/golden/reference_cv_vectors.npz): weights exp(y/T), y - min(y) + 1e-6, softmax(y/T) or the rank, scaled to mean 1."""
    y_true = np.asarray(y_true)
    y_pred = np.asarray(y_pred)
    if weight_method == "exponential":
        w = np.exp(y_true / temperature)
    elif weight_method == "linear":
        w = y_true - np.min(y_true) + 1e-6
    elif weight_method == "softmax":
        w = np.exp(y_true / temperature)
        w = w / np.sum(w) * len(w)
    elif weight_method == "rank":
        w = np.argsort(np.argsort(y_true)) + 1
    else:
        raise ValueError(f"Unknown weight_method: {weight_method}")
    w = w / np.mean(w)
    return np.average((y_true - y_pred) ** 2, weights=w)


def _score(y_val, y_pred, scoring, weighted_mse_method="exponential", weighted_mse_factor=1.0):
    """One fold's score, lower is better (gp_utils.py:603-619: sklearn's mean_squared_error / mean_absolute_error / -r2_score,
    or weighted_mse_by_probability)."""
    if scoring == "mse":
        return float(np.mean((y_val - y_pred) ** 2))
    if scoring == "mae":
        return float(np.mean(np.abs(y_val - y_pred)))
    if scoring == "r2":
        ss_res = np.sum((y_val - y_pred) ** 2)
        ss_tot = np.sum((y_val - np.mean(y_val)) ** 2)
        return float(-(1.0 - ss_res / ss_tot))
    if scoring == "weighted_mse":
        return float(weighted_mse_by_probability(y_val, y_pred, weight_method=weighted_mse_method, temperature=weighted_mse_factor))
    raise ValueError(f"Unsupported scoring method: {scoring}")


def _inverse_map(y_scaler):
    """y_scaler.inverse_transform as a plain function of a 1-D array.  An affine scaler (no_scaler, StandardScaler,
    MinMaxScaler, ...) is recognised by probing and replaced by a * y + b: sklearn's validation costs 50 us per call, and a
    CV search makes 1750 of them."""
    probe = np.array([[-1.7], [0.0], [0.9], [3.3]])
    try:
        out = np.asarray(y_scaler.inverse_transform(probe), dtype=np.float64).ravel()
        if np.all(np.isfinite(out)):
            b = out[1]
            a = (out[3] - out[0]) / (probe[3, 0] - probe[0, 0])
            if np.allclose(out, a * probe.ravel() + b, rtol=1e-13, atol=1e-13 * (abs(a) + abs(b) + 1)):
                return lambda v, a=a, b=b: a * np.asarray(v, dtype=np.float64).ravel() + b
    except Exception:  # noqa: BLE001
        pass
    return lambda v: y_scaler.inverse_transform(np.asarray(v).reshape(-1, 1)).flatten()


def kfold_splits(n, k_folds, rng):
    """The validation sets of sklearn's KFold(n_splits=k, shuffle=True) on `rng`'s stream (gp_utils.py:538): one shuffle of
    arange(n), cut into k consecutive pieces, the first n % k one longer (pinned: tests/golden/reference_cv_vectors.npz)."""
    return np.array_split(rng.permutation(n), k_folds)


def _segment_scores(y_val, y_pred, off, scoring, weighted_mse_method="exponential", weighted_mse_factor=1.0):
    """_score for every segment [off[b], off[b + 1]) of the concatenated un-scaled (y_val, y_pred) at once (np.add.reduceat);
    None where the rule has no vectorised form (the caller then scores segment by segment)."""
    lens = np.diff(off).astype(np.float64)
    starts = np.asarray(off[:-1], dtype=np.int64)
    if len(starts) == 0 or np.any(lens <= 0):
        return None
    seg = lambda v: np.add.reduceat(v, starts)  # noqa: E731
    e2 = (y_val - y_pred) ** 2
    with np.errstate(all="ignore"):
        if scoring == "mse":
            return seg(e2) / lens
        if scoring == "mae":
            return seg(np.abs(y_val - y_pred)) / lens
        if scoring == "r2":
            mean = np.repeat(seg(y_val) / lens, np.diff(off))
            return -(1.0 - seg(e2) / seg((y_val - mean) ** 2))
        if scoring == "weighted_mse" and weighted_mse_method in ("exponential", "softmax", "linear"):
            if weighted_mse_method == "linear":
                w = y_val - np.repeat(np.minimum.reduceat(y_val, starts), np.diff(off)) + 1e-6
            else:
                w = np.exp(y_val / weighted_mse_factor)
            return seg(w * e2) / seg(w)                   # = np.average(e2, weights=w / mean(w))
    return None


def _cv_scores(gp, hyper_rows, fold_of, k_folds, _y, dev, inv, scoring, weighted_mse_method, weighted_mse_factor, batch):
    """Scores [C][k] of C hyper-parameter rows (HipGP.full_hyper) whose folds are given as fold_of [C, n] (fold number of every
    row, -1 = in no fold): the row lists of all C k jobs are built on the device (alabi_cv_fold_lists: ascending rows on both
    sides, as sklearn's KFold.split yields them), ONE batched library call, one read-back of the held-out means, vectorised scoring."""
    import torch
    C = len(hyper_rows)
    out = np.full((C, k_folds), np.inf)
    if C == 0:
        return out
    from . import _lib
    fold_of = np.ascontiguousarray(fold_of, dtype=np.int8)
    n = fold_of.shape[1]
    counts = np.stack([(fold_of == q).sum(axis=1) for q in range(k_folds)], axis=1).astype(np.int64)   # [C, k] rows per fold
    used = (fold_of >= 0).sum(axis=1).astype(np.int64)
    va_off = np.zeros(C * k_folds + 1, dtype=np.int64); np.cumsum(counts.ravel(), out=va_off[1:])
    tr_off = np.zeros(C * k_folds + 1, dtype=np.int64); np.cumsum((used[:, None] - counts).ravel(), out=tr_off[1:])
    d0 = dev[0].device
    fo = torch.as_tensor(fold_of, device=d0)                                                            # [C, n]
    tr_dev = torch.empty(max(int(tr_off[-1]), 1), dtype=torch.int32, device=d0)
    va_dev = torch.empty(max(int(va_off[-1]), 1), dtype=torch.int32, device=d0)
    tr_off_dev, va_off_dev = torch.as_tensor(tr_off, device=d0), torch.as_tensor(va_off, device=d0)   # (named: they must outlive the call)
    _lib.check(_lib.lib().alabi_cv_fold_lists(_lib.ptr(fo), C, n, k_folds, _lib.ptr(tr_off_dev), _lib.ptr(va_off_dev), _lib.ptr(tr_dev),
                                              _lib.ptr(va_dev), _lib.current_stream()), "alabi_cv_fold_lists")
    tr_dev, va_dev = tr_dev[:int(tr_off[-1])], va_dev[:int(va_off[-1])]
    hyper = np.repeat(np.asarray(hyper_rows, dtype=np.float64), k_folds, axis=0)
    ll, status, mu, off = batch.fit_predict_indexed(dev[0], dev[1], hyper, tr_dev, tr_off, va_dev, va_off)
    if mu is None:
        return out
    mu_host = mu.cpu().numpy()
    val_rows = va_dev.cpu().numpy()
    good = (status == 0) & np.isfinite(ll) & (np.diff(off) > 0)
    fin = np.isfinite(mu_host)
    if not np.all(fin):
        good &= np.add.reduceat((~fin).astype(np.int64), off[:-1].clip(max=len(mu_host) - 1)) == 0
    sc = None
    try:
        y_val, y_pred = inv(_y[val_rows]), inv(np.where(fin, mu_host, 0.0))
        sc = _segment_scores(y_val, y_pred, off, scoring, weighted_mse_method, weighted_mse_factor)
    except Exception:  # noqa: BLE001 - e.g. a scaler that rejects the whole array: score fold by fold below
        sc = None
    flat = out.reshape(-1)
    if sc is not None:
        ok = good & np.isfinite(sc)
        flat[ok] = sc[ok]
        return out
    for b in np.flatnonzero(good):
        try:
            v = _score(inv(_y[val_rows[off[b]:off[b + 1]]]), inv(mu_host[off[b]:off[b + 1]]), scoring, weighted_mse_method,
                       weighted_mse_factor)
        except Exception:  # noqa: BLE001 - the reference marks a fold that raises as failed
            continue
        if np.isfinite(v):
            flat[b] = v
    return out


def cv_fold_scores(gp, candidates, fold_sets, _theta, _y, y_scaler, scoring="mse", weighted_mse_method="exponential",
                   weighted_mse_factor=1.0, batch=None, dev=None, inv=None):
    """Fold scores of several hyper-parameter vectors in ONE batched library call (gp_utils.py:511-637 per candidate):
    ``fold_sets[c]`` = the k validation index sets of candidate c; per fold the GP is factorised on the other rows with
    ``candidates[c]``, its log-likelihood checked, the held-out rows predicted, and the score taken on UN-scaled values
    (``y_scaler.inverse_transform``).  Returns [len(candidates)][k] floats, np.inf = failed fold (not positive definite,
    non-finite likelihood or predictions, as the reference's per-fold try/except)."""
    from .gp import _to_dev
    from .gp_batch import HipGPBatch
    _y = np.asarray(_y, dtype=np.float64).ravel()
    if not len(candidates):
        return []
    k_folds = len(fold_sets[0])
    if any(len(f) != k_folds for f in fold_sets):
        raise ValueError("every candidate needs the same number of folds")
    own = batch is None
    if own:
        batch = HipGPBatch(gp.ndim, gp.kernel_name)
    try:
        inv = _inverse_map(y_scaler) if inv is None else inv
        if dev is None:
            dev = (_to_dev(np.ascontiguousarray(_theta, dtype=np.float64), 2), _to_dev(np.ascontiguousarray(_y, dtype=np.float64)))
        fold_of = np.full((len(candidates), len(_y)), -1, dtype=np.int8)
        for c, folds in enumerate(fold_sets):
            for k, f in enumerate(folds):
                fold_of[c, np.asarray(f, dtype=np.int64)] = k
        rows = [gp.full_hyper(hp) for hp in candidates]
        return _cv_scores(gp, rows, fold_of, k_folds, _y, dev, inv, scoring, weighted_mse_method, weighted_mse_factor, batch).tolist()
    finally:
        if own:
            batch.close()


def _evaluate_candidate(hyperparams, gp, _theta, _y, y_scaler, k_folds, scoring, rng, **kw):
    """Fold scores of one hyper-parameter vector (gp_utils.py:511-637); np.inf marks a failed fold."""
    if not np.all(np.isfinite(hyperparams)):
        return None
    return cv_fold_scores(gp, [hyperparams], [kfold_splits(len(_theta), k_folds, rng)], _theta, _y, y_scaler, scoring, **kw)[0]


def _mean_scores(cands, gp, _theta, _y, y_scaler, k_folds, scoring, rng, batch=None, ranks=None, weighted_mse_method="exponential",
                 weighted_mse_factor=1.0):
    """Mean fold score per candidate.  The fold permutations are drawn first, in candidate order (the random stream is the
    same as when the candidates are evaluated one after the other), then ALL (candidate, fold) jobs of the stage go to one
    batched call.  ``ranks=(rank, world)``: this process evaluates candidates rank, rank + world, ... only (the caller
    combines the partial score vectors: np.inf elsewhere)."""
    from .gp import _to_dev
    from .gp_batch import HipGPBatch
    out = np.full(len(cands), np.inf)
    n = len(_theta)
    _y = np.asarray(_y, dtype=np.float64).ravel()
    if k_folds > 127:
        raise ValueError("k_folds must be <= 127")
    # sklearn's KFold(shuffle=True): one shuffle of arange(n) cut into k consecutive pieces, the first n % k one longer
    pattern = np.repeat(np.arange(k_folds, dtype=np.int8), [n // k_folds + (1 if q < n % k_folds else 0) for q in range(k_folds)])
    sel, perms = [], []
    for i, hp in enumerate(cands):
        if not np.all(np.isfinite(hp)):
            continue
        perm = rng.permutation(n)                             # drawn for EVERY finite candidate: the stream does not depend on `ranks`
        if ranks is not None and i % ranks[1] != ranks[0]:
            continue
        sel.append(i); perms.append(perm)
    if not sel:
        return out
    fold_of = np.empty((len(sel), n), dtype=np.int8)
    np.put_along_axis(fold_of, np.asarray(perms, dtype=np.int64), pattern[None, :], axis=1)
    dev = (_to_dev(np.ascontiguousarray(_theta, dtype=np.float64), 2), _to_dev(np.ascontiguousarray(_y, dtype=np.float64)))
    own = batch is None
    if own:
        batch = HipGPBatch(gp.ndim, gp.kernel_name)
    try:
        scores = _cv_scores(gp, [gp.full_hyper(cands[i]) for i in sel], fold_of, k_folds, _y, dev, _inverse_map(y_scaler), scoring,
                            weighted_mse_method, weighted_mse_factor, batch)
    finally:
        if own:
            batch.close()
    for i, s in zip(sel, scores):
        ok = s[np.isfinite(s)]
        if len(ok):
            out[i] = np.mean(ok)
    return out


def _perturbed(best, n_candidates, width, rng):
    """Stage-2/3 candidates: the incumbent plus N(0, width) perturbations (gp_utils.py:1234-1367)."""
    best = np.asarray(best, dtype=np.float64)
    ls = best[2:] if len(best) > 2 else np.array([])
    uniform = len(ls) > 1 and np.allclose(ls, ls[0])
    cands = [best.copy()]
    for _ in range(n_candidates - 1):
        if uniform:
            c = best.copy()
            c[:2] += rng.normal(0, width, 2)
            c[2:] += rng.normal(0, width)
        else:
            c = best + rng.normal(0, width, len(best))
        cands.append(c)
    return np.array(cands)


def optimize_gp_kfold_cv(gp, _theta, _y, hyperparameter_candidates, y_scaler, k_folds=5, scoring="mse", pool=None,
                         stage2_candidates=None, stage2_width=0.5, stage3_candidates=None, stage3_width=0.2,
                         weighted_mse_method="exponential", weighted_mse_factor=1.0, verbose=True, random_state=None,
                         ranks=None, reduce_scores=None):
    """Pick the candidate with the lowest mean k-fold validation score, refine around it in up to two
    further random-search stages, set it on ``gp`` and factorise on all the data.  Returns ``gp`` or
    None when every candidate failed (gp_utils.py:640-1231)."""
    _theta = np.asarray(_theta, dtype=np.float64)
    _y = np.asarray(_y, dtype=np.float64).squeeze()
    cands = np.atleast_2d(np.asarray(hyperparameter_candidates, dtype=np.float64))
    n = len(_theta)
    if len(_y) != n:
        raise ValueError(f"_theta and _y must have same length, got {len(_theta)} and {len(_y)}")
    if n < k_folds:
        raise ValueError(f"Number of samples ({n}) must be >= k_folds ({k_folds})")
    if k_folds < 2:
        raise ValueError(f"k_folds must be >= 2, got {k_folds}")
    rng = np.random.RandomState(random_state)
    from .gp_batch import HipGPBatch
    batch = HipGPBatch(gp.ndim, gp.kernel_name)
    kw = dict(batch=batch, weighted_mse_method=weighted_mse_method, weighted_mse_factor=weighted_mse_factor, ranks=ranks)
    try:
        return _optimize_gp_kfold_cv(gp, _theta, _y, cands, y_scaler, k_folds, scoring, stage2_candidates, stage2_width,
                                     stage3_candidates, stage3_width, verbose, rng, kw, reduce_scores)
    finally:
        batch.close()


def _optimize_gp_kfold_cv(gp, _theta, _y, cands, y_scaler, k_folds, scoring, stage2_candidates, stage2_width, stage3_candidates,
                          stage3_width, verbose, rng, kw, reduce_scores):
    def _mean_scores_all(c):
        s = _mean_scores(c, gp, _theta, _y, y_scaler, k_folds, scoring, rng, **kw)
        return s if reduce_scores is None else reduce_scores(s)
    scores = _mean_scores_all(cands)
    if np.all(np.isinf(scores)):
        return None
    best = cands[int(np.argmin(scores))]
    best_score = float(np.min(scores))
    if verbose:
        print(f"CV stage 1: best {scoring} = {best_score:.6g} over {len(cands)} candidates")
    stages = []
    if stage2_candidates is not None:
        stages.append((stage2_candidates, stage2_width))
        if stage3_candidates is not None:
            stages.append((stage3_candidates, stage3_width))
    for k, (ncand, width) in enumerate(stages, start=2):
        c = _perturbed(best, int(ncand), width, rng)
        s = _mean_scores_all(c)
        if not np.all(np.isinf(s)) and np.min(s) < best_score:
            best, best_score = c[int(np.argmin(s))], float(np.min(s))
        if verbose:
            print(f"CV stage {k}: best {scoring} = {best_score:.6g}")
    try:
        gp.set_parameter_vector(best)
        gp.compute(_theta)
    except Exception:  # noqa: BLE001
        return gp
    return gp
