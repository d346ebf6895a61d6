"""GP construction and hyper-parameter selection helpers on top of HipGP.

Mirrors the parts of alabi/gp_utils.py that the hot path's callers need:
``configure_gp`` (gp_utils.py:170-248), the log-normal length-scale regulariser
``regularization_term`` / ``regularization_gradient`` (gp_utils.py:30-108, values pinned by
tests/golden), and the staged random-search k-fold cross-validation
``optimize_gp_kfold_cv`` (gp_utils.py:511-637, :640-1231, :1234-1367).  Every factorisation,
likelihood and held-out prediction runs in the HIP library (one ``gp.compute`` per fold).
"""
from __future__ import annotations

import copy

import numpy as np

from .gp import HipGP

__all__ = ["configure_gp", "regularization_term", "regularization_gradient", "optimize_gp_kfold_cv"]


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


def _score(y_val, y_pred, scoring):
    if scoring == "mse":
        return float(np.mean((y_val - y_pred) ** 2))
    if scoring == "mae":
        return float(np.mean(np.abs(y_val - y_pred)))
    if scoring == "r2":
        ss_res = np.sum((y_val - y_pred) ** 2)
        ss_tot = np.sum((y_val - np.mean(y_val)) ** 2)
        return float(-(1.0 - ss_res / ss_tot))
    raise ValueError(f"Unsupported scoring method: {scoring}")


def _fold_score(gp_fold, hyperparams, theta_dev, y_dev, _y, train_dev, val_dev, val, inv, scoring, stream):
    """One fold: factorise on `train`, predict `val` (gp_utils.py:568-600) -- one library call on its own HIP stream; the
    training subset is gathered on the device from the resident full set."""
    import torch
    try:
        with torch.cuda.stream(stream):
            gp_fold.set_parameter_vector(hyperparams)
            ll, mu = gp_fold.fit_predict_device(theta_dev.index_select(0, train_dev), y_dev.index_select(0, train_dev),
                                                theta_dev.index_select(0, val_dev))
            if not np.isfinite(ll):
                raise ValueError("GP log-likelihood is invalid")
            _y_pred = mu.cpu().numpy()
        if not np.all(np.isfinite(_y_pred)):
            raise ValueError("GP predictions contain NaN or Inf values")
        return _score(inv(_y[val]), inv(_y_pred), scoring)
    except Exception:  # noqa: BLE001
        return np.inf


def _inverse_map(y_scaler):
    """y_scaler.inverse_transform as a plain function of a 1-D array.  An affine scaler (no_scaler, StandardScaler,
    MinMaxScaler, ...) is recognised by probing and replaced by a * y + b: sklearn's validation costs 50 us per call, and a
    CV search would make 1750 of them under the GIL."""
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


class _FoldWorkers:
    """T GP copies, T HIP streams and T host threads working through (candidate, fold) jobs: the jobs are independent, each
    factorisation fills only part of the chip and every call blocks on a read-back, so they are issued concurrently (the
    reference maps them over a process pool, gp_utils.py:640-700).  ctypes releases the GIL during the library calls;
    distinct handles are thread-safe.  T = ALABI_CV_THREADS (default k, at most 12; more threads only contend for the GIL); 1 runs the jobs one after the other."""

    def __init__(self, gp, k_folds):
        import os
        import queue
        import torch
        from concurrent.futures import ThreadPoolExecutor
        self.k = k_folds
        self.nthreads = max(1, min(12, int(os.environ.get("ALABI_CV_THREADS", k_folds))))
        self.free = queue.SimpleQueue()
        for _ in range(self.nthreads):
            self.free.put((copy.deepcopy(gp), torch.cuda.Stream() if self.nthreads > 1 else torch.cuda.current_stream()))
        self.pool = ThreadPoolExecutor(max_workers=self.nthreads) if self.nthreads > 1 else None

    def _job(self, args):
        res = self.free.get()
        try:
            return _fold_score(res[0], *args, res[1])
        finally:
            self.free.put(res)

    def run_many(self, jobs):
        """jobs: the argument tuples of _fold_score after the GP and before the stream -> scores in the same order"""
        import torch
        if self.pool is None:
            return [self._job(j) for j in jobs]
        torch.cuda.current_stream().synchronize()
        return list(self.pool.map(self._job, jobs))

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=True)


def _fold_jobs(hyperparams, _theta, _y, folds, y_scaler, scoring, inv=None, dev=None):
    """dev = (theta_dev, y_dev): the full training set resident on the device (uploaded once per search)"""
    import torch
    from .gp import _to_dev
    k_folds = len(folds)
    inv = _inverse_map(y_scaler) if inv is None else inv
    if dev is None:
        dev = (_to_dev(np.ascontiguousarray(_theta, dtype=np.float64), 2), _to_dev(np.ascontiguousarray(_y, dtype=np.float64)))
    jobs = []
    for k in range(k_folds):
        val = np.sort(folds[k])
        train = np.sort(np.concatenate([folds[j] for j in range(k_folds) if j != k]))
        idx = torch.as_tensor(np.concatenate([train, val]), device=dev[0].device)
        jobs.append((hyperparams, dev[0], dev[1], _y, idx[:len(train)], idx[len(train):], val, inv, scoring))
    return jobs


def _evaluate_candidate(hyperparams, gp, _theta, _y, y_scaler, k_folds, scoring, rng, workers=None):
    """Fold scores of one hyper-parameter vector (gp_utils.py:511-637); np.inf marks a failed fold."""
    if not np.all(np.isfinite(hyperparams)):
        return None
    n = len(_theta)
    perm = rng.permutation(n)                      # KFold(shuffle=True, random_state=None)
    folds = np.array_split(perm, k_folds)
    own = workers is None
    if own:
        workers = _FoldWorkers(gp, k_folds)
    try:
        return workers.run_many(_fold_jobs(hyperparams, _theta, _y, folds, y_scaler, scoring))
    finally:
        if own:
            workers.close()


def _mean_scores(cands, gp, _theta, _y, y_scaler, k_folds, scoring, rng):
    """Mean fold score per candidate.  The fold permutations are drawn first, in candidate order (the random stream is the
    same as when the candidates are evaluated one after the other), then all (candidate, fold) jobs go to the workers."""
    out = np.full(len(cands), np.inf)
    n = len(_theta)
    inv = _inverse_map(y_scaler)
    from .gp import _to_dev
    dev = (_to_dev(np.ascontiguousarray(_theta, dtype=np.float64), 2), _to_dev(np.ascontiguousarray(_y, dtype=np.float64)))
    jobs, owner = [], []
    for i, hp in enumerate(cands):
        if not np.all(np.isfinite(hp)):
            continue
        folds = np.array_split(rng.permutation(n), k_folds)
        jb = _fold_jobs(hp, _theta, _y, folds, y_scaler, scoring, inv, dev)
        jobs.extend(jb); owner.extend([i] * len(jb))
    if not jobs:
        return out
    workers = _FoldWorkers(gp, k_folds)
    try:
        scores = np.asarray(workers.run_many(jobs), dtype=np.float64)
    finally:
        workers.close()
    owner = np.asarray(owner)
    for i in np.unique(owner):
        s = scores[owner == i]
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
                         weighted_mse_method="exponential", weighted_mse_factor=1.0, verbose=True, random_state=None):
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
    scores = _mean_scores(cands, gp, _theta, _y, y_scaler, k_folds, scoring, rng)
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
        s = _mean_scores(c, gp, _theta, _y, y_scaler, k_folds, scoring, rng)
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
