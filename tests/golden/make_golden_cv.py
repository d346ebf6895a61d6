#!/usr/bin/env python3
"""Generate tests/golden/reference_cv_vectors.npz by EXECUTING the reference's own k-fold worker.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden_cv.py

``alabi/gp_utils.py`` loads by file path with inert ``george`` / ``skopt`` placeholder modules (tests/golden/make_golden.py
does the same).  Its ``_evaluate_candidate_worker`` (gp_utils.py:511-637) touches the GP only through the george protocol
(deepcopy, set_parameter_vector, compute, get_parameter_vector, log_likelihood, predict), so it runs here with
``oracle.gp_oracle.OracleGP`` standing in for george.GP: the fold split (sklearn KFold on the global NumPy stream), the
validity checks, the y un-scaling and the four scoring rules are the REFERENCE's code; only the GP algebra inside is the
restatement.  ``weighted_mse_by_probability`` (gp_utils.py:449-508) is evaluated directly as well.

Stored: inputs (theta, y, hyper-parameter vectors, seeds, the folds KFold produced) and the fold scores the reference
returned.  Nothing of the reference's text is copied.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "reference_cv_vectors.npz")


def main():
    warnings.filterwarnings("ignore")
    import types
    from sklearn.model_selection import KFold
    from sklearn.preprocessing import FunctionTransformer, MinMaxScaler
    import make_golden as mg
    from oracle.gp_oracle import OracleGP

    mg._placeholders()
    ut = mg._load("ref_utility", f"{mg.REF}/utility.py")
    pkg = types.ModuleType("alabi"); pkg.utility = ut
    sys.modules.setdefault("alabi", pkg); sys.modules.setdefault("alabi.utility", ut)
    gpu = mg._load("ref_gp_utils", f"{mg.REF}/gp_utils.py")

    rng = np.random.RandomState(20261005)
    out = {}
    n, d, k = 157, 3, 5                                         # ragged: 157 = 32 + 32 + 31 + 31 + 31
    theta = rng.uniform(-2.0, 2.0, (n, d))
    A = rng.randn(d, d); prec = A @ A.T / d + 0.5 * np.eye(d)
    lnlike = -0.5 * np.einsum("ni,ij,nj->n", theta, prec, theta)          # a smooth negative log-likelihood surface
    scalers = {
        "none": (ut.no_scaler, lnlike.copy()),
        "nlog": (ut.nlog_scaler, ut.nlog_scaler.transform((lnlike - 0.1).reshape(-1, 1)).flatten()),
        "minmax": (MinMaxScaler().fit(lnlike.reshape(-1, 1)), None),
    }
    scalers["minmax"] = (scalers["minmax"][0], scalers["minmax"][0].transform(lnlike.reshape(-1, 1)).flatten())
    # hyper-parameter vectors [mean, log white noise, log amplitude, log M x d]
    cands = np.array([
        np.concatenate([[0.0, -12.0, 1.0], np.log([2.0, 3.0, 1.5])]),
        np.concatenate([[-1.0, -8.0, 0.3], np.log([0.6, 5.0, 2.5])]),
        np.concatenate([[0.5, -14.0, 2.0], np.log([4.0, 4.0, 4.0])]),
    ])
    seeds = np.array([11, 12, 13])
    out.update(cv_theta=theta, cv_lnlike=lnlike, cv_cands=cands, cv_seeds=seeds, cv_k=np.array(k))
    for name, (scaler, _y) in scalers.items():
        out[f"cv_y_{name}"] = _y
        for ci, hp in enumerate(cands):
            gp = OracleGP(d, mean=0.0, log_white_noise=-12.0, log_amp=0.0, log_M=np.zeros(d))
            # the folds the worker is about to draw: KFold(shuffle=True, random_state=None) consumes the global stream
            np.random.seed(int(seeds[ci]))
            val_sets = [v for _, v in KFold(n_splits=k, shuffle=True, random_state=None).split(theta)]
            if name == "none":
                width = max(len(v) for v in val_sets)
                out[f"cv_val_{ci}"] = np.array([np.pad(v, (0, width - len(v)), constant_values=-1) for v in val_sets])
            for scoring in ("mse", "mae", "r2", "weighted_mse"):
                np.random.seed(int(seeds[ci]))
                idx, scores, status = gpu._evaluate_candidate_worker(
                    (ci, hp, gp, theta, _y, scaler, k, scoring, "exponential", 1.0))
                assert status == "success", status
                out[f"cv_{name}_{scoring}_{ci}"] = np.asarray(scores, dtype=np.float64)
    # weighted MSE rule on its own, every weighting method
    yt = rng.normal(-5.0, 3.0, 64); yp = yt + rng.normal(0, 0.3, 64)
    out.update(wmse_true=yt, wmse_pred=yp)
    for method in ("exponential", "linear", "softmax", "rank"):
        for temp in (1.0, 2.5):
            out[f"wmse_{method}_{temp}"] = np.array(gpu.weighted_mse_by_probability(yt, yp, weight_method=method, temperature=temp))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
