"""Soak of the task-queue Cholesky (default for N = 961..4096) under the hyper-parameter searches that factorise on several host
threads and streams at once: 4-D Gaussian shells, 1100 training points, 60 active-learning iterations with a fit every 10
(ml and cv).  Run with ALABI_VERBOSE=1: every wait that runs out (fallback to the launch-per-step path) is reported on stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.preprocessing import StandardScaler
from alabi_amd import SurrogateModel
from alabi_amd.benchmarks import gaussian_shells_nd
g = gaussian_shells_nd(4)
for method in ("ml", "cv"):
    sm = SurrogateModel(lnlike_fn=g["fn"], bounds=g["bounds"], savedir="/tmp/alabi_soak_tq", verbose=False, random_state=5, cache=False)
    sm.init_samples(ntrain=1100, ntest=300)
    t0 = time.perf_counter()
    sm.init_gp(hyperopt_method=method, y_scaler=StandardScaler())
    t1 = time.perf_counter()
    def mse():
        mu = np.asarray(sm.surrogate_log_likelihood(sm.theta_test)).ravel()
        return float(np.mean((mu - np.asarray(sm.y_test).ravel()) ** 2) / np.var(sm.y_test))
    e0 = mse()
    sm.active_train(niter=60, algorithm="bape", gp_opt_freq=10)
    print(f"{method}: init_gp {t1 - t0:.1f} s, 60 iterations with 6 fits in {time.perf_counter() - t1:.1f} s; ntrain {sm.ntrain}; "
          f"scaled test MSE {e0:.4f} -> {mse():.4f}", flush=True)
