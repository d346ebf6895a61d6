"""Soak of the active-learning loop on a 4-D Gaussian-shell problem: 400 iterations with a hyper-parameter fit every 10 (ml and cv),
append-refits across several 64-point padding boundaries, threaded restarts / folds."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sklearn.preprocessing import StandardScaler
from alabi_amd import SurrogateModel
from alabi_amd.benchmarks import gaussian_shells_nd
g = gaussian_shells_nd(4)
for method in ("ml", "cv"):
    sm = SurrogateModel(lnlike_fn=g["fn"], bounds=g["bounds"], savedir="/tmp/alabi_soak", verbose=False, random_state=3, cache=False)
    sm.init_samples(ntrain=300, ntest=300)
    sm.init_gp(hyperopt_method=method, y_scaler=StandardScaler())
    def mse():
        mu = np.asarray(sm.surrogate_log_likelihood(sm.theta_test)).ravel()
        return float(np.mean((mu - np.asarray(sm.y_test).ravel()) ** 2) / np.var(sm.y_test))
    e0 = mse(); t0 = time.perf_counter()
    sm.active_train(niter=400, algorithm="bape", gp_opt_freq=10)
    dt = time.perf_counter() - t0
    print(f"{method}: 400 iterations with 40 hyper-parameter fits in {dt:.1f} s; ntrain {sm.ntrain}; scaled test MSE {e0:.3f} -> {mse():.3f}")
