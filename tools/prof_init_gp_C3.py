import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sklearn.preprocessing import StandardScaler
from alabi_amd import SurrogateModel
from alabi_amd.workloads import make_config
cfg = make_config("C3")
from scipy.stats import multivariate_normal
rng = np.random.RandomState(2)
cov = cfg.get("cov")
def fn(x):
    x = np.asarray(x).ravel()
    return float(-0.5 * x @ np.linalg.solve(cov, x)) if cov is not None else float(-0.5 * x @ x)
for method in ("ml", "cv"):
    sm = SurrogateModel(lnlike_fn=fn, bounds=[(-3, 3)] * 10, savedir="/tmp/alabi_t10", verbose=False, random_state=0, cache=False)
    sm.init_samples(ntrain=2000, ntest=200)
    t0 = time.perf_counter(); sm.init_gp(hyperopt_method=method, y_scaler=StandardScaler()); t1 = time.perf_counter()
    mu = np.asarray(sm.surrogate_log_likelihood(sm.theta_test)).ravel()
    print(f"{method}: init_gp at N=2000 d=10 {t1-t0:.2f} s; test MSE / var(y) = {np.mean((mu - np.asarray(sm.y_test).ravel())**2) / np.var(sm.y_test):.3e}")
    t0 = time.perf_counter(); sm.active_train(niter=30, gp_opt_freq=15); print(f"   30 active iterations {time.perf_counter()-t0:.2f} s")
