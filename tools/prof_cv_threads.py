import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sklearn.preprocessing import StandardScaler
from alabi_amd import SurrogateModel
def fn(x):
    x = np.asarray(x).ravel(); return float(-0.5 * x @ x)
for thr in ("1", "5"):
    os.environ["ALABI_CV_THREADS"] = thr
    for n in (500, 2000):
        sm = SurrogateModel(lnlike_fn=fn, bounds=[(-3, 3)] * 10, savedir="/tmp/alabi_t10", verbose=False, random_state=0, cache=False)
        sm.init_samples(ntrain=n, ntest=200)
        t0 = time.perf_counter(); sm.init_gp(hyperopt_method="cv", y_scaler=StandardScaler()); t1 = time.perf_counter()
        mu = np.asarray(sm.surrogate_log_likelihood(sm.theta_test)).ravel()
        print(f"threads {thr} N={n}: init_gp(cv) {t1-t0:.2f} s; hyper {np.round(sm.gp.get_parameter_vector()[:4], 4)}; test MSE/var {np.mean((mu - np.asarray(sm.y_test).ravel())**2) / np.var(sm.y_test):.3e}")
