import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sklearn.preprocessing import StandardScaler
from alabi_amd import SurrogateModel
def fn(x):
    x = np.asarray(x).ravel(); return float(-0.5 * x @ x)
for thr in ("1", "3"):
    os.environ["ALABI_ML_THREADS"] = thr
    for n in (500, 2000):
        sm = SurrogateModel(lnlike_fn=fn, bounds=[(-3, 3)] * 10, savedir="/tmp/alabi_t10", verbose=False, random_state=0, cache=False)
        sm.init_samples(ntrain=n, ntest=200)
        t0 = time.perf_counter(); sm.init_gp(hyperopt_method="ml", y_scaler=StandardScaler()); t1 = time.perf_counter()
        t2 = time.perf_counter(); sm.active_train(niter=20, gp_opt_freq=10); t3 = time.perf_counter()
        print(f"threads {thr} N={n}: init_gp(ml) {t1-t0:.3f} s; 20 iterations with 2 refits {t3-t2:.3f} s; hyper {np.round(sm.gp.get_parameter_vector()[:4], 5)}")
