import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from alabi_amd import SurrogateModel
from alabi_amd.benchmarks import gaussian_shells_nd
g = gaussian_shells_nd(5)
for method in ("ml", "cv"):
    sm = SurrogateModel(lnlike_fn=g["fn"], bounds=g["bounds"], savedir="/tmp/alabi_t", verbose=False, random_state=0, cache=False)
    sm.init_samples(ntrain=1000, ntest=100)
    t0 = time.perf_counter(); sm.init_gp(hyperopt_method=method); t1 = time.perf_counter()
    print(method, "init_gp %.2f s" % (t1 - t0), "test mse", sm.training_results.get("test_mse", [None])[-1] if sm.training_results.get("test_mse") else None)
    t0 = time.perf_counter(); sm.active_train(niter=20, gp_opt_freq=10); print("  20 active iterations %.2f s" % (time.perf_counter() - t0))
    t0 = time.perf_counter(); sm.run_emcee(nwalkers=64, nsteps=2000); print("  run_emcee 64x2000 %.2f s" % (time.perf_counter() - t0), "acc", np.mean(sm.acc_frac))
