"""End-to-end sanity on the reference's own 2-D Rosenbrock benchmark (benchmarks.py:46-52): test error of the surrogate before and
after active learning with bape / agp (scan + zoom + polish acquisition), and the posterior mean from run_emcee."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sklearn.preprocessing import StandardScaler
from alabi_amd import SurrogateModel
from alabi_amd.benchmarks import rosenbrock
for algo in ("bape", "agp"):
    sm = SurrogateModel(lnlike_fn=rosenbrock["fn"], bounds=rosenbrock["bounds"], savedir="/tmp/alabi_demo", verbose=False, random_state=1, cache=False)
    sm.init_samples(ntrain=50, ntest=500)
    # standardised y: the reference builds the amplitude box from var(y) on a log parameter (core.py:654), which only works
    # for y of order one -- its own examples use small-range functions or a y scaler
    sm.init_gp(hyperopt_method="ml", y_scaler=StandardScaler())
    def mse():
        mu = sm.surrogate_log_likelihood(sm.theta_test)
        return float(np.mean((np.asarray(mu).ravel() - np.asarray(sm.y_test).ravel()) ** 2))
    e0 = mse(); t0 = time.perf_counter()
    sm.active_train(niter=150, algorithm=algo, gp_opt_freq=25)
    dt = time.perf_counter() - t0
    print(f"{algo}: test MSE {e0:.4g} -> {mse():.4g} after 150 iterations ({dt:.2f} s, {dt/150*1e3:.1f} ms each incl. 6 hyper-parameter fits); ntrain {sm.ntrain}")
    sm.run_emcee(nwalkers=32, nsteps=5000)
    print("   emcee: mean", np.round(np.mean(sm.emcee_samples, axis=0), 3), "acc", round(float(np.mean(sm.acc_frac)), 3))
