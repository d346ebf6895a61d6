"""cProfile of the 2-D Rosenbrock demo's active-learning loop (small N: host overhead dominates)."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sklearn.preprocessing import StandardScaler
from alabi_amd import SurrogateModel
from alabi_amd import benchmarks as bm
sm = SurrogateModel(lnlike_fn=bm.rosenbrock["fn"], bounds=bm.rosenbrock["bounds"], savedir="/tmp/alabi_demo_prof", verbose=False, random_state=0, cache=False)
sm.init_samples(ntrain=50, ntest=200)
sm.init_gp(y_scaler=StandardScaler())
sm.active_train(niter=5, algorithm="bape", gp_opt_freq=1000)
torch.cuda.synchronize(); t0 = time.perf_counter()
sm.active_train(niter=40, algorithm="bape", gp_opt_freq=1000)
torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / 40 * 1e3:.2f} ms per iteration without hyper-fits (N ~ 60-100)")
t0 = time.perf_counter(); sm._opt_gp(**sm.opt_gp_kwargs); torch.cuda.synchronize(); print(f"one hyper-parameter re-optimisation (cv): {(time.perf_counter() - t0) * 1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable()
sm.active_train(niter=40, algorithm="bape", gp_opt_freq=1000)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(22)
