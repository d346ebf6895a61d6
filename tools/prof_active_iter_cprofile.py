"""cProfile of the active-learning iteration at C3 size (N = 2000, d = 10, 16 384-candidate scan + zoom + polish + append)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import SurrogateModel
from alabi_amd.benchmarks import gaussian_nd
g = gaussian_nd(10, seed=2)
sm = SurrogateModel(lnlike_fn=g["fn"], bounds=g["bounds"], savedir="/tmp/alabi_prof", verbose=False, random_state=0, cache=False)
sm.init_samples(ntrain=2000)
sm.init_gp(hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 3})
sm.active_train(niter=3, algorithm="bape", gp_opt_freq=1000)
torch.cuda.synchronize(); t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
sm.active_train(niter=20, algorithm="bape", gp_opt_freq=1000)
torch.cuda.synchronize(); pr.disable()
print(f"{(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per iteration (under cProfile)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
