import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import SurrogateModel
from alabi_amd.benchmarks import gaussian_nd
g = gaussian_nd(10, seed=2)
sm = SurrogateModel(lnlike_fn=g["fn"], bounds=g["bounds"], savedir="/tmp/alabi_prof", verbose=False, random_state=0, cache=False)
sm.init_samples(ntrain=2000)
sm.init_gp(hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 2})
sm.active_train(niter=2, algorithm="bape", gp_opt_freq=1000)
for refine in (0, 1, 2, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): sm.find_next_point(optimizer_kwargs={"refine": refine})
    torch.cuda.synchronize(); print("refine", refine, "%.2f ms per find_next_point" % ((time.perf_counter() - t0) / 10 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): sm.find_next_point(optimizer_kwargs={"refine": 4})
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
