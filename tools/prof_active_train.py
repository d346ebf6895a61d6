"""Time one active-learning iteration at N~2000, d=10 (find_next_point by candidate scan, refit, bookkeeping)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import SurrogateModel
from alabi_amd.benchmarks import gaussian_nd

g = gaussian_nd(10, seed=2)
sm = SurrogateModel(lnlike_fn=g["fn"], bounds=g["bounds"], savedir="/tmp/alabi_prof", verbose=False, random_state=0, cache=False)
N0 = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sm.init_samples(ntrain=N0)
t0 = time.perf_counter()
sm.init_gp(hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 3})
print(f"init_gp (ML, 3 iterations, analytic device gradient): {time.perf_counter()-t0:.2f} s")
for ncand in (16384, 32768, 65536, 1_000_000):
    sm.active_train(niter=2, algorithm="bape", gp_opt_freq=1000, optimizer_kwargs={"ncand": ncand})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    niter = 10
    sm.active_train(niter=niter, algorithm="bape", gp_opt_freq=1000, optimizer_kwargs={"ncand": ncand})
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / niter
    r = sm.training_results
    print(f"ncand={ncand}: {dt*1e3:.1f} ms per active-learning iteration at N={sm.ntrain} "
          f"(acquisition {np.mean(r['obj_fn_opt_time'][-niter:])*1e3:.1f} ms, refit {np.mean(r['gp_train_time'][-niter:])*1e3:.1f} ms)")
