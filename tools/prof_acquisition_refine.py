import numpy as np, torch, time, sys
sys.path.insert(0, "/root/repo")
from alabi_amd import SurrogateModel
"""Quality and cost of the acquisition search: plain candidate scan vs scan + zoom stages (C2-like 5-D shell, AGP and BAPE)."""
from alabi_amd.benchmarks import gaussian_shells_nd
g = gaussian_shells_nd(5)
sm = SurrogateModel(lnlike_fn=g["fn"], bounds=g["bounds"], savedir="/tmp/alabi_prof", verbose=False, random_state=0, cache=False)
sm.init_samples(ntrain=500)
sm.init_gp(hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 3})
for algo in ("agp", "bape"):
  sm.active_train(niter=1, algorithm=algo, gp_opt_freq=1000, optimizer_kwargs={"ncand": 4096, "refine": 0})
  print(algo)
  for refine, ncand, nper, polish in ((0, 65536, 0, 0), (0, 1000000, 0, 0), (3, 65536, 16384, 0), (4, 65536, 4096, 0), (5, 65536, 4096, 0),
                                      (4, 32768, 4096, 0), (4, 32768, 4096, 30), (2, 32768, 4096, 30), (0, 32768, 0, 30),
                                      (4, 16384, 4096, 30), (4, 8192, 4096, 30), (4, 16384, 2048, 30)):
        vals = []; t0 = time.perf_counter()
        for rep in range(8):
            sm.random_state = rep
            sm.find_next_point(optimizer_kwargs={"ncand": ncand, "refine": refine, "nrefine": max(nper, 1), "polish": polish})
            vals.append(sm.last_acquisition_value)
        torch.cuda.synchronize()
        print(f"  refine {refine} x {nper} ncand {ncand} polish {polish}: {(time.perf_counter()-t0)/8*1e3:.1f} ms per call, acquisition minimum found {np.round(vals, 3)}")
