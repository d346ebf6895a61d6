"""The reference's default sampling call at C3 (run_emcee with 5e4 steps, 256 walkers): wall time and where the host part goes."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import SurrogateModel
from alabi_amd.workloads import make_config
cfg = make_config("C3")
f = "/tmp/c3_train.npz"; np.savez(f, theta=cfg["X"], y=cfg["y"].reshape(-1, 1))
sm = SurrogateModel(lnlike_fn=cfg["fn"], bounds=cfg["bounds"], savedir="/tmp/alabi_re", verbose=False, random_state=0, cache=False)
sm.init_samples(train_file=f); sm.init_gp(hyperopt_method="ml")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sm.run_emcee(nwalkers=256, nsteps=50_000, min_ess=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run_emcee(5e4 steps, 256 walkers) call {rep}: {dt:.3f} s (sampling {sm.emcee_sampler.last_run_seconds:.3f} s), kept {sm.emcee_samples.shape[0]} samples, tau {sm.autcorr_time:.1f}", flush=True)
pr = cProfile.Profile(); pr.enable()
sm.run_emcee(nwalkers=256, nsteps=50_000, min_ess=0)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
t0 = time.perf_counter(); full = sm.emcee_samples_full; print(f"emcee_samples_full (first access) {time.perf_counter() - t0:.3f} s, shape {full.shape}")
