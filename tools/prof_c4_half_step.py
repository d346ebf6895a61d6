import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import EnsembleSampler, HipGP
from alabi_amd.workloads import make_config
cfg = make_config("C4"); h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
s = EnsembleSampler(cfg["W"], cfg["d"], gp, cfg["y"], cfg["bounds"], seed=1)
s.run_mcmc(cfg["p0"], 64, store=False); torch.cuda.synchronize()
t0 = time.perf_counter(); s.run_mcmc(None, 1024, store=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"C4 W={cfg['W']} N={cfg['N']} MULTI={os.environ.get('ALABI_ENS_MULTI')} THREADS={os.environ.get('ALABI_ENS_THREADS')}: {dt/2048*1e6:.2f} us per half step, {cfg['W']*1024/dt:.3g} samples/s, path {s.last_path}")
