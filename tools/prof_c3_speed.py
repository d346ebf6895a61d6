import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import EnsembleSampler, HipGP
from alabi_amd.workloads import make_config
cfg = make_config("C3"); h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
s = EnsembleSampler(cfg["W"], cfg["d"], gp, cfg["y"], cfg["bounds"], seed=5)
s.run_mcmc(cfg["p0"], 64, store=False); torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); s.run_mcmc(None, 4096, store=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); s.reset()
print("C3 us per half step %.4f  samples/s %.4e  acc %.4f path %s" % (best / 8192 * 1e6, 256 * 4096 / best, float(s.acceptance_fraction.mean()), s.last_path))
