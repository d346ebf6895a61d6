import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import HipGP, EnsembleSampler
from alabi_amd.workloads import make_config
cfg = make_config("C2"); h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
rng = np.random.RandomState(0)
for W in (64, 128, 256):
    p0 = rng.uniform(cfg["bounds"][:, 0] * 0.1, cfg["bounds"][:, 1] * 0.1, (W, cfg["d"]))
    for nsteps in (2000, 20000):
        s = EnsembleSampler(W, cfg["d"], gp, cfg["y"], cfg["bounds"], seed=1)
        t0 = time.perf_counter(); s.run_mcmc(p0, nsteps); torch.cuda.synchronize(); t1 = time.perf_counter()
        s2 = EnsembleSampler(W, cfg["d"], gp, cfg["y"], cfg["bounds"], seed=1)
        t2 = time.perf_counter(); s2.run_mcmc(p0, nsteps); torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"C2 N={cfg['N']} W={W} nsteps={nsteps}: first {t1-t0:.4f} s, second sampler {t3-t2:.4f} s = {(t3-t2)/nsteps*1e6:.2f} us/step, {W*nsteps/(t3-t2):.3e} samples/s")
