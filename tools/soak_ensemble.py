"""Soak run of the persistent ensemble kernel: 200k steps per configuration, counts time-out fallbacks."""
import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import EnsembleSampler, HipGP
from alabi_amd.workloads import make_config
cfg = make_config("C3"); h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
for E, W in ((1, 256), (2, 256), (1, 1024)):
    p0 = np.random.RandomState(1).uniform(cfg["bounds"][:, 0] * 0.5, cfg["bounds"][:, 1] * 0.5, (W * E, cfg["d"]))
    s = EnsembleSampler(W, cfg["d"], gp, cfg["y"], cfg["bounds"], seed=5, n_ensembles=E)
    s.run_mcmc(p0, 8, store=False)
    t0 = time.perf_counter(); n = 0
    for rep in range(20):
        s.run_mcmc(None, 10000, store=False); n += 10000
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"E={E} W={W}: {n} steps, {W*E*n/dt:.3g} samples/s, path {s.last_path}, fallbacks {getattr(s, 'stream_fallbacks', 0)}, acceptance {s.acceptance_fraction.mean():.3f}, logp finite {bool(torch.isfinite(s._logp).all())}")
# the group kernel (ens_group_kernel) at its own configurations: 100k / 20k steps, fallbacks, acceptance against the launch-per-half-step path
import os
for name, nrep in (("C4", 10), ("C5", 2)):
    cfg = make_config(name); h = cfg["hyper"]
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    acc = {}
    for path in ("group", "half"):
        os.environ["ALABI_ENS_STREAM"] = "1" if path == "group" else "0"
        s = EnsembleSampler(cfg["W"], cfg["d"], gp, cfg["y"], cfg["bounds"], seed=5)
        s.run_mcmc(cfg["p0"], 8, store=False)
        t0 = time.perf_counter(); n = 0
        for rep in range(nrep if path == "group" else 1):
            s.run_mcmc(None, 10000, store=False); n += 10000
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        acc[path] = s.acceptance_fraction.mean()
        print(f"{name} W={cfg['W']} N={cfg['N']}: {n} steps, {cfg['W']*n/dt:.3g} samples/s, path {s.last_path}, fallbacks {getattr(s, 'stream_fallbacks', 0)}, "
              f"acceptance {acc[path]:.4f}, logp finite {bool(torch.isfinite(s._logp).all())}", flush=True)
    del gp
