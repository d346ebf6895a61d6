"""Timing of the ensemble paths on the large configurations (C4: N=5000 W=1024; C5-sized: N=10000 d=20 W=2048) and any
--config / --W / --N override: microseconds per half step and samples/s, per path.
  python tools/prof_group_kernel.py [--configs C4,C5] [--steps 1024] [--paths group,half]
Environment switches of the group kernel: ALABI_ENS_GROUP_Q / _G (blocking), ALABI_ENS_GROUP_THREADS, ALABI_ENS_GROUP_XCD."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from alabi_amd import EnsembleSampler, HipGP  # noqa: E402
from alabi_amd.workloads import make_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C4,C5")
ap.add_argument("--steps", type=int, default=1024)
ap.add_argument("--paths", default="group,half")
ap.add_argument("--W", type=int, default=None)
ap.add_argument("--N", type=int, default=None)
ap.add_argument("--E", type=int, default=1)
args = ap.parse_args()

for name in args.configs.split(","):
    cfg = make_config(name, N=args.N, W=args.W)
    h = cfg["hyper"]
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    gp.compute(cfg["X"])
    p0 = cfg["p0"] if args.E == 1 else np.concatenate([cfg["p0"]] * args.E)
    for path in args.paths.split(","):
        os.environ["ALABI_ENS_STREAM"] = "0" if path == "half" else "1"
        os.environ["ALABI_ENS_GROUP"] = {"group": "1", "stream": "0", "half": "0"}[path]
        s = EnsembleSampler(cfg["W"], cfg["d"], gp, cfg["y"], cfg["bounds"], seed=1, n_ensembles=args.E)
        s.run_mcmc(p0, 64, store=False)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            s.run_mcmc(None, args.steps, store=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            s.reset()
        print(f"{name} N={cfg['N']} d={cfg['d']} W={cfg['W']} E={args.E} path={s.last_path:22s} "
              f"{best / (2 * args.steps) * 1e6:7.2f} us per half step  {cfg['W'] * args.E * args.steps / best:.3e} samples/s", flush=True)
