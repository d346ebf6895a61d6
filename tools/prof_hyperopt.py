"""Time of one ML hyper-parameter fit (reference path core.py:1237-1327) with the analytic device gradient."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config
for name in ("C2", "C3"):
    cfg = make_config(name); h = cfg["hyper"]
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    y = cfg["y"]
    gp.grad_log_likelihood(y); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 10
    for _ in range(n):
        gp.dirty = True
        g = gp.grad_log_likelihood(y)
    torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); gfd = gp.grad_log_likelihood_fd(y); torch.cuda.synchronize(); tf = time.perf_counter() - t0
    print(f"{name}: N={cfg['N']} d={cfg['d']}: factorise + analytic gradient {1e3 * ta:.2f} ms; central differences {1e3 * tf:.1f} ms; "
          f"max rel diff {np.max(np.abs(g - gfd) / (np.abs(gfd) + 1e-9 * np.max(np.abs(gfd)))):.2e}")
