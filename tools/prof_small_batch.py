"""Latency of variance requests of at most 16 queries: cached-L^-1 path vs the substitution kernel (C3, C4)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config
for name in ("C3", "C4"):
    cfg = make_config(name); h = cfg["hyper"]
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    y = torch.as_tensor(cfg["y"], device="cuda")
    Xs = torch.as_tensor(np.random.RandomState(0).uniform(cfg["bounds"][:, 0], cfg["bounds"][:, 1], (16, cfg["d"])), device="cuda")
    for env in ("1", "0"):
        os.environ["ALABI_PV_SMALL"] = env; os.environ["ALABI_PV_W"] = env      # "0": the substitution kernel (no cached L^-1)
        for M in (1, 16):
            r = gp.predict_device(y, Xs[:M], return_var=True); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50): r = gp.predict_device(y, Xs[:M], return_var=True)
            torch.cuda.synchronize()
            print(name, "small" if env == "1" else "substitution", "M", M, "%.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3), "var", r[1][:2].cpu().numpy())
    os.environ["ALABI_PV_SMALL"] = "1"; os.environ["ALABI_PV_W"] = "1"; a = gp.predict_device(y, Xs, return_var=True)[1].cpu().numpy()
    os.environ["ALABI_PV_SMALL"] = "0"; os.environ["ALABI_PV_W"] = "0"; b = gp.predict_device(y, Xs, return_var=True)[1].cpu().numpy()
    print(name, "max |var_small - var_substitution| / amp = %.2e" % (np.max(np.abs(a - b)) / np.exp(h["log_amp"])))
