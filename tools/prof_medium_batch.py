"""Mean + variance for 17 ... 2048 queries: groups of 16 through predict_var_small_kernel (ALABI_PV_SMALL_MAX, default 512)
against the K* pre-pass + tile product (ALABI_PV_SMALL_MAX=16)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config
for name in (sys.argv[1:] or ["C3", "C4"]):
    cfg = make_config(name); h = cfg["hyper"]
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    y = torch.as_tensor(cfg["y"], device="cuda")
    Xs = torch.as_tensor(np.random.RandomState(0).uniform(cfg["bounds"][:, 0], cfg["bounds"][:, 1], (2048, cfg["d"])), device="cuda")
    for M in (17, 64, 128, 256, 512, 1024, 2048):
        row = []
        for mx in ("16", "4096"):
            os.environ["ALABI_PV_SMALL_MAX"] = mx
            r = gp.predict_device(y, Xs[:M], return_var=True); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30): r = gp.predict_device(y, Xs[:M], return_var=True)
            torch.cuda.synchronize(); row.append(((time.perf_counter() - t0) / 30 * 1e3, r[1].cpu().numpy()))
        print(f"{name} M={M}: tiles {row[0][0]:.3f} ms, groups of 16 {row[1][0]:.3f} ms, max |dvar|/amp {np.max(np.abs(row[0][1]-row[1][1]))/np.exp(h['log_amp']):.1e}", flush=True)
