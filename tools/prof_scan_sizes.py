"""utility_scan (mean + variance + acquisition value + arg-min) at the candidate counts the active-learning iteration uses
(16 384 first pass, 4096 per zoom stage) and around them: ms per call and TFLOP/s of the variance product (N^2 flops per candidate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP, utility as ut
from alabi_amd.workloads import make_config
for name in (sys.argv[1:] or ["C3"]):
    cfg = make_config(name); h = cfg["hyper"]; N = len(cfg["X"])
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    rng = np.random.RandomState(0)
    for M in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        cand = torch.as_tensor(rng.uniform(cfg["bounds"][:, 0], cfg["bounds"][:, 1], (M, cfg["d"])), device="cuda")
        for _ in range(3): out = ut.utility_scan(gp, cfg["y"], cand, cfg["bounds"], algorithm="bape", return_all=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): out = ut.utility_scan(gp, cfg["y"], cand, cfg["bounds"], algorithm="bape", return_all=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{name} N={N} M={M:6d}: {dt*1e3:7.3f} ms per scan, {M * float(N) ** 2 / dt / 1e12:5.1f} TFLOP/s", flush=True)
