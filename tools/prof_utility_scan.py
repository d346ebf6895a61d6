"""utility_scan / predict_device timings at the scan and zoom-stage batch sizes (C3)."""
import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.utility import utility_scan
from alabi_amd.workloads import make_config
cfg = make_config("C3"); h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
y = torch.as_tensor(cfg["y"], device="cuda")
lo = torch.as_tensor(cfg["bounds"][:, 0], device="cuda"); hi = torch.as_tensor(cfg["bounds"][:, 1], device="cuda")
for M in (65536, 4096):
    cand = lo + (hi - lo) * torch.rand((M, cfg["d"]), dtype=torch.float64, device="cuda")
    for ra in (False, True):
        utility_scan(gp, y, cand, cfg["bounds"], algorithm="bape", return_all=ra); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): utility_scan(gp, y, cand, cfg["bounds"], algorithm="bape", return_all=ra)
        torch.cuda.synchronize(); print("utility_scan M", M, "return_all", ra, "%.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    t0 = time.perf_counter()
    for _ in range(10): gp.predict_device(y, cand, return_var=True)
    torch.cuda.synchronize(); print("predict_device M", M, "%.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
