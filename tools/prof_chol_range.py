import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import HipGP
for N in (1024, 2000, 3072, 4096, 5000, 6000, 7000, 8192):
    d = 10
    X = np.random.RandomState(N).uniform(-3, 3, (N, d)); log_M = np.log(np.full(d, 30.0))
    for tag, tq, gk, near in (("default", None, None, None), ("queue gk=4 near=2", "1", "4", "2"), ("queue gk=8 near=2", "1", "8", "2"), ("queue off", "0", None, None)):
        for k, v in (("ALABI_CHOL_TASKS", tq), ("ALABI_CHOL_GK", gk), ("ALABI_CHOL_NEAR", near)):
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        gp = HipGP(d, 0.0, -12.0, 0.0, log_M); gp.compute(X); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); gp.compute(X); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"N={N:6d} {tag:20s}: {best*1e3:8.3f} ms  {N**3/3/best/1e12:6.2f} TFLOP/s", flush=True)
        del gp
