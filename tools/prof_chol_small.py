"""Small matrices (3 .. 15 block columns): the one-launch task queue (ALABI_CHOL_TASKS=1) against the launch-per-step path (=0), which is
the default below 16 block columns.  Best of 8 HipGP.compute calls, wall time incl. assembly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP
for N in (192, 256, 320, 384, 512, 640, 768, 896, 960):
    d = 6
    X = np.random.RandomState(N).uniform(-3, 3, (N, d)); log_M = np.log(np.full(d, 20.0))
    row = []
    for tq in ("1", "0"):
        os.environ["ALABI_CHOL_TASKS"] = tq
        gp = HipGP(d, 0.0, -12.0, 0.0, log_M); gp.compute(X); torch.cuda.synchronize()
        best = 1e9
        for _ in range(8):
            t0 = time.perf_counter(); gp.compute(X); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        row.append((best, gp.solver.factor_path))
    print(f"N={N:5d} ({(N + 63) // 64:2d} block columns): queue {row[0][0]*1e3:.3f} ms ({row[0][1]}), steps {row[1][0]*1e3:.3f} ms ({row[1][1]})", flush=True)
