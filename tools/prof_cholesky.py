"""Cholesky (assembly + factorisation) time at several sizes, rank-64 path against the panel path with and without look-ahead.
Run on the GPU box: PYTHONPATH=. python tools/prof_cholesky.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP

sizes = [int(a) for a in sys.argv[1:]] or [2000, 3072, 5000, 10000]
for N in sizes:
    d = 10 if N < 8000 else 20
    rng = np.random.RandomState(N)
    X = rng.uniform(-3, 3, (N, d))
    log_M = np.log(np.full(d, 30.0 if d == 10 else 60.0))
    variants = [("task queue (one launch)", "0", "1", "1")] if N <= 8192 else []
    variants += [("rank-64", "0", "1", "0"), ("panel-4 serial", "4", "0", "0"), ("panel-4 look-ahead", "4", "1", "0"), ("panel-2 look-ahead", "2", "1", "0")] + ([("panel-6 look-ahead", "6", "1", "0"), ("panel-8 look-ahead", "8", "1", "0")] if N >= 5500 else [])
    for tag, panel, la, tq in variants:
        os.environ["ALABI_CHOL_PANEL"] = panel; os.environ["ALABI_CHOL_LOOKAHEAD"] = la; os.environ["ALABI_CHOL_TASKS"] = tq
        gp = HipGP(d, 0.0, -12.0, 0.0, log_M)
        gp.compute(X); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); gp.compute(X); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"N={N:6d} {tag:24s}: {best*1e3:8.3f} ms  {N**3/3/best/1e12:6.2f} TFLOP/s", flush=True)
        del gp
