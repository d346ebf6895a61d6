"""K assembly + Cholesky (HipGP.compute, default path) at a list of sizes: best of 6 calls, TFLOP/s of N^3/3, and (N <= 5000) the
distance to the launch-per-step factor.   python tools/prof_chol_sizes.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP
sizes = [int(a) for a in sys.argv[1:]] or [1024, 1600, 2000, 3072, 4096, 5000, 8192, 10000, 16000]
for N in sizes:
    d = 10 if N < 8000 else 20
    X = np.random.RandomState(N).uniform(-3, 3, (N, d)); log_M = np.log(np.full(d, 30.0 if d == 10 else 60.0))
    gp = HipGP(d, 0.0, -12.0, 0.0, log_M); gp.compute(X); torch.cuda.synchronize()
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter(); gp.compute(X); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    res = ""
    if N <= 5000:
        L = gp.solver.get_factor(); K = L @ L.T
        gp2 = HipGP(d, 0.0, -12.0, 0.0, log_M); os.environ["ALABI_CHOL_TASKS"] = "0"; gp2.compute(X); os.environ.pop("ALABI_CHOL_TASKS")
        L2 = gp2.solver.get_factor(); K2 = L2 @ L2.T
        res = f"  |LL^T - L2L2^T|/|K| {float((K - K2).abs().max() / K2.abs().max()):.1e}  |L - L2|/|L| {float((L - L2).abs().max() / L2.abs().max()):.1e} ({gp2.solver.factor_path})"
    print(f"N={N:6d} {gp.solver.factor_path:6s}: {best*1e3:8.3f} ms  {N**3/3/best/1e12:6.2f} TFLOP/s{res}", flush=True)
    del gp
