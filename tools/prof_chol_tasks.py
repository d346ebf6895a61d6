"""One-launch task-queue Cholesky: time by (block columns per far update, width of the near band) against the default path.
  python tools/prof_chol_tasks.py N [N ...]   (ALABI_CHOL_GK / ALABI_CHOL_NEAR are set per variant)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP

sizes = [int(a) for a in sys.argv[1:]] or [2000, 5000, 10000]
for N in sizes:
    d = 10 if N < 8000 else 20
    rng = np.random.RandomState(N)
    X = rng.uniform(-3, 3, (N, d))
    log_M = np.log(np.full(d, 30.0 if d == 10 else 60.0))
    variants = [("default path", None, None, None)] + [(f"queue gk={g} near={n}", "1", str(g), str(n))
                                                      for g, n in [tuple(int(v) for v in a.split(',')) for a in os.environ.get('CHOL_SHAPES', '1,1 4,2 8,2 16,2 16,4').split()]]
    ref = None
    for tag, tq, gk, near in variants:
        for k, v in (("ALABI_CHOL_TASKS", tq), ("ALABI_CHOL_GK", gk), ("ALABI_CHOL_NEAR", near)):
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        gp = HipGP(d, 0.0, -12.0, 0.0, log_M)
        gp.compute(X); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); gp.compute(X); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        ld = gp.solver.log_determinant
        if ref is None: ref = ld
        print(f"N={N:6d} {tag:24s}: {best*1e3:8.3f} ms  {N**3/3/best/1e12:6.2f} TFLOP/s  logdet diff {abs(ld-ref)/abs(ref):.1e}", flush=True)
        del gp
