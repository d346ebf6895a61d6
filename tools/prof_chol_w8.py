"""Task-queue Cholesky: four waves per workgroup (chol_tasks_kernel) against eight (chol_tasks8_kernel), wall time incl. assembly.
  python tools/prof_chol_duo.py N [N ...]     (ALABI_CHOL_GK / ALABI_CHOL_NEAR via CHOL_SHAPES="gk,near gk,near")"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP

sizes = [int(a) for a in sys.argv[1:]] or [2000, 5000, 10000]
shapes = [tuple(a.split(',')) for a in os.environ.get('CHOL_SHAPES', '').split()] or [(None, None)]
for N in sizes:
    d = 10 if N < 8000 else 20
    rng = np.random.RandomState(N)
    X = rng.uniform(-3, 3, (N, d))
    log_M = np.log(np.full(d, 30.0 if d == 10 else 60.0))
    ref = None
    for gk, near in shapes:
        for duo in ("0", "1"):
            os.environ["ALABI_CHOL_TASKS"] = "1"; os.environ["ALABI_CHOL_W8"] = duo
            for k, v in (("ALABI_CHOL_GK", gk), ("ALABI_CHOL_NEAR", near)):
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
            gp = HipGP(d, 0.0, -12.0, 0.0, log_M)
            gp.compute(X); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); gp.compute(X); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            ld = gp.solver.log_determinant
            if ref is None: ref = ld
            print(f"N={N:6d} gk={gk} near={near} w8={duo}: {best*1e3:8.3f} ms  {N**3/3/best/1e12:6.2f} TFLOP/s  logdet diff {abs(ld-ref)/abs(ref):.1e}", flush=True)
            del gp
