"""Soak of the one-launch task-queue Cholesky (chain-bound sizes from N = 1000 and the sizes that use grouped updates with ordinary (L2-cached) operand loads
behind an acquire: repeated factorisations of DIFFERENT matrices, |L L^T - K| <= 1e-12 |K| checked on the device every time
(a stale operand tile would show up as an O(1) residual), while a second stream keeps the chip's caches busy with unrelated
traffic (uneven load).  python tools/soak_cholesky.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
side = torch.cuda.Stream()
junk = torch.empty(64 << 20, dtype=torch.float64, device="cuda")          # 512 MB streamed on the side
worst = {}
t0 = time.perf_counter()
paths = {}
for N, d in ((1000, 4), (1600, 10), (2000, 10), (2500, 6), (3333, 8), (4096, 10), (5000, 10), (8192, 10), (10000, 20)):
    log_M = np.log(np.full(d, 30.0 if d <= 10 else 60.0))
    gp = HipGP(d, 0.0, -12.0, 0.0, log_M)
    isd = torch.as_tensor(np.exp(-0.5 * log_M), device="cuda")
    w = 0.0
    for r in range(reps if N <= 5000 else max(reps // 3, 4)):
        X = np.random.RandomState(1000 * N + r).uniform(-3, 3, (N, d))
        with torch.cuda.stream(side):
            junk.mul_(1.0000001)                                           # unrelated HBM / L2 traffic while the queue runs
        gp.compute(X)
        paths[gp.solver.factor_path] = paths.get(gp.solver.factor_path, 0) + 1
        if N <= 2000 and r % 8:                                            # (the small sizes mostly for the path count: the residual every 8th time)
            continue
        L = gp.solver.get_factor()
        Xs = torch.as_tensor(X, device="cuda") * isd
        r2 = torch.cdist(Xs, Xs).pow_(2)
        K = torch.exp(-0.5 * r2); K.diagonal().add_(np.exp(-12.0))
        R = L @ L.T - K
        w = max(w, float(R.abs().max() / K.abs().max()))
        del R, K, r2, L
    worst[N] = w
    print(f"N={N}: worst |L L^T - K| / |K| over the repetitions: {w:.2e}", flush=True)
    assert w <= 1e-11, (N, w)
    del gp
print("soak ok in %.1f s" % (time.perf_counter() - t0), worst, "factorisation paths:", paths)
assert set(paths) == {"queue"}, paths                                      # no wait of the task queue ever ran out
