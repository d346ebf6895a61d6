"""One evaluation of the polish step (predict_grad_host: value + gradient of mean and variance at ONE point, back on the host):
microseconds per call at small and large training sets, and the pieces (set_y check, H2D of the point, launch, D2H)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP
for N, d in ((100, 2), (500, 5), (2000, 10)):
    rng = np.random.RandomState(N)
    X = rng.uniform(-3, 3, (N, d)); y = np.sin(X.sum(1))
    gp = HipGP(d, 0.0, -10.0, 0.0, np.log(np.full(d, 4.0))); gp.compute(X)
    x = rng.uniform(-1, 1, (1, d))
    for _ in range(20): gp.predict_grad_host(y, x)
    t0 = time.perf_counter()
    for _ in range(300): gp.predict_grad_host(y, x)
    t_all = (time.perf_counter() - t0) / 300
    yd = torch.as_tensor(y, device="cuda"); xd = torch.as_tensor(x, device="cuda")
    t0 = time.perf_counter()
    for _ in range(300): gp.predict_grad_device(yd, xd)
    torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / 300
    t0 = time.perf_counter()
    for _ in range(300): gp.predict_grad_device(yd, xd); torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / 300
    print(f"N={N} d={d}: predict_grad_host {t_all*1e6:.0f} us per call; device tensors in/out, no sync {t_dev*1e6:.0f} us; with a sync per call {t_sync*1e6:.0f} us", flush=True)
