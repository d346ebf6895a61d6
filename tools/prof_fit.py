"""rocprofv3 driver: K assembly + Cholesky + alpha at a BASELINE config.  python tools/prof_fit.py [C3] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config
cfg = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3"); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
X = torch.as_tensor(cfg["X"], device="cuda"); y = torch.as_tensor(cfg["y"], device="cuda")
gp.compute(X); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    gp.compute(X)
torch.cuda.synchronize(); t_fit = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps):
    gp._y_set = False; gp.predict_device(y, X[:1])
torch.cuda.synchronize(); t_alpha = (time.perf_counter() - t0) / reps
N = cfg["N"]
print(f"{cfg['name']}: N={N} assemble+Cholesky {t_fit*1e3:.3f} ms ({N**3/3/t_fit/1e12:.2f} TFLOP/s), alpha {t_alpha*1e3:.3f} ms")
