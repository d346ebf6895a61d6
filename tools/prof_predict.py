"""Driver for rocprofv3 runs of the batched predict kernels (C3 sizes): python tools/prof_predict.py [M] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = make_config(os.environ.get("CFG", "C3"))
h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
gp.compute(cfg["X"])
y = torch.as_tensor(cfg["y"], device="cuda")
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
lo = torch.as_tensor(cfg["bounds"][:, 0], device="cuda"); hi = torch.as_tensor(cfg["bounds"][:, 1], device="cuda")
Xs = lo + (hi - lo) * torch.rand((M, cfg["d"]), dtype=torch.float64, device="cuda", generator=gen)
gp.predict_device(y, Xs, return_var=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    gp.predict_device(y, Xs, return_var=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
N = cfg["N"]
print(f"predict mean+var: M={M} N={N} {dt*1e3:.2f} ms  {M/dt:.3g} pts/s  {M*N*N/dt/1e12:.2f} TFLOP/s (N^2 flops per point)")
