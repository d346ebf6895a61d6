"""Does a latency-bound launch chain (Cholesky, alpha solve at N=2000) run faster when the rest of the chip is busy?
(Question: are single-workgroup kernels slow because the idle chip sits at a low clock?)"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config
cfg = make_config("C3"); h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
X = torch.as_tensor(cfg["X"], device="cuda"); y = torch.as_tensor(cfg["y"], device="cuda")
def fit(reps=20):
    gp.compute(X); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): gp.compute(X)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def alpha(reps=20):
    gp._y_set = False; gp.predict_device(y, X[:1]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): gp._y_set = False; gp.predict_device(y, X[:1])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("idle chip : fit %.3f ms, alpha %.3f ms" % (fit(), alpha()))
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
for frac in (1,):
    with torch.cuda.stream(side):
        for _ in range(400): c = a @ b                       # ~0.4 s of background matmuls on the side stream
    tf, ta = fit(), alpha()
    side.synchronize()
    print("busy chip : fit %.3f ms, alpha %.3f ms (fp32 GEMMs running on another stream)" % (tf, ta))
print("idle again: fit %.3f ms, alpha %.3f ms" % (fit(), alpha()))
