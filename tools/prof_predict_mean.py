"""Mean-only predict at C3 sizes (N = 2000, d = 10), HIP-event time per call for several batch sizes:
PYTHONPATH=. python tools/prof_predict_mean.py [M ...]   (ALABI_PM_MFMA=0 selects the vector kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config

sizes = [int(a) for a in sys.argv[1:]] or [65536, 1_000_000, 4_000_000]
cfg = make_config(os.environ.get("CFG", "C3"))
h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
gp.compute(cfg["X"])
y = torch.as_tensor(cfg["y"], device="cuda")
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
lo = torch.as_tensor(cfg["bounds"][:, 0], device="cuda"); hi = torch.as_tensor(cfg["bounds"][:, 1], device="cuda")
for M in sizes:
    Xs = lo + (hi - lo) * torch.rand((M, cfg["d"]), dtype=torch.float64, device="cuda", generator=gen)
    for _ in range(3):
        gp.predict_device(y, Xs)
    torch.cuda.synchronize()
    reps = 10
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gp.predict_device(y, Xs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"predict mean: M={M} N={cfg['N']} d={cfg['d']}: {ms:.3f} ms per call, {M / ms * 1e3:.4g} pts/s", flush=True)
