"""Phase breakdown of the persistent ensemble kernel (needs a build with -DALABI_STREAM_PROF: see tools/README.md)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import EnsembleSampler, HipGP, _lib
from alabi_amd.workloads import make_config
cfg = make_config("C3", N=int(os.environ.get("PROF_N", "2000")))
h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
s = EnsembleSampler(cfg["W"], cfg["d"], gp, cfg["y"], cfg["bounds"], seed=1)
s.run_mcmc(cfg["p0"], 1024); torch.cuda.synchronize()
t0 = time.perf_counter(); s.run_mcmc(None, 1024); torch.cuda.synchronize(); dt = time.perf_counter() - t0
out = (ctypes.c_longlong * 16)()
L = _lib.lib()
L.alabi_debug_stream_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
print("rc", L.alabi_debug_stream_prof(out), "path", s.last_path, "wall us/half-step", 1e6 * dt / 2048)
v = list(out)[:6]
n = max(v[4], 1)
names = ["poll wait (wave 0)", "form proposal + barrier A", "compute waves (A -> B)", "tree sum + accept + row store + loop"]
tot = sum(v[:4])
for nm, x in zip(names, v[:4]):
    print(f"{nm:36s} {x / n:9.1f} ticks/item  {100.0 * x / tot:5.1f}%")
print("items", n, "ticks total", v[5], "ticks/item", v[5] / n, "=> tick ns", 1e9 * dt / v[5])
