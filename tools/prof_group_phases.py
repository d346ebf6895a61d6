"""Phase breakdown of the group ensemble kernel (needs a build with EXTRA=-DALABI_GROUP_PROF: see tools/README.md).
  python tools/prof_group_phases.py [C4|C5] [N]"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import EnsembleSampler, HipGP, _lib
from alabi_amd.workloads import make_config
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg = make_config(name, N=int(sys.argv[2]) if len(sys.argv) > 2 else None)
h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
os.environ["ALABI_ENS_GROUP"] = "1"
s = EnsembleSampler(cfg["W"], cfg["d"], gp, cfg["y"], cfg["bounds"], seed=1)
s.run_mcmc(cfg["p0"], 1024); torch.cuda.synchronize()
t0 = time.perf_counter(); s.run_mcmc(None, 1024); torch.cuda.synchronize(); dt = time.perf_counter() - t0
out = (ctypes.c_longlong * 16)()
L = _lib.lib()
L.alabi_debug_group_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
print("rc", L.alabi_debug_group_prof(out), name, "N", cfg["N"], "path", s.last_path, "wall us/half-step %.3f" % (1e6 * dt / 2048))
names = ["rows phase (poll + accept tests)", "proposals + barrier A", "kernel sums", "barrier B + partial stores"]
for off, who in ((0, "first-proposal wave"), (8, "wave 0 (publisher)")):
    v = list(out)[off:off + 7]
    n = max(v[4], 1)
    print(who, " ".join(f"| {nm}: {0.01 * x / n:.3f} us" for nm, x in zip(names, v[:4])), f"| sum {0.01 * sum(v[:4]) / n:.3f} us",
          f"| rows phase = set-up {0.01 * v[5] / n:.3f} + polling {0.01 * v[6] / n:.3f} + accept tests {0.01 * (v[0] - v[5] - v[6]) / n:.3f}")
