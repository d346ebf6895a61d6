"""predict mean+variance: one MFMA wave per SIMD (predict_var_w_kernel) vs two (predict_var_w2_kernel), C3/C4/C5."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config
names = sys.argv[1:] or ["C3", "C4"]
for name in names:
    cfg = make_config(name); h = cfg["hyper"]; N = cfg["N"]
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    y = torch.as_tensor(cfg["y"], device="cuda")
    Ms = (256, 4096, 16384, 65536, 262144) if N <= 5000 else (65536, 262144)
    Xs = torch.as_tensor(np.random.RandomState(0).uniform(cfg["bounds"][:, 0], cfg["bounds"][:, 1], (max(Ms), cfg["d"])), device="cuda")
    os.environ["ALABI_PV_W"] = "1"
    ref = {}
    for env in ("0", "1"):
        os.environ["ALABI_PV_W2"] = env
        for M in Ms:
            r = gp.predict_device(y, Xs[:M], return_var=True); torch.cuda.synchronize()
            reps = 3 if M * N * N > 2e13 else 10
            t0 = time.perf_counter()
            for _ in range(reps): r = gp.predict_device(y, Xs[:M], return_var=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
            v = r[1].cpu().numpy()
            same = "" if env == "0" else " bit-identical to w1: %s" % np.array_equal(v, ref[M])
            if env == "0": ref[M] = v
            print(f"{name} W2={env} M={M}: {dt*1e3:.3f} ms  {M/dt:.3e} pts/s  {M*N*N/dt/1e12:.1f} TFLOP/s{same}", flush=True)
