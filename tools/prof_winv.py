"""Building W = L^-1: recursive block inversion (gp_inverse.hip) against the substitution chains (ALABI_WINV_DNC=0), and the
difference of the variances they give."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config
for name in (sys.argv[1:] or ["C2", "C3", "C4"]):
    cfg = make_config(name); h = cfg["hyper"]
    y = torch.as_tensor(cfg["y"], device="cuda")
    Xs = torch.as_tensor(np.random.RandomState(0).uniform(cfg["bounds"][:, 0], cfg["bounds"][:, 1], (4096, cfg["d"])), device="cuda")
    res = {}
    for env in ("0", "1"):
        os.environ["ALABI_WINV_DNC"] = env; os.environ["ALABI_PV_W"] = "1"
        gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
        gp.predict_device(y, Xs[:16], return_var=True); torch.cuda.synchronize()      # builds W once (allocations)
        ts = []
        for _ in range(5):
            gp.compute(cfg["X"]); gp.predict_device(y, Xs[:1]); torch.cuda.synchronize()
            t0 = time.perf_counter(); gp.predict_device(y, Xs[:16], return_var=True); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[env] = gp.predict_device(y, Xs, return_var=True)[1].cpu().numpy()
        print(f"{name} N={cfg['N']} DNC={env}: first 16-query variance after a factorisation (W build + product) {min(ts)*1e3:.3f} ms", flush=True)
    amp = np.exp(h["log_amp"])
    os.environ["ALABI_PV_W"] = "0"
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    sub = gp.predict_device(y, Xs, return_var=True)[1].cpu().numpy()
    print(f"{name}: max |var_dnc - var_chain| / amp = {np.max(np.abs(res['1'] - res['0'])) / amp:.2e};  vs substitution kernel: dnc {np.max(np.abs(res['1'] - sub)) / amp:.2e}, chain {np.max(np.abs(res['0'] - sub)) / amp:.2e}")
