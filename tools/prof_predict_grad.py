"""Latency of alabi_gp_predict_grad (mu, var and their gradients in the query point) at C3 / C4, and the cost of one
reference-shaped gradient (utility.py:511-623: 2d kernel rows + explicit K^-1) in the oracle for comparison."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.workloads import make_config
for name in ("C3", "C4"):
    cfg = make_config(name); h = cfg["hyper"]
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    y = torch.as_tensor(cfg["y"], device="cuda")
    Xs = torch.as_tensor(np.random.RandomState(0).uniform(cfg["bounds"][:, 0], cfg["bounds"][:, 1], (64, cfg["d"])), device="cuda")
    for M in (1, 16, 64):
        r = gp.predict_grad_device(y, Xs[:M]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): r = gp.predict_grad_device(y, Xs[:M])
        torch.cuda.synchronize()
        print(name, "predict_grad M", M, "%.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    from oracle.gp_oracle import OracleGP
    from oracle import utility_oracle as uo
    cfg = make_config("C3"); h = cfg["hyper"]
    o = OracleGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"]); o._compute_alpha(cfg["y"])
    t0 = time.perf_counter(); uo.grad_gp_var_prediction(cfg["X"][0] + 0.01, o); print("C3 oracle reference-shaped grad var: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
# one L-BFGS-B polish (30 iterations at most) of a scan incumbent at C3: value + gradient from one device call per evaluation
from alabi_amd import utility as ut
cfg = make_config("C3"); h = cfg["hyper"]
gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
rng = np.random.RandomState(1)
cand = rng.uniform(cfg["bounds"][:, 0], cfg["bounds"][:, 1], (32768, cfg["d"]))
th, u0, _ = ut.utility_scan(gp, cfg["y"], cand, cfg["bounds"], algorithm="bape")
ut.polish_point(gp, cfg["y"], th, cfg["bounds"], "bape", maxiter=30)
t0 = time.perf_counter(); n = 5
for _ in range(n): x, u = ut.polish_point(gp, cfg["y"], th, cfg["bounds"], "bape", maxiter=30)
print("C3 polish from the best of 32768 candidates: %.2f ms, bape %.6f -> %.6f" % ((time.perf_counter() - t0) / n * 1e3, u0, u))
