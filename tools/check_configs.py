"""Parity + timing at BASELINE.json's larger configs (C2, C4, C5) against the CPU oracle on a subset of points.
Run on the GPU box: PYTHONPATH=. python tools/check_configs.py [C2 C4 C5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP, EnsembleSampler
from alabi_amd.utility import utility_scan
from alabi_amd.workloads import make_config
from oracle.gp_oracle import OracleGP
from oracle.utility_oracle import utility_batch

for name in (sys.argv[1:] or ["C2", "C4", "C5"]):
    cfg = make_config(name); h = cfg["hyper"]; d, N, W = cfg["d"], cfg["N"], cfg["W"]
    gp = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    torch.cuda.synchronize(); t0 = time.perf_counter(); gp.compute(cfg["X"]); torch.cuda.synchronize(); t_fit = time.perf_counter() - t0
    t0 = time.perf_counter(); gp.compute(cfg["X"]); torch.cuda.synchronize(); t_fit2 = time.perf_counter() - t0
    y = torch.as_tensor(cfg["y"], device="cuda")
    rng = np.random.RandomState(0)
    Xs = rng.uniform(cfg["bounds"][:, 0], cfg["bounds"][:, 1], (300, d))
    mu, var = gp.predict(cfg["y"], Xs, return_var=True)
    t0 = time.perf_counter(); o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"]); t_cpu = time.perf_counter() - t0
    mu_o, var_o = o.predict(cfg["y"], Xs, return_var=True)
    amp = np.exp(h["log_amp"])
    print(f"{name}: N={N} d={d}  fit {t_fit2*1e3:.2f} ms (first {t_fit*1e3:.1f}; CPU oracle {t_cpu*1e3:.0f} ms)  "
          f"max|dmu|/(|mu|+1) {np.max(np.abs(mu-mu_o)/(np.abs(mu_o)+1)):.2e}  max|dvar|/amp {np.max(np.abs(var-var_o))/amp:.2e}  "
          f"logL rel {abs(gp.log_likelihood(cfg['y'])-o.log_likelihood(cfg['y']))/abs(o.log_likelihood(cfg['y'])):.1e}")
    M = 100_000 if name != "C5" else 1_000_000
    gen = torch.Generator(device="cuda"); gen.manual_seed(6)
    lo = torch.as_tensor(cfg["bounds"][:, 0], device="cuda"); hi = torch.as_tensor(cfg["bounds"][:, 1], device="cuda")
    cand = lo + (hi - lo) * torch.rand((M, d), dtype=torch.float64, device="cuda", generator=gen)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    utility_scan(gp, y, cand, cfg["bounds"], "bape"); torch.cuda.synchronize(); t_first = time.perf_counter() - t0   # workspace + L^-1 cache
    t0 = time.perf_counter(); best, val, idx = utility_scan(gp, y, cand, cfg["bounds"], "bape"); torch.cuda.synchronize(); t_scan = time.perf_counter() - t0
    sub = cand[idx:idx + 1].cpu().numpy()
    m1, v1 = o.predict(cfg["y"], sub, return_var=True)
    print(f"    BAPE scan over {M} candidates: {t_scan*1e3:.1f} ms [first call incl. workspace allocation and the L^-1 cache {t_first*1e3:.1f} ms] ({M/t_scan:.3g} cand/s, {M*N*N/t_scan/1e12:.1f} TFLOP/s); "
          f"best u {val:.6g} vs oracle at that point {utility_batch('bape', m1, v1, sub, cfg['bounds'])[0]:.6g}")
    s = EnsembleSampler(W, d, gp, y, cfg["bounds"], seed=3)
    s.run_mcmc(cfg["p0"], 600, store=False); nst = 2000
    t0 = time.perf_counter(); s.run_mcmc(None, nst); dt = time.perf_counter() - t0
    print(f"    ensemble W={W}: {W*nst/dt:.3g} samples/s ({dt/nst*1e6:.1f} us/step), acceptance {s.acceptance_fraction.mean():.2f}")
    del gp, s, o
    torch.cuda.empty_cache()
