"""get_autocorr_time(tol=0) on a device chain: the library's transforms (alabi_chain_autocorr) against torch.fft (rocFFT: run-time
kernel compilation for every new length), first and second call, over chain lengths.  python tools/prof_autocorr.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd.mcmc_utils import integrated_time
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for n_t, n_w, n_d in ((20000, 256, 10), (50000, 256, 10), (50001, 256, 10), (100000, 128, 5), (13000, 128, 5)):
    x = torch.randn((n_t, n_w, n_d), dtype=torch.float64, device="cuda").cumsum(0) * 0.01 + torch.randn((n_t, n_w, n_d), dtype=torch.float64, device="cuda")
    for mode in ("1", "0"):
        os.environ["ALABI_ACF_NATIVE"] = mode
        ts = []
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter(); tau = integrated_time(x, tol=0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"n_t={n_t} n_w={n_w} n_d={n_d} ({x.numel() * 8 / 1e6:.0f} MB) {'native ' if mode == '1' else 'rocFFT '}: first call {ts[0] * 1e3:8.1f} ms, second {ts[1] * 1e3:7.1f} ms, tau[0] {tau[0]:.3f}", flush=True)
    del x
