"""Where the FIRST init_gp(hyperopt_method="cv") of a process spends its time (C3): workspace sizes, torch first-use, library
first-use.  usage: python tools/prof_init_gp_first_call.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
t00 = time.perf_counter()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print(f"torch cuda init {time.perf_counter() - t00:.3f} s")
from alabi_amd import SurrogateModel, HipGP
from alabi_amd.workloads import make_config
from alabi_amd import gp_utils, gp_batch
cfg = make_config("C3")


def stamp(label, t0):
    torch.cuda.synchronize()
    print(f"  {label:48s} {1e3 * (time.perf_counter() - t0):8.1f} ms", flush=True)
    return time.perf_counter()


for budget in (os.environ.get("ALABI_BATCH_BYTES", "default"),):
    t0 = time.perf_counter()
    g = HipGP(10); g.compute(cfg["X"]); t0 = stamp("first HipGP.compute (library load, handle)", t0)
    fo = torch.zeros((100, 2000), dtype=torch.int8, device="cuda")
    kk = torch.arange(5, dtype=torch.int8, device="cuda")[None, :, None]
    m = fo[:, None, :] == kk; nz = m.nonzero()[:, 2].to(torch.int32); t0 = stamp("torch mask / nonzero first use", t0)
    bt = gp_batch.HipGPBatch(10); t0 = stamp("HipGPBatch()", t0)
    X = torch.as_tensor(cfg["X"], device="cuda"); y = torch.as_tensor(cfg["y"], device="cuda")
    rng = np.random.RandomState(0)
    h = cfg["hyper"]; row = np.r_[h["mean"], -12.0, h["log_amp"], 1.0, h["log_M"]]
    for B in (5, 500, 500, 250):
        folds = [np.array_split(rng.permutation(2000), 5) for _ in range(B // 5)]
        tr = [np.sort(np.concatenate([f[q] for q in range(5) if q != k])) for f in folds for k in range(5)]
        va = [np.sort(f[k]) for f in folds for k in range(5)]
        t0 = time.perf_counter()
        bt.fit_predict(X, y, np.tile(row, (B, 1)), tr, va); t0 = stamp(f"fit_predict B={B}", t0)
    bt.close(); t0 = stamp("close", t0)
for rep in range(2):
    sm = SurrogateModel(lnlike_fn=cfg["fn"], bounds=cfg["bounds"], savedir="/tmp/alabi_fc", verbose=False, random_state=rep, cache=False)
    sm.init_samples(ntrain=2000)
    t0 = time.perf_counter(); sm.init_gp(hyperopt_method="cv"); stamp(f"init_gp(cv) call {rep}", t0)
