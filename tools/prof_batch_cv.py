"""Batched fit (alabi_gp_batch_fit_predict) at the shape of init_gp(hyperopt_method="cv") on C3: B jobs of 0.8 N = 1600 rows, d = 10.
Prints ms per call, fits/s and the aggregate factorisation rate (N^3/3 flops per job) for a sweep of the queue's knobs, then
init_gp(cv) / init_gp(ml) end to end.  usage: python tools/prof_batch_cv.py [B] [sweep|-] [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from alabi_amd.gp_batch import HipGPBatch
from alabi_amd.workloads import make_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
sweep = len(sys.argv) > 2 and sys.argv[2] == "sweep"
cfg_name = sys.argv[3] if len(sys.argv) > 3 else "C3"
cfg = make_config(cfg_name.rstrip("x"))
X, y, h, d = cfg["X"], cfg["y"], cfg["hyper"], cfg["d"]
n = len(X)
rng = np.random.RandomState(0)
hyper, train, val = [], [], []
for b in range(B):
    if b % 5 == 0:
        folds = np.array_split(rng.permutation(n), 5)
        row = np.r_[h["mean"], -12.0 + rng.uniform(-1, 1), h["log_amp"] + 0.2 * rng.randn(), 1.0, h["log_M"] + 0.3 * rng.randn(d)]
    k = b % 5
    val.append(np.sort(folds[k])); train.append(np.sort(np.concatenate([folds[q] for q in range(5) if q != k]))); hyper.append(row)
hyper = np.array(hyper)
Xd, yd = torch.as_tensor(X, device="cuda"), torch.as_tensor(y, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
bt = HipGPBatch(d)
bt.fit_predict(Xd, yd, hyper[:5], train[:5], val[:5]); torch.cuda.synchronize(); t1 = time.perf_counter()
bt.fit_predict(Xd, yd, hyper, train, val); torch.cuda.synchronize(); t2 = time.perf_counter()
bt.fit_predict(Xd, yd, hyper, train, val); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"first call with 5 jobs (handle, module load) {1e3 * (t1 - t0):.1f} ms; first call with {B} jobs (workspace, task list) {1e3 * (t2 - t1):.1f} ms; second {1e3 * (t3 - t2):.1f} ms", flush=True)
N = len(train[0])
flops = B * N ** 3 / 3.0


def run(tag, reps=3):
    bt.fit_predict(Xd, yd, hyper, train, val)                      # warm-up (task list, workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ll, st, mu, off = bt.fit_predict(Xd, yd, hyper, train, val)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{tag:34s} B={B} N={N}: {dt * 1e3:8.2f} ms per call = {dt / B * 1e6:7.1f} us per fit, {flops / dt / 1e12:6.2f} TFLOP/s "
          f"(factorisations only), ok {int(np.sum(st == 0))}/{B}, timeouts {bt.timeouts}", flush=True)
    return ll, mu


settings = [None]                                   # None: whatever the environment says (the library's defaults)
if sweep:
    settings = [(8, w, gk, None, "1") for gk in (6, 8, 10, 12) for w in (0, 6, 8, 12, 16)]
ref = None
for setting in settings:
    if setting is None:
        ll, mu = run("environment / defaults")
        ref = (ll.copy(), mu.clone())
        continue
    lists, window, gk, near, u4 = setting
    os.environ["ALABI_BATCH_LISTS"], os.environ["ALABI_BATCH_WINDOW"] = str(lists), str(window)
    os.environ["ALABI_BATCH_LEFT"] = "0" if gk == "left0" else "1"
    for k, v in (("ALABI_BATCH_GK", gk), ("ALABI_CHOL_NEAR", near), ("ALABI_CHOL_UPDATE4", u4)):
        if v is None or v == "left0":
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    ll, mu = run(f"lists={lists} window={window} gk={gk} near={near} update4={u4}")
    if ref is None:
        ref = (ll.copy(), mu.clone())
    else:
        assert np.array_equal(ll, ref[0]) and torch.equal(mu, ref[1]), "results depend on the queue order"
if sweep:
    for k in ("ALABI_BATCH_LISTS", "ALABI_BATCH_WINDOW", "ALABI_BATCH_GK", "ALABI_BATCH_LEFT", "ALABI_CHOL_NEAR", "ALABI_CHOL_UPDATE4"):
        os.environ.pop(k, None)
if sweep or cfg_name != "C3":
    sys.exit(0)
os.environ["ALABI_BATCH_QUEUE"] = "0"
if B <= 100:
    run("launch-per-step fallback", reps=1)
os.environ.pop("ALABI_BATCH_QUEUE")
bt.close()

from sklearn.preprocessing import StandardScaler
from alabi_amd import SurrogateModel
for method in ("cv", "ml"):
    for rep in range(2):
        sm = SurrogateModel(lnlike_fn=cfg["fn"], bounds=cfg["bounds"], savedir="/tmp/alabi_prof_cv", verbose=False, random_state=0, cache=False)
        sm.init_samples(ntrain=2000, ntest=200)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sm.init_gp(hyperopt_method=method, y_scaler=StandardScaler())
        torch.cuda.synchronize(); t1 = time.perf_counter()
        mu = np.asarray(sm.surrogate_log_likelihood(sm.theta_test)).ravel()
        print(f"init_gp({method}) at C3 (N=2000, d=10), run {rep}: {t1 - t0:.3f} s; test MSE / var(y) = "
              f"{np.mean((mu - np.asarray(sm.y_test).ravel()) ** 2) / np.var(sm.y_test):.3e}; hyper {np.round(sm.gp.get_parameter_vector()[:4], 4)}", flush=True)
