"""Soak of the batched fit (alabi_gp_batch_fit_predict): many calls with DIFFERENT hyper-parameters, fold sets and sizes (mixed block
counts in one launch, some jobs not positive definite), under unrelated traffic on a second stream; after every call a sample of
jobs is compared with the single-matrix path (factor bit for bit, log-likelihood, held-out mean) and the queue must never time out.
python tools/soak_batch.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.gp_batch import HipGPBatch
from alabi_amd.workloads import make_config

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = make_config("C3")
X, y, h, d = cfg["X"], cfg["y"], cfg["hyper"], cfg["d"]
n = len(X)
Xd, yd = torch.as_tensor(X, device="cuda"), torch.as_tensor(y, device="cuda")
bt = HipGPBatch(d)
side = torch.cuda.Stream()
junk = torch.empty(32 << 20, dtype=torch.float64, device="cuda")
os.environ["ALABI_CHOL_TASKS"] = "1"            # the single-matrix comparison through the task queue at every size
rng = np.random.RandomState(0)
t0 = time.perf_counter(); fits = 0; npd = 0; checked = 0; worst_ll = 0.0; worst_mu = 0.0
for c in range(calls):
    B = int(rng.choice([5, 40, 125, 250, 500]))
    hyper, train, val = [], [], []
    for b in range(B):
        N = int(rng.choice([1600, 1600, 1600, 1601, 1537, 1200, 777, 300, 64, 130]))
        perm = rng.permutation(n)
        train.append(np.sort(perm[:N])); val.append(np.sort(perm[N:N + int(rng.randint(1, 400))]))
        wn = -12.0 + rng.uniform(-2, 2) if rng.rand() > 0.02 else -80.0           # now and then: no nugget (may fail)
        hyper.append(np.r_[h["mean"] + 0.2 * rng.randn(), wn, h["log_amp"] + 0.3 * rng.randn(), 1.0, h["log_M"] + 0.4 * rng.randn(d)])
    hyper = np.array(hyper)
    with torch.cuda.stream(side):
        junk.mul_(1.0000001)
    ll, st, mu, off = bt.fit_predict(Xd, yd, hyper, train, val)
    assert bt.timeouts == 0, "the batched queue timed out"
    fits += B; npd += int(np.sum(st != 0))
    mu = mu.cpu().numpy()
    for b in rng.choice(B, size=min(B, 3), replace=False):
        N = len(train[b])
        g = HipGP(d, hyper[b, 0], hyper[b, 1], hyper[b, 2], hyper[b, 4:])
        ok = g.compute(X[train[b]], quiet=True)
        assert ok == (st[b] == 0), (c, b, ok, st[b])
        if not ok:
            continue
        if N > 128:
            assert torch.equal(g.solver.get_factor(), bt.get_factor(int(b), N)), (c, b, N)
        ll_s = g.log_likelihood(y[train[b]]); mu_s = g.predict(y[train[b]], X[val[b]], return_cov=False)
        worst_ll = max(worst_ll, abs(ll[b] - ll_s) / (abs(ll_s) + 1))
        worst_mu = max(worst_mu, float(np.max(np.abs(mu[off[b]:off[b + 1]] - mu_s)) / (np.max(np.abs(mu_s - hyper[b, 0])) + 1e-300)))
        checked += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"soak ok: {calls} calls, {fits} fits ({npd} not positive definite, reported as such), {checked} jobs compared with the single-matrix "
      f"path (factors bit-identical; worst log-likelihood difference {worst_ll:.1e} relative, worst held-out mean difference {worst_mu:.1e} of its "
      f"range), queue time-outs {bt.timeouts}, {dt:.1f} s")
assert worst_ll <= 1e-9 and worst_mu <= 1e-7
