"""Debug helper: where do the factors of the batched queue and of the single-matrix queue differ (tile coordinates), and is each
of them reproducible from run to run?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ALABI_CHOL_TASKS"] = "1"
import numpy as np, torch
from alabi_amd import HipGP
from alabi_amd.gp_batch import HipGPBatch

n, d, N = 1800, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 1600
rng = np.random.RandomState(0)
X = rng.uniform(-3, 3, (n, d)); y = np.sin(X.sum(1))
B = 8
train = [np.sort(rng.permutation(n)[:N]) for _ in range(B)]
val = [np.arange(5) for _ in range(B)]
hyper = np.array([np.r_[0.0, -10.0, 0.1 * b, 1.0, np.log(2.0) + 0.1 * rng.randn(d)] for b in range(B)])
Xd, yd = torch.as_tensor(X, device="cuda"), torch.as_tensor(y, device="cuda")
bt = HipGPBatch(d)
def batch():
    bt.fit_predict(Xd, yd, hyper, train, val)
    return [bt.get_factor(b, N).cpu().numpy() for b in range(B)]
def single(b):
    g = HipGP(d, hyper[b, 0], hyper[b, 1], hyper[b, 2], hyper[b, 4:]); g.compute(X[train[b]])
    return g.solver.get_factor().cpu().numpy()
def first_diff(a, c):
    dd = a != c
    if not dd.any(): return None
    nb = (len(a) + 63) // 64
    t = [(i, j, int(dd[64*i:64*i+64, 64*j:64*j+64].sum())) for j in range(nb) for i in range(j, nb) if dd[64*i:64*i+64, 64*j:64*j+64].any()]
    i, j, c_ = t[0]
    blk = dd[64*i:64*i+64, 64*j:64*j+64]
    rows = np.flatnonzero(blk.any(1)); cols = np.flatnonzero(blk.any(0))
    return t[:6], len(t), (rows.min(), rows.max()), (cols.min(), cols.max()), float(np.max(np.abs(a - c)))
if __name__ == '__main__':
  b1, b2 = batch(), batch()
  s1 = [single(b) for b in range(B)]; s2 = [single(b) for b in range(B)]
  for b in range(B):
    print(b, "batch vs batch:", first_diff(b1[b], b2[b]), "| single vs single:", first_diff(s1[b], s2[b]), "| batch vs single:", first_diff(b1[b], s1[b]), flush=True)
    K = s1[b] @ s1[b].T
    print("   residual single", np.max(np.abs(K - b1[b] @ b1[b].T)) / np.max(np.abs(K)))
