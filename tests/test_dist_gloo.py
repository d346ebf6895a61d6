"""world_size-2 (and 3) gloo runs on CPU: the walker-sharded ensemble (alabi_amd/dist.py) must produce the SAME
chain on every rank as the single-process oracle run -- the partition / all-gather logic is what is under test, so
the compute backend is an oracle-backed stand-in with the HipBackend interface."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Same four methods as alabi_amd.dist.HipBackend, on NumPy (TEST stand-in, lives in tests/ only)."""

    def __init__(self, W, d, seed, lnprob_batch):
        from oracle import stretch_oracle as so
        self.so = so
        self.W, self.d, self.seed, self.lnp = W, d, seed, lnprob_batch
        self.chunk = 7          # deliberately not a divisor of nsteps
        self._draws = []

    def lnprob(self, coords):
        return torch.as_tensor(self.lnp(coords.numpy()))

    def draw(self, step0, n, a):
        self._draws = [self.so.draw_step_randoms(self.seed, step0 + t, self.W) for t in range(n)]

    def order(self, t):
        order, n0 = self._draws[t][0], self._draws[t][1]
        return torch.as_tensor(order.astype(np.int64)), n0

    def half_step(self, coords, logp, t, split, begin, end, a, n_accept):
        order, n0, u_z, partner, u_acc = self._draws[t]
        S = order[:n0] if split == 0 else order[n0:]
        C = order[n0:] if split == 0 else order[:n0]
        c = coords.numpy(); lp = logp.numpy(); na = n_accept.numpy()
        mine = S[begin:end]
        if len(mine) == 0:
            return
        cp = c[C[partner[mine]]]
        zz = ((a - 1.0) * u_z[mine] + 1.0) ** 2.0 / a
        q = cp - (cp - c[mine]) * zz[:, None]
        new = self.lnp(q)
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = (self.d - 1.0) * np.log(zz) + new - lp[mine] > np.log(u_acc[mine])
        c[mine[acc]] = q[acc]; lp[mine[acc]] = new[acc]; na[mine[acc]] += 1


def _lnp(q):
    P = np.array([[2.0, 0.3, 0.0], [0.3, 1.0, -0.2], [0.0, -0.2, 0.5]])
    inside = np.all((q > -4.0) & (q < 4.0), axis=1)
    return np.where(inside, -0.5 * np.einsum("ni,ij,nj->n", q, P, q), -np.inf)


def _worker(rank, world, port, W, nsteps, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alabi_amd.dist import ShardedEnsemble
    p0 = np.random.RandomState(5).uniform(-2, 2, (W, 3))
    ens = ShardedEnsemble(OracleBackend(W, 3, 99, _lnp))
    chain, coords, logp, nacc = ens.run(torch.as_tensor(p0), nsteps, step0=3, thin_by=2)
    out[rank] = (chain.numpy(), coords.numpy(), logp.numpy(), nacc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,W", [(2, 16), (2, 13), (3, 20)])
def test_sharded_ensemble_matches_single_process(world, W):
    from oracle import stretch_oracle as so
    nsteps = 24
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), W, nsteps, out), nprocs=world, join=True)
    p0 = np.random.RandomState(5).uniform(-2, 2, (W, 3))
    chain, _, nacc, coords, logp = so.run_ensemble(p0, nsteps, _lnp, seed=99, thin_by=2, step0=3)
    for r in range(world):
        c_r, co_r, lp_r, na_r = out[r]
        assert np.array_equal(c_r, chain), f"rank {r}: chain differs from the single-process run"
        assert np.array_equal(co_r, coords) and np.array_equal(lp_r, logp)
        assert np.array_equal(na_r, nacc)


def test_slice_bounds_cover_everything():
    from alabi_amd.dist import slice_bounds
    for n in (0, 1, 7, 128, 129):
        for world in (1, 2, 3, 8):
            cuts = [slice_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in cuts]
            assert max(sizes) - min(sizes) <= 1


def _scan_worker(rank, world, port, M, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alabi_amd.dist import sharded_utility_scan
    from oracle.utility_oracle import utility_batch
    cand, mu, var, bounds = _scan_problem(M)

    def scan(b, e):                                  # stand-in for utility_scan on this rank's slice (NumPy oracle)
        u = utility_batch("bape", mu[b:e], var[b:e], cand[b:e], bounds)
        fin = np.isfinite(u)
        if not fin.any():
            return np.inf, -1
        i = int(np.flatnonzero(fin)[np.argmin(u[fin])])
        return float(u[i]), i

    out[rank] = sharded_utility_scan(scan, M)
    dist.barrier()
    dist.destroy_process_group()


def _scan_problem(M):
    rs = np.random.RandomState(11)
    cand = rs.uniform(-1.2, 1.2, (M, 3))             # some outside the unit box -> +inf utility
    mu = rs.normal(size=M)
    var = rs.uniform(0.01, 2.0, M)
    if M > 40:
        cand[:M // 3] = 5.0                          # the whole first slice of a 3-rank run is outside the box
        mu[M // 2] = mu[M - 2] = 50.0; var[M // 2] = var[M - 2] = 1.0; cand[M // 2] = cand[M - 2] = 0.1   # an exact tie across ranks
    return cand, mu, var, np.array([[-1.0, 1.0]] * 3)


@pytest.mark.parametrize("world,M", [(2, 1001), (3, 90), (2, 1)])
def test_sharded_candidate_scan_matches_single_process(world, M):
    """C5's candidate-scan shard: partition M over the ranks, all-reduce ONE (value, global index) pair; every rank ends
    with the arg-min of the full scan (ties -> smallest index; a rank without finite candidates contributes nothing)."""
    from oracle.utility_oracle import utility_batch
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_scan_worker, args=(world, _free_port(), M, out), nprocs=world, join=True)
    cand, mu, var, bounds = _scan_problem(M)
    u = utility_batch("bape", mu, var, cand, bounds)
    fin = np.isfinite(u)
    ref_i = int(np.flatnonzero(fin)[np.argmin(u[fin])]) if fin.any() else -1
    cover = []
    for r in range(world):
        v, i, (b, e) = out[r]
        assert i == ref_i and (i < 0 or v == u[ref_i]), (r, v, i, ref_i)
        cover.append((b, e))
    assert cover[0][0] == 0 and cover[-1][1] == M and all(cover[k][1] == cover[k + 1][0] for k in range(world - 1))


def _helpers_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alabi_amd import dist as adist
    assert adist.world_info() == (rank, world)
    # the CV search over several ranks: every rank scores candidates rank, rank + world, ... (+inf elsewhere), one MIN all-reduce
    full = np.array([3.0, np.inf, 1.5, 2.5, 0.25, np.inf, 7.0])
    mine = np.where(np.arange(len(full)) % world == rank, full, np.inf)
    red = adist.allreduce_array(mine, "min")
    tot = adist.allreduce_array([float(rank), 1.0], "sum")
    # the best `ntop` candidates of every slice, gathered in rank order
    rows = np.array([[10.0 * rank + k, rank * 100 + k] for k in range(3)])
    allrows = adist.allgather_rows(rows)
    # replica samples of different lengths
    samples = adist.gather_replicas(np.full((2 + rank, 2), float(rank)))
    out[rank] = (red, tot, allrows, samples)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_collective_helpers_of_the_surrogate_model_paths(world):
    """alabi_amd.dist.allreduce_array / allgather_rows / gather_replicas: what SurrogateModel.init_gp(cv), find_next_point and
    run_emcee exchange between ranks (score vectors, top candidates, replica samples) -- identical on every rank."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_helpers_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    full = np.array([3.0, np.inf, 1.5, 2.5, 0.25, np.inf, 7.0])
    for r in range(world):
        red, tot, allrows, samples = out[r]
        np.testing.assert_array_equal(red, full)
        np.testing.assert_array_equal(tot, [sum(range(world)), world])
        want = np.array([[10.0 * q + k, q * 100 + k] for q in range(world) for k in range(3)])
        np.testing.assert_array_equal(allrows, want)
        np.testing.assert_array_equal(samples, np.vstack([np.full((2 + q, 2), float(q)) for q in range(world)]))


def test_world_info_without_a_process_group():
    from alabi_amd import dist as adist
    assert adist.world_info() == (0, 1)
    np.testing.assert_array_equal(adist.allreduce_array([1.0, 2.0], "min"), [1.0, 2.0])
    np.testing.assert_array_equal(adist.allgather_rows(np.ones((2, 3))), np.ones((2, 3)))
