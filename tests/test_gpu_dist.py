"""The sharded-ensemble path with the REAL HIP kernels: two ranks (both on cuda:0, gloo rendezvous because RCCL
refuses two ranks on one device) must reproduce the single-process HIP chain bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, nsteps, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import make_problem
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.dist import HipBackend, ShardedEnsemble
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, h = make_problem(300, 4, 7)
    gp = HipGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(X)
    bounds = np.array([[-3.0, 3.0]] * 4)
    sampler = EnsembleSampler(W, 4, gp, y, bounds, seed=4242)
    ens = ShardedEnsemble(HipBackend(sampler))
    p0 = np.random.RandomState(9).uniform(-2, 2, (W, 4))
    chain, coords, logp, nacc = ens.run(torch.as_tensor(p0, device="cuda"), nsteps, step0=0)
    # the same run with the whole step loop inside the library (alabi_ens_run_sharded; host callback instead of RCCL here)
    from alabi_amd.dist import ShardedRun, sharded_utility_scan
    from alabi_amd.utility import utility_scan
    run_c = ShardedRun(sampler)
    chain_c, coords_c, logp_c, nacc_c = run_c.run(torch.as_tensor(p0, device="cuda"), nsteps, step0=0, thin_by=1)
    assert torch.equal(chain_c, chain) and torch.equal(coords_c, coords) and torch.equal(logp_c, logp) and torch.equal(nacc_c, nacc)
    # candidate scan sharded over the two ranks: one (value, index) pair all-reduced
    cand = torch.as_tensor(np.random.RandomState(3).uniform(-3.2, 3.2, (5001, 4)), device="cuda")
    full = utility_scan(gp, y, cand, bounds, "bape")
    v, gi, (b, e) = sharded_utility_scan(lambda b_, e_: utility_scan(gp, y, cand[b_:e_], bounds, "bape")[1:3], cand.shape[0])
    assert gi == full[2] and v == full[1], (rank, v, gi, full[1:])
    out[rank] = (chain.cpu().numpy(), nacc.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_process_hip_chain():
    import torch
    import torch.multiprocessing as mp
    from conftest import make_problem
    from alabi_amd import EnsembleSampler, HipGP
    assert torch.cuda.is_available()
    W, nsteps = 26, 60
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, W, nsteps, out), nprocs=2, join=True)
    X, y, h = make_problem(300, 4, 7)
    gp = HipGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(X)
    bounds = np.array([[-3.0, 3.0]] * 4)
    ref = EnsembleSampler(W, 4, gp, y, bounds, seed=4242)
    p0 = np.random.RandomState(9).uniform(-2, 2, (W, 4))
    ref.run_mcmc(p0, nsteps)
    chain = ref.get_chain()
    for r in range(2):
        c_r, n_r = out[r]
        assert np.array_equal(c_r, chain), f"rank {r}"
        assert np.array_equal(n_r, ref._naccept.cpu().numpy())


def test_sharded_run_single_rank_and_rccl_loads(monkeypatch):
    """nranks == 1 through the C loop equals the plain run, with the memcpy stand-in and with a REAL one-rank RCCL
    communicator (ncclCommInitRank + one ncclAllGather per half step on the stream); librccl.so resolves."""
    import ctypes as C
    import torch
    from conftest import make_problem
    from alabi_amd import EnsembleSampler, HipGP, _lib
    from alabi_amd.dist import ShardedRun
    X, y, h = make_problem(300, 4, 7)
    gp = HipGP(4, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(X)
    bounds = np.array([[-3.0, 3.0]] * 4)
    W, nsteps = 31, 75
    p0 = np.random.RandomState(9).uniform(-2, 2, (W, 4))
    ref = EnsembleSampler(W, 4, gp, y, bounds, seed=77); ref.run_mcmc(p0, nsteps, thin_by=3)
    s = EnsembleSampler(W, 4, gp, y, bounds, seed=77)
    chain, coords, logp, nacc = ShardedRun(s).run(torch.as_tensor(p0, device="cuda"), nsteps, thin_by=3)
    assert np.array_equal(chain.cpu().numpy(), ref.get_chain())
    assert np.array_equal(nacc.cpu().numpy(), ref._naccept.cpu().numpy())
    monkeypatch.setenv("ALABI_DIST_FORCE_RCCL", "1")
    s2 = EnsembleSampler(W, 4, gp, y, bounds, seed=77)
    run2 = ShardedRun(s2)
    chain2, _, _, nacc2 = run2.run(torch.as_tensor(p0, device="cuda"), nsteps, thin_by=3)
    torch.cuda.synchronize()
    assert np.array_equal(chain2.cpu().numpy(), ref.get_chain()) and np.array_equal(nacc2.cpu().numpy(), nacc.cpu().numpy())
    # two full chunks (1024 steps each) plus a remainder, thinned chain, RCCL communicator; with the graph opted in
    # (ALABI_ENS_SHARD_GRAPH=1: measured slower than eager launches, hence not the default) the chunks ARE replayed
    monkeypatch.setenv("ALABI_ENS_SHARD_GRAPH", "1")
    ref3 = EnsembleSampler(W, 4, gp, y, bounds, seed=5); ref3.run_mcmc(p0, 2100, thin_by=7)
    s3 = EnsembleSampler(W, 4, gp, y, bounds, seed=5)
    run3 = ShardedRun(s3)
    chain3, coords3, logp3, nacc3 = run3.run(torch.as_tensor(p0, device="cuda"), 2100, thin_by=7)
    assert np.array_equal(chain3.cpu().numpy(), ref3.get_chain())
    assert np.array_equal(nacc3.cpu().numpy(), ref3._naccept.cpu().numpy())
    assert np.array_equal(coords3.cpu().numpy(), ref3.get_last_sample().coords)
    st = run3.stats()
    assert st["graph_replays"] == 2 and st["eager_chunks"] == 1 and st["graph_captures"] == 1 and not st["failed"], st
    # the captured launches carry the handle's settings: a changed setting must re-capture, not replay stale arguments
    s4 = EnsembleSampler(W, 4, gp, y, bounds, seed=5, logp_affine=(2.0, -1.0))
    ref4 = EnsembleSampler(W, 4, gp, y, bounds, seed=5, logp_affine=(2.0, -1.0)); ref4.run_mcmc(p0, 1100)
    chain4, _, _, _ = ShardedRun(s4).run(torch.as_tensor(p0, device="cuda"), 1100)
    assert np.array_equal(chain4.cpu().numpy(), ref4.get_chain())
    monkeypatch.delenv("ALABI_ENS_SHARD_GRAPH")             # the default: the same chain, every chunk enqueued eagerly
    run5 = ShardedRun(EnsembleSampler(W, 4, gp, y, bounds, seed=5))
    chain5, _, _, _ = run5.run(torch.as_tensor(p0, device="cuda"), 2100, thin_by=7)
    st5 = run5.stats()
    assert np.array_equal(chain5.cpu().numpy(), ref3.get_chain()) and st5["graph_replays"] == 0 and st5["eager_chunks"] == 3
    del run2
    buf = C.create_string_buffer(128)
    _lib.check(_lib.lib().alabi_dist_unique_id(buf), "alabi_dist_unique_id")
    assert any(b != 0 for b in buf.raw)
