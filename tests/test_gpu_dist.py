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
