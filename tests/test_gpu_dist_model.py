"""The multi-GPU paths reached THROUGH SurrogateModel (reference: alabi/core.py:2300, :2322 run_emcee(multi_proc=True) hands a
pool to emcee; :349-369 pool factory; gp_utils.py:640-700 CV candidates over the pool), rehearsed with two ranks that share
cuda:0 under a gloo rendezvous (RCCL refuses two ranks on one device; the all-gather then goes through host memory):

* run_emcee(sampler_kwargs={"shard": True}): ONE ensemble over both ranks = the single-process chain, bit for bit;
* run_emcee() default: replicas -- rank 0's ensemble is the single-process run, rank 1's is another one, ``emcee_samples`` is the
  concatenation of both on both ranks;
* active_train: the candidate scan of find_next_point is sharded, both ranks append the point one rank alone picks;
* init_gp(hyperopt_method="cv"): the candidates are dealt over the ranks, the chosen hyper-parameters are those of one rank;
* a failing collective on one rank ends BOTH processes with a non-zero status instead of leaving the peer in the all-gather.
UNMEASURED on multi-GPU hardware (the driver's scaling run is the only multi-GPU execution)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(tmp, ntrain=120, seed=3):
    from alabi_amd import SurrogateModel
    from alabi_amd.benchmarks import gaussian_shells_nd
    g = gaussian_shells_nd(3)
    sm = SurrogateModel(lnlike_fn=g["fn"], bounds=g["bounds"], savedir=str(tmp), verbose=False, random_state=seed, cache=False)
    sm.init_samples(ntrain=ntrain)
    return sm


def _run_all(sm, tag, out, rank):
    """The same calls on every rank (and in the single-process reference)."""
    sm.init_gp(hyperopt_method="cv", cv_n_candidates=10, cv_stage2_candidates=6, cv_stage3_candidates=4)
    out[(tag, rank, "hyper")] = np.array(sm.gp.get_parameter_vector())
    sm.active_train(niter=3, gp_opt_freq=100, optimizer_kwargs={"ncand": 4096, "nrefine": 512, "polish": 10})
    out[(tag, rank, "theta")] = np.array(sm._theta)
    sm.run_emcee(nwalkers=20, nsteps=150, min_ess=0, sampler_kwargs={"shard": True}, burn=10, thin=1)
    out[(tag, rank, "shard_full")] = np.array(sm.emcee_samples_full)
    out[(tag, rank, "shard_mode")] = (sm.emcee_mode, sm.emcee_sampler.last_path)
    sm.run_emcee(nwalkers=20, nsteps=150, min_ess=0, burn=10, thin=1)
    out[(tag, rank, "rep_samples")] = np.array(sm.emcee_samples)
    out[(tag, rank, "rep_local")] = np.array(sm.emcee_sampler.get_chain(discard=10, thin=1, flat=True))
    out[(tag, rank, "rep_mode")] = sm.emcee_mode


def _worker(rank, world, port, tmp, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sm = _model(os.path.join(tmp, f"r{rank}"))
    _run_all(sm, "two", out, rank)
    dist.barrier()
    dist.destroy_process_group()


def test_surrogate_model_on_two_ranks_matches_one_rank(tmp_path):
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    for r in range(2):
        os.makedirs(tmp_path / f"r{r}", exist_ok=True)
    mp.spawn(_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    ref = {}
    os.makedirs(tmp_path / "one", exist_ok=True)
    _run_all(_model(tmp_path / "one"), "one", ref, 0)
    out = dict(out)
    # init_gp(cv): candidates dealt over the ranks, one MIN all-reduce of the score vector per stage
    for r in range(2):
        np.testing.assert_array_equal(out[("two", r, "hyper")], ref[("one", 0, "hyper")])
    # active_train: both ranks appended the same three points, the ones a single rank picks
    np.testing.assert_array_equal(out[("two", 0, "theta")], out[("two", 1, "theta")])
    assert out[("two", 0, "theta")].shape == ref[("one", 0, "theta")].shape
    np.testing.assert_allclose(out[("two", 0, "theta")], ref[("one", 0, "theta")], rtol=1e-9, atol=1e-11)
    # sharded ensemble: the single-process chain, bit for bit, on both ranks
    assert out[("two", 0, "shard_mode")] == ("sharded", "sharded") and ref[("one", 0, "shard_mode")][0] == "single"
    if np.array_equal(out[("two", 0, "theta")], ref[("one", 0, "theta")]):      # (same training set: same GP, so the chains must agree)
        for r in range(2):
            np.testing.assert_array_equal(out[("two", r, "shard_full")], ref[("one", 0, "shard_full")])
    np.testing.assert_array_equal(out[("two", 0, "shard_full")], out[("two", 1, "shard_full")])
    # replicas: different ensembles per rank, the gathered samples are their concatenation in rank order, on both ranks
    assert out[("two", 0, "rep_mode")] == "replicas" and ref[("one", 0, "rep_mode")] == "single"
    both = np.vstack([out[("two", 0, "rep_local")], out[("two", 1, "rep_local")]])
    for r in range(2):
        np.testing.assert_array_equal(out[("two", r, "rep_samples")], both)
    assert not np.array_equal(out[("two", 0, "rep_local")], out[("two", 1, "rep_local")])
    if np.array_equal(out[("two", 0, "theta")], ref[("one", 0, "theta")]):
        np.testing.assert_array_equal(out[("two", 0, "rep_local")], ref[("one", 0, "rep_local")])   # rank 0 = the single-process run


def _failing_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import make_problem
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.dist import ShardedRun
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, h = make_problem(200, 3, 7)
    gp = HipGP(3, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(X)
    s = EnsembleSampler(12, 3, gp, y, np.array([[-3.0, 3.0]] * 3), seed=1)
    calls = [0]

    def gather(block):                                       # the collective: fine for a while, then it fails on rank 1 only
        calls[0] += 1
        if rank == 1 and calls[0] == 7:
            raise RuntimeError("injected failure of the all-gather on rank 1")
        out = torch.empty(world * block.size, dtype=torch.float64)
        dist.all_gather_into_tensor(out, torch.from_numpy(block))
        return out.numpy()
    run = ShardedRun(s, allgather=gather)
    p0 = np.random.RandomState(2).uniform(-2, 2, (12, 3))
    run.run(torch.as_tensor(p0, device="cuda"), 50)
    os._exit(0)                                              # not reached on either rank


def test_failing_collective_ends_every_rank(tmp_path):
    """Rank 1's all-gather fails in the fourth step: rank 1 logs and exits with status 70; rank 0, inside the same all-gather,
    sees its peer go away (gloo raises), and exits the same way -- nobody is left waiting, the parent sees non-zero statuses."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    alive = [p.is_alive() for p in procs]
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert not any(alive), "a rank was left inside the collective"
    assert procs[1].exitcode == 70, procs[1].exitcode
    assert procs[0].exitcode not in (0, None), procs[0].exitcode
