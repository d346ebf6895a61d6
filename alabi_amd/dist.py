"""Multi-GPU ensemble sampling: one process per GPU, ``torch.distributed`` (backend "nccl" is RCCL
over xGMI on ROCm; "gloo" for the CPU tests).

Two ways the path shards:

* ``replicas``  -- independent ensembles / chains, one per rank, different seeds, NO data-path
  collective; samples are concatenated at the end.  This is how ``bench.py --gpus N`` scales the
  headline configuration (256 walkers per GPU).
* ``ShardedEnsemble`` -- ONE ensemble of W walkers whose active half is partitioned across ranks.
  Within a half step the proposals are independent given the frozen complementary half, so each
  rank runs the HIP half-step kernel on its slice [begin, end) of the active list and the updated
  (coords, logp) rows are exchanged with ONE all-gather per half step.  The random draws are
  counter-based (seed, step, walker), so every rank generates identical lists and the chain is
  bit-identical for any number of ranks.

The compute backend is injected: ``HipBackend`` (product) wraps the C ABI; the gloo tests pass an
oracle-backed stand-in with the same four methods to check the partition / gather logic on CPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

__all__ = ["HipBackend", "ShardedEnsemble", "ShardedRun", "sharded_utility_scan", "reduce_min_index", "slice_bounds",
           "gather_replicas", "world_info", "allreduce_array", "allgather_rows"]


def world_info(group=None):
    """(rank, world) of the initialised process group, (0, 1) without one -- what SurrogateModel consults to decide whether
    its multi-GPU paths are on (one process per GPU, launched by torch.distributed.run)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _comm_device(group=None):
    """Tensors handed to a collective live on the GPU under "nccl" (= RCCL) and on the host under "gloo" (CPU tests, and the
    test rig with several ranks on one GPU)."""
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def allreduce_array(values, op="min", group=None):
    """Element-wise all-reduce (min / max / sum) of a small float64 host array; returns a NumPy array, identical on every rank."""
    v = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
    if world_info(group)[1] == 1:
        return v.copy()
    t = torch.as_tensor(v, device=_comm_device(group)).clone()
    dist.all_reduce(t, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op], group=group)
    return t.cpu().numpy()


def allgather_rows(rows, group=None):
    """All-gather of equally shaped float64 blocks [m, c] (one per rank) -> [world * m, c] on every rank, rank order."""
    r = np.ascontiguousarray(np.asarray(rows, dtype=np.float64))
    world = world_info(group)[1]
    if world == 1:
        return r.copy()
    dev = _comm_device(group)
    send = torch.as_tensor(r, device=dev).reshape(-1).clone()
    recv = torch.empty(world * send.numel(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv.cpu().numpy().reshape((world * r.shape[0],) + r.shape[1:])


def slice_bounds(n, world, rank):
    """[begin, end) of rank's contiguous share of n items (shares differ by at most one)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class HipBackend:
    """The four operations ShardedEnsemble needs, on the HIP library."""

    def __init__(self, sampler):
        self.s = sampler
        sampler._ensure_ens()
        self.W, self.d = sampler.nwalkers, sampler.ndim
        self.device = sampler._naccept.device
        self.chunk = 256

    def lnprob(self, coords):
        return self.s.compute_log_prob(coords)

    def draw(self, step0, n, a):
        _lib.check(_lib.lib().alabi_ens_draw(self.s._ens, int(step0), int(n), float(a), _lib.current_stream()), "alabi_ens_draw")

    def order(self, t):
        out = torch.empty(self.W, dtype=torch.int32, device=self.device)
        n0 = C.c_int(0)
        _lib.check(_lib.lib().alabi_ens_step_lists(self.s._ens, int(t), _lib.ptr(out), C.byref(n0), _lib.current_stream()),
                   "alabi_ens_step_lists")
        return out.long(), int(n0.value)

    def half_step(self, coords, logp, t, split, begin, end, a, n_accept):
        st = _lib.lib().alabi_ens_half_step(self.s._ens, _lib.ptr(coords), _lib.ptr(logp), int(t), int(split), int(begin),
                                            int(end), _lib.ptr(n_accept), _lib.current_stream())
        _lib.check(st, "alabi_ens_half_step")


class ShardedEnsemble:
    def __init__(self, backend, group=None):
        self.b = backend
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.W, self.d = backend.W, backend.d

    def _exchange(self, coords, logp, S):
        """All-gather the rows of the active list S (each rank owns slice_bounds(len(S)))."""
        if self.world == 1:
            return
        nS = S.numel()
        per = -(-nS // self.world)                       # equal-size slots (all_gather_into_tensor needs it)
        b, e = slice_bounds(nS, self.world, self.rank)
        send = torch.zeros((per, self.d + 1), dtype=torch.float64, device=coords.device)
        mine = S[b:e]
        send[: e - b, : self.d] = coords.index_select(0, mine)
        send[: e - b, self.d] = logp.index_select(0, mine)
        recv = torch.empty((self.world * per, self.d + 1), dtype=torch.float64, device=coords.device)
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            # test rig only (several ranks sharing ONE GPU cannot use RCCL): stage the same all-gather through host memory
            recv_h = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(recv_h, send.cpu(), group=self.group)
            recv.copy_(recv_h)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        for r in range(self.world):
            rb, re = slice_bounds(nS, self.world, r)
            if r == self.rank or re == rb:
                continue
            rows = recv[r * per: r * per + (re - rb)]
            coords.index_copy_(0, S[rb:re], rows[:, : self.d])
            logp.index_copy_(0, S[rb:re], rows[:, self.d])

    def run(self, coords, nsteps, step0=0, a=2.0, thin_by=1, store=True):
        """Returns (chain[nsteps//thin_by, W, d] or None, coords, logp, n_accept) -- identical on every rank."""
        dev = coords.device
        coords = coords.clone()
        logp = self.b.lnprob(coords).clone()
        n_accept = torch.zeros(self.W, dtype=torch.int64, device=dev)
        nstore = nsteps // thin_by if store else 0
        chain = torch.empty((nstore, self.W, self.d), dtype=torch.float64, device=dev) if nstore else None
        done = 0
        while done < nsteps:
            n = min(self.b.chunk, nsteps - done)
            self.b.draw(step0 + done, n, a)
            for t in range(n):
                order, n0 = self.b.order(t)
                for split in (0, 1):
                    S = order[:n0] if split == 0 else order[n0:]
                    b, e = slice_bounds(S.numel(), self.world, self.rank)
                    self.b.half_step(coords, logp, t, split, b, e, a, n_accept)
                    self._exchange(coords, logp, S)
                k = done + t + 1
                if nstore and k % thin_by == 0:
                    chain[k // thin_by - 1] = coords
            done += n
        if self.world > 1:   # each walker was counted by exactly one rank
            if n_accept.is_cuda and dist.get_backend(self.group) == "gloo":
                h = n_accept.cpu()
                dist.all_reduce(h, group=self.group)
                n_accept.copy_(h)
            else:
                dist.all_reduce(n_accept, group=self.group)
        return chain, coords, logp, n_accept


class ShardedRun:
    """The same sharded ensemble with the WHOLE step loop inside the library (alabi_ens_run_sharded): per half step the
    half-step kernel on this rank's slice writes its new rows straight into the rank's segment of a history block and ONE
    in-place all-gather completes the block -- two enqueues per half step, no pack / unpack kernels, no host read-back; a
    full chunk of steps is one hipGraph replay (RCCL or single rank).  The all-gather is RCCL's ncclAllGather on a communicator the library creates
    itself (rank 0 draws the unique id, torch.distributed only broadcasts its 128 bytes); under a "gloo" process group
    (test rig: several ranks on ONE GPU, which RCCL refuses) a host callback stands in for it."""

    def __init__(self, sampler, group=None, allgather=None):
        """``allgather``: optional host function ``f(block: float64[count]) -> float64[world * count]`` (rank order) that
        replaces the collective -- the library then calls it once per half step through its callback communicator
        (alabi_dist_comm_create_callback).  Default: RCCL under "nccl", ``dist.all_gather_into_tensor`` on host copies under
        "gloo"."""
        self.s = sampler
        sampler._ensure_ens()
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.W, self.d = sampler.nwalkers, sampler.ndim
        lib = _lib.lib()
        comm = C.c_void_p()
        self._cb = None
        self.last_callback_error = None
        self.last_chain_logp = None
        if self.world > 1 and (allgather is not None or dist.get_backend(group) != "nccl"):
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            hip.hipStreamSynchronize.argtypes = [C.c_void_p]
            world = self.world

            def _allgather(send, recv, count, user, stream):          # device pointers; staged through host memory
                try:
                    hip.hipStreamSynchronize(C.c_void_p(stream))
                    h = np.empty(count, dtype=np.float64)
                    if hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), C.c_void_p(send), count * 8, 2) != 0:
                        return 1
                    if allgather is not None:
                        o = np.ascontiguousarray(allgather(h), dtype=np.float64)
                        if o.shape != (world * count,):
                            raise ValueError("allgather must return world * count values")
                    else:
                        out = torch.empty(world * count, dtype=torch.float64)
                        dist.all_gather_into_tensor(out, torch.from_numpy(h), group=group)
                        o = out.numpy()
                    return 0 if hip.hipMemcpy(C.c_void_p(recv), o.ctypes.data_as(C.c_void_p), world * count * 8, 1) == 0 else 1
                except Exception as ex:  # noqa: BLE001  (no exception may cross the C boundary: keep its text for the caller)
                    self.last_callback_error = repr(ex)
                    return 1
            self._cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p)(_allgather)
            _lib.check(lib.alabi_dist_comm_create_callback(C.cast(self._cb, C.c_void_p), None, self.rank, self.world,
                                                           C.byref(comm)), "alabi_dist_comm_create_callback")
        else:
            uid = None
            if self.world == 1 and os.environ.get("ALABI_DIST_FORCE_RCCL") == "1":   # one-rank RCCL communicator (rehearsal)
                buf = C.create_string_buffer(128)
                _lib.check(lib.alabi_dist_unique_id(buf), "alabi_dist_unique_id")
                uid = buf
            if self.world > 1:
                box = [None]
                if self.rank == 0:
                    buf = C.create_string_buffer(128)
                    _lib.check(lib.alabi_dist_unique_id(buf), "alabi_dist_unique_id")
                    box[0] = buf.raw
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                uid = C.create_string_buffer(box[0], 128)
            _lib.check(lib.alabi_dist_comm_create(uid, self.rank, self.world, C.byref(comm)), "alabi_dist_comm_create")
        self._comm = comm

    def __del__(self):
        try:
            if getattr(self, "_comm", None) is not None:
                torch.cuda.synchronize()
                _lib.lib().alabi_dist_comm_destroy(self._comm)
        except Exception:  # noqa: BLE001
            pass
        self._comm = None

    def stats(self):
        """Counters of the C loop on this communicator: full chunks replayed from the captured hipGraph, chunks enqueued launch
        by launch, graph captures, and whether the communicator is dead after a failed run."""
        out = (C.c_longlong * 4)()
        _lib.check(_lib.lib().alabi_dist_comm_stats(self._comm, out), "alabi_dist_comm_stats")
        return dict(graph_replays=int(out[0]), eager_chunks=int(out[1]), graph_captures=int(out[2]), failed=bool(out[3]))

    def run(self, coords, nsteps, step0=0, a=2.0, thin_by=1, store=True, logp0=None):
        """Returns (chain[nsteps//thin_by, W, d] or None, coords, logp, n_accept) -- identical on every rank; the stored
        log-probabilities are left in ``self.last_chain_logp``.

        A failure on ONE rank (a HIP error, a collective that returns an error) would leave its peers inside an all-gather that
        never completes, and returning to the caller would only move the hang to the next collective: with more than one rank
        the error is logged and THIS PROCESS ENDS with exit status 70 (``os._exit``: no atexit handlers, no further
        collectives) -- the launcher (torch.distributed.run) then takes the other ranks down."""
        dev = coords.device
        coords = coords.clone()
        logp = (self.s.compute_log_prob(coords) if logp0 is None else logp0).clone()
        n_accept = torch.zeros(self.W, dtype=torch.int64, device=dev)
        nstore = nsteps // thin_by if store else 0
        chain = torch.empty((nstore, self.W, self.d), dtype=torch.float64, device=dev) if nstore else None
        chain_lp = torch.empty((nstore, self.W), dtype=torch.float64, device=dev) if nstore else None
        # on the sampler's own stream: a chunk is captured into a hipGraph, which the null stream (torch's default) cannot do
        run_stream = self.s._stream
        run_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(run_stream):
            st = _lib.lib().alabi_ens_run_sharded(self.s._ens, self._comm, _lib.ptr(coords), _lib.ptr(logp), int(step0), int(nsteps),
                                                  int(thin_by), float(a), _lib.ptr(chain), _lib.ptr(chain_lp), _lib.ptr(n_accept),
                                                  C.c_void_p(run_stream.cuda_stream))
        if st == 0:
            torch.cuda.current_stream().wait_stream(run_stream)
        if st != 0:
            detail = self.last_callback_error or _lib.lib().alabi_last_error().decode()
            msg = f"alabi_ens_run_sharded failed on rank {self.rank} of {self.world} (status {st}): {detail}"
            if self.world > 1:
                import sys
                print(msg + " -- peers may be inside the all-gather: ending this process (exit status 70)", file=sys.stderr, flush=True)
                os._exit(70)
            raise RuntimeError(msg)
        self.last_chain_logp = chain_lp
        return chain, coords, logp, n_accept


def reduce_min_index(val, idx, group=None):
    """All-reduce of ONE (value, global index) pair: the smallest value wins, ties go to the smallest index; a rank
    without a finite candidate contributes (+inf, -1).  Returns (value, index) on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(val), int(idx)
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    pair = torch.tensor([float(val), float(idx)], dtype=torch.float64, device=dev)   # indices < 2^53 are exact in fp64
    allp = torch.empty(2 * world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(allp, pair, group=group)
    allp = allp.cpu().numpy().reshape(world, 2)
    best_v, best_i = np.inf, -1
    for v, i in allp:
        if i >= 0 and (v < best_v or (v == best_v and i < best_i)):
            best_v, best_i = float(v), int(i)
    return best_v, best_i


def sharded_utility_scan(scan_fn, M, group=None):
    """Candidate scan sharded over the ranks (BASELINE.json config C5: BAPE over 10^6
    candidates on 8 GPUs).  The M candidates are partitioned into contiguous slices; every rank holds the (replicated,
    redundantly factorised) GP and scores ITS slice with ``scan_fn(begin, end) -> (best_value, best_local_index)``
    (product: alabi_amd.utility.utility_scan on candidates[begin:end]); ONE (value, global index) pair is all-reduced.
    Returns (value, global index, (begin, end)) -- identical value / index on every rank; index -1 when no candidate
    is finite.  The reference has no equivalent (its parallel axis is a process pool over scipy restarts,
    alabi/utility.py:1030-1163)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    b, e = slice_bounds(int(M), world, rank)
    val, li = (np.inf, -1) if e == b else scan_fn(b, e)
    gi = b + int(li) if (li is not None and int(li) >= 0 and np.isfinite(val)) else -1
    v, i = reduce_min_index(val if gi >= 0 else np.inf, gi, group)
    return v, i, (b, e)


def gather_replicas(samples, group=None):
    """Concatenate per-rank flat sample arrays [n_r, d] on every rank (replica mode post-processing)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return samples
    objs = [None] * dist.get_world_size(group)
    dist.all_gather_object(objs, np.asarray(samples), group=group)
    return np.concatenate(objs, axis=0)
