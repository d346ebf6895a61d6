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

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

__all__ = ["HipBackend", "ShardedEnsemble", "slice_bounds", "gather_replicas"]


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


def gather_replicas(samples, group=None):
    """Concatenate per-rank flat sample arrays [n_r, d] on every rank (replica mode post-processing)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return samples
    objs = [None] * dist.get_world_size(group)
    dist.all_gather_object(objs, np.asarray(samples), group=group)
    return np.concatenate(objs, axis=0)
