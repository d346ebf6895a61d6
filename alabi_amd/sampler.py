"""GPU-resident ensemble sampler exposing the emcee.EnsembleSampler members alabi uses.

Reference seam: ``EnsembleSampler(nwalkers, ndim, log_prob_fn,
pool=)``, ``.run_mcmc(p0, nsteps, progress=True)``, ``.get_chain(discard, thin, flat)``,
``.get_last_sample().coords``, ``.acceptance_fraction``, ``.get_autocorr_time(tol=0)``
(alabi/core.py:2319-2387, alabi/mcmc_utils.py:45).

Instead of a Python ``log_prob_fn`` called once per walker, the log-probability
(surrogate GP mean + uniform box prior, alabi/core.py:2073-2100) is evaluated inside the
HIP half-step kernel; walker coordinates, log-probabilities, the random draws and the
chain never leave HBM during a run.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from . import _lib
from .gp import HipGP, _dev, _to_dev
from .mcmc_utils import integrated_time

__all__ = ["EnsembleSampler", "State"]


class State:
    def __init__(self, coords, log_prob=None):
        self.coords = coords
        self.log_prob = log_prob
        self.blobs = None
        self.random_state = None

    def __iter__(self):
        return iter((self.coords, self.log_prob, self.random_state))


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, gp, y, bounds, seed=None, a=2.0, pool=None, live_dangerously=False,
                 n_ensembles=1, logp_affine=(1.0, 0.0), normal_prior=None, logp_map=None, prior_fn=None, like_fn=None,
                 gate_box=True, shard=False, group=None, **unused):
        """``n_ensembles`` > 1 runs that many INDEPENDENT ensembles of ``nwalkers`` walkers in the same kernel
        launches (rows [e*nwalkers, (e+1)*nwalkers) of every array belong to ensemble e).
        ``logp_affine=(scale, shift)``: log-probability = scale * GP mean + shift inside the box (an affine y scaler).
        ``normal_prior=(mean[d], std[d])``: independent normal priors on top of the box (NaN / non-positive std = none on
        that coordinate), the reference's ``lnprior_normal``.
        ``logp_map``: None, "nlog" or "log" -- the inverse of the reference's non-affine y scalers (-10^x, 10^x,
        alabi/utility.py:62-71) applied to the (affinely mapped) GP mean inside the kernels.
        ``prior_fn`` / ``like_fn``: arbitrary HOST callables on a batch of points in the sampler's coordinates
        ([n,d] -> [n]); the reference's lnprob = like_fn + prior_fn with any Python callable (alabi/core.py:2073-2100).
        With either one set, every half step is split into a propose launch, the host call and an accept launch
        (alabi_ens_propose / alabi_ens_accept): the ensemble, the draws and the accept test stay on the device.  With
        ``like_fn=None`` the surrogate part is evaluated by the propose kernel; ``gate_box`` says whether that value is
        -inf outside ``bounds`` (True when the prior is the box itself, False when ``prior_fn`` is the whole prior).
        ``shard=True`` under an initialised ``torch.distributed`` group of more than one rank (``group``: default WORLD): ONE
        ensemble whose active half is partitioned over the ranks, an all-gather of the new rows per half step
        (alabi_amd.dist.ShardedRun -> alabi_ens_run_sharded); every rank must construct the sampler with the same arguments and
        make the same calls, and every rank ends with the same chain (counter-based draws).  The reference's analogue:
        ``EnsembleSampler(..., pool=pool)`` spreading one ensemble's lnprob calls over processes (alabi/core.py:2300, :2322)."""
        if not isinstance(gp, HipGP):
            raise TypeError("EnsembleSampler needs the HipGP surrogate (the log-probability is fused into the kernel)")
        self.nwalkers = int(nwalkers)
        self.n_ensembles = int(n_ensembles)
        self.total_walkers = self.nwalkers * self.n_ensembles
        self.ndim = int(ndim)
        if self.ndim != gp.ndim:
            raise ValueError("ndim does not match the GP")
        if self.nwalkers < 2 * self.ndim and not live_dangerously:
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than twice the "
                               "number of dimensions.")
        self.gp = gp
        self._y = y
        self.bounds = np.ascontiguousarray(np.asarray(bounds, dtype=np.float64).reshape(self.ndim, 2))
        self.a = float(a)
        self.logp_affine = (float(logp_affine[0]), float(logp_affine[1]))
        self.normal_prior = None
        if normal_prior is not None:
            m = np.ascontiguousarray(np.asarray(normal_prior[0], dtype=np.float64).reshape(self.ndim))
            sd = np.ascontiguousarray(np.asarray(normal_prior[1], dtype=np.float64).reshape(self.ndim))
            self.normal_prior = (m, sd)
        if logp_map not in (None, "nlog", "log"):
            raise ValueError("logp_map must be None, 'nlog' or 'log'")
        self.logp_map = logp_map
        self.prior_fn_host = prior_fn
        self.like_fn_host = like_fn
        self.generic = prior_fn is not None or like_fn is not None
        self.gate_box = bool(gate_box)
        if self.generic and self.n_ensembles != 1:
            raise ValueError("host callables need n_ensembles == 1")
        self.shard, self.group, self._sharded, self._sharded_ens = bool(shard), group, None, None
        if self.shard and (self.generic or self.n_ensembles != 1):
            raise ValueError("shard=True needs the fused log-probability (no host callables) and n_ensembles == 1")
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).view(np.uint64)[0])
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.iteration = 0
        self._rng_step = 0      # global step index of the counter-based draws: never rewound (reset() keeps it, as emcee's
                                # reset() keeps the RandomState), so a burn-in / reset / production run uses fresh draws
        self._coords = None
        self._logp = None
        self._chains = []
        self._chain_lps = []
        self._thins = []
        self._naccept = torch.zeros(self.total_walkers, dtype=torch.int64, device=_dev())
        self._stream = torch.cuda.Stream()
        self._ens = None
        self._ens_gp_handle = None
        self.last_run_seconds = 0.0

    # ------------------------------------------------------------------ handle lifetime
    def _ensure_ens(self):
        if self.like_fn_host is None:
            self.gp.predict_device(self._y, torch.zeros((1, self.ndim), dtype=torch.float64, device=_dev()))  # alpha ready
            h = self.gp.handle
        else:                    # the surrogate is not evaluated on the device: the GP only owns the ensemble handle
            if self.gp._handle is None:
                self.gp._ensure_handle(64)
            h = self.gp._handle
        if self._ens is not None and self._ens_gp_handle is not None and self._ens_gp_handle.value == h.value:
            return
        self._release()
        e = C.c_void_p()
        st = _lib.lib().alabi_ens_create(h, self.nwalkers, self.ndim, self.n_ensembles,
                                         _lib.host_doubles(self.bounds.ravel()), C.c_ulonglong(self.seed), C.byref(e))
        _lib.check(st, "alabi_ens_create")
        if self.logp_affine != (1.0, 0.0):
            _lib.check(_lib.lib().alabi_ens_set_logp_affine(e, self.logp_affine[0], self.logp_affine[1]),
                       "alabi_ens_set_logp_affine")
        if self.normal_prior is not None:
            _lib.check(_lib.lib().alabi_ens_set_normal_prior(e, _lib.host_doubles(self.normal_prior[0]),
                                                             _lib.host_doubles(self.normal_prior[1])),
                       "alabi_ens_set_normal_prior")
        if self.logp_map is not None:
            _lib.check(_lib.lib().alabi_ens_set_logp_map(e, {"nlog": 1, "log": 2}[self.logp_map]), "alabi_ens_set_logp_map")
        self._ens = e
        self._ens_gp_handle = C.c_void_p(h.value)

    def _release(self):
        if getattr(self, "_ens", None) is not None:
            try:
                torch.cuda.synchronize()
                _lib.lib().alabi_ens_destroy(self._ens)
            except Exception:
                pass
        self._ens = None
        self._ens_gp_handle = None

    def __del__(self):
        self._release()

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_ens"] = None
        st["_ens_gp_handle"] = None
        st["_stream"] = None
        st["_sharded"] = None
        st["_sharded_ens"] = None
        st["group"] = None
        st["prior_fn_host"] = None      # host callables (often closures) are not part of the saved state
        st["like_fn_host"] = None
        for k in ("_coords", "_logp", "_naccept"):
            st[k] = None if st[k] is None else st[k].cpu()
        st["_chains"] = [c.cpu() for c in self._chains]
        st["_chain_lps"] = [c.cpu() for c in self._chain_lps]
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if "_rng_step" not in st:                       # saved before the draw counter was separated from `iteration`
            self._rng_step = st.get("iteration", 0)
        if torch.cuda.is_available():
            self._stream = torch.cuda.Stream()
            for k in ("_coords", "_logp", "_naccept"):
                if getattr(self, k) is not None:
                    setattr(self, k, getattr(self, k).to(_dev()))

    # ------------------------------------------------------------------------- sampling
    def compute_log_prob(self, coords):
        """Surrogate mean + box prior for an ensemble of points (device in, device out)."""
        self._ensure_ens()
        c = _to_dev(coords, 2)
        if c.shape != (self.total_walkers, self.ndim):
            raise ValueError("coords must have shape (nwalkers * n_ensembles, ndim)")
        lp = torch.empty(self.total_walkers, dtype=torch.float64, device=c.device)
        if self.generic:
            return self._host_log_prob(c, self.surrogate(c) if self.like_fn_host is None else None)
        _lib.check(_lib.lib().alabi_ens_lnprob(self._ens, _lib.ptr(c), _lib.ptr(lp), _lib.current_stream()), "alabi_ens_lnprob")
        return lp

    def surrogate(self, points):
        """y_scaler^-1(GP mean) at arbitrary points [M,d] in the sampler's coordinates: no box gate, no prior (device tensor)."""
        self._ensure_ens()
        c = _to_dev(points, 2)
        out = torch.empty(c.shape[0], dtype=torch.float64, device=c.device)
        _lib.check(_lib.lib().alabi_ens_surrogate(self._ens, _lib.ptr(c), int(c.shape[0]), _lib.ptr(out), _lib.current_stream()),
                   "alabi_ens_surrogate")
        return out

    def _host_log_prob(self, q_dev, like_dev):
        """like (device values, or like_fn on the host) + prior_fn on the host for a batch of points; device tensor out.
        NaN raises as emcee does ("Probability function returned NaN")."""
        qh = q_dev.cpu().numpy()
        if like_dev is not None:
            lp = like_dev.cpu().numpy().copy()
        elif self.gate_box:                          # a host likelihood under the box prior: only evaluated inside the box
            inside = np.all((qh > self.bounds[:, 0]) & (qh < self.bounds[:, 1]), axis=1)
            lp = np.full(qh.shape[0], -np.inf)
            if inside.any():
                lp[inside] = np.asarray(self.like_fn_host(qh[inside]), dtype=np.float64).reshape(-1)
        else:
            lp = np.asarray(self.like_fn_host(qh), dtype=np.float64).reshape(-1).copy()
        if self.prior_fn_host is not None:
            lp = lp + np.asarray(self.prior_fn_host(qh), dtype=np.float64).reshape(-1)
        if lp.shape != (qh.shape[0],):
            raise ValueError("like_fn / prior_fn must return one value per point")
        if np.any(np.isnan(lp)):
            raise ValueError("Probability function returned NaN")
        return torch.as_tensor(lp, device=q_dev.device)

    def _run_generic(self, nsteps, thin_by, chain, chain_lp):
        """Half steps split around the host callables (alabi_ens_propose -> host -> alabi_ens_accept)."""
        lib, W, d = _lib.lib(), self.nwalkers, self.ndim
        n0 = (W + 1) // 2
        dev = self._coords.device
        stream = _lib.current_stream()
        want_like = self.like_fn_host is None
        if self.prior_fn_host is None and self.like_fn_host is None:
            raise RuntimeError("this sampler was set up with host callables (prior_fn / like_fn) that are not part of its "
                               "saved state: create a new sampler with them")
        # steps per draw: at most the library's draw-buffer chunk (4M / walkers, between 16 and 1024 steps)
        cap = max(16, min(1024, (4 << 20) // max(self.total_walkers, 1)))
        done = 0
        while done < nsteps:
            n = min(256, cap, nsteps - done)
            _lib.check(lib.alabi_ens_draw(self._ens, self._rng_step + done, n, self.a, stream), "alabi_ens_draw")
            for t in range(n):
                for split in (0, 1):
                    nS = n0 if split == 0 else W - n0
                    if nS == 0:
                        continue
                    q = torch.empty((nS, d), dtype=torch.float64, device=dev)
                    like = torch.empty(nS, dtype=torch.float64, device=dev) if want_like else None
                    _lib.check(lib.alabi_ens_propose(self._ens, _lib.ptr(self._coords), t, split, int(self.gate_box),
                                                     _lib.ptr(q), _lib.ptr(like), stream), "alabi_ens_propose")
                    lp_new = self._host_log_prob(q, like)
                    _lib.check(lib.alabi_ens_accept(self._ens, _lib.ptr(self._coords), _lib.ptr(self._logp), t, split,
                                                    _lib.ptr(q), _lib.ptr(lp_new), _lib.ptr(self._naccept), stream),
                               "alabi_ens_accept")
                k = done + t + 1
                if chain is not None and k % thin_by == 0:
                    chain[k // thin_by - 1].copy_(self._coords)
                    chain_lp[k // thin_by - 1].copy_(self._logp)
            done += n
        torch.cuda.current_stream().synchronize()

    def run_mcmc(self, initial_state, nsteps, thin_by=1, progress=False, store=True, skip_initial_state_check=False, **kw):
        nsteps = int(nsteps)
        thin_by = int(thin_by)
        if thin_by < 1:
            raise ValueError("thin_by must be a positive integer")
        if initial_state is None:
            if self._coords is None:
                raise ValueError("Cannot have `initial_state=None` if run_mcmc has never been called.")
        else:
            coords = initial_state.coords if isinstance(initial_state, State) else initial_state
            coords = _to_dev(coords, 2).clone()
            if coords.shape != (self.total_walkers, self.ndim):
                raise ValueError("incompatible input dimensions: initial state must be (nwalkers * n_ensembles, ndim)")
            if not skip_initial_state_check and self.nwalkers > 1:
                c = coords.cpu().numpy()
                c = c - c.mean(axis=0)
                smax = np.abs(c).max(axis=0)
                if np.any(smax == 0):
                    raise ValueError("Initial state has a large condition number. Make sure that your walkers are "
                                     "linearly independent for the best performance")
                if np.linalg.cond((c / smax).astype(float)) > 1e8:
                    raise ValueError("Initial state has a large condition number. Make sure that your walkers are "
                                     "linearly independent for the best performance")
            self._coords = coords
            self._logp = self.compute_log_prob(coords)
            if torch.isnan(self._logp).any():
                raise ValueError("The initial log_prob was NaN")
        self._ensure_ens()
        if self.shard:
            from .dist import ShardedRun, world_info
            if world_info(self.group)[1] > 1:
                t0s = time.perf_counter()
                nstore = nsteps // thin_by if store else 0
                if self._sharded is None or self._sharded.s is not self or self._sharded_ens is not self._ens:
                    self._sharded, self._sharded_ens = ShardedRun(self, self.group), self._ens
                ch, co, lp, nacc = self._sharded.run(self._coords, nsteps, step0=self._rng_step, a=self.a, thin_by=thin_by,
                                                     store=bool(nstore), logp0=self._logp)
                torch.cuda.current_stream().synchronize()
                self._coords, self._logp = co, lp
                self._naccept += nacc
                self.last_path = "sharded"
                self.last_run_seconds = time.perf_counter() - t0s
                self.iteration += nsteps
                self._rng_step += nsteps
                if nstore:
                    self._chains.append(ch); self._chain_lps.append(self._sharded.last_chain_logp); self._thins.append(thin_by)
                return State(self._coords.cpu().numpy(), self._logp.cpu().numpy())
        nstore = nsteps // thin_by if store else 0
        dev = self._coords.device
        chain = torch.empty((nstore, self.total_walkers, self.ndim), dtype=torch.float64, device=dev) if nstore else None
        chain_lp = torch.empty((nstore, self.total_walkers), dtype=torch.float64, device=dev) if nstore else None
        t0 = time.perf_counter()
        if self.generic:
            self._run_generic(nsteps, thin_by, chain, chain_lp)
            self.last_path = "host-callback"
            self.last_run_seconds = time.perf_counter() - t0
            self.iteration += nsteps
            self._rng_step += nsteps
            if nstore:
                self._chains.append(chain); self._chain_lps.append(chain_lp); self._thins.append(thin_by)
            return State(self._coords.cpu().numpy(), self._logp.cpu().numpy())
        # the backup copies are taken on the current stream BEFORE the run stream is made to wait for it: the run's
        # kernels (which update coords / logp in place) are then ordered after them
        backup = (self._coords.clone(), self._logp.clone(), self._naccept.clone())
        self._stream.wait_stream(torch.cuda.current_stream())

        def _run():
            with torch.cuda.stream(self._stream):
                return _lib.lib().alabi_ens_run(self._ens, _lib.ptr(self._coords), _lib.ptr(self._logp), self._rng_step,
                                                nsteps, thin_by, self.a, _lib.ptr(chain), _lib.ptr(chain_lp),
                                                _lib.ptr(self._naccept), C.c_void_p(self._stream.cuda_stream))

        st = _run()
        if st == _lib.TIMEOUT:
            # the persistent kernel gave up on a hand-off (e.g. the GPU was shared and its workgroups were not all
            # resident): restore the state and repeat the run with one launch per half step
            self._stream.synchronize()
            self._coords.copy_(backup[0]); self._logp.copy_(backup[1]); self._naccept.copy_(backup[2])
            _lib.check(_lib.lib().alabi_ens_set_stream(self._ens, 0), "alabi_ens_set_stream")
            self.stream_fallbacks = getattr(self, "stream_fallbacks", 0) + 1
            st = _run()
        _lib.check(st, "alabi_ens_run")
        path = C.c_int(0)
        _lib.lib().alabi_ens_last_path(self._ens, C.byref(path))
        self.last_path = {1: "stream", 3: "group"}.get(path.value, "launch-per-half-step")
        self.last_stream_kernel = {1: "ens_stream_kernel", 3: "ens_group_kernel"}.get(path.value)
        self.group_plan = None
        if path.value == 3:                                   # which instantiation of ens_group_kernel ran (tests pin it)
            plan = (C.c_int * 8)()
            _lib.lib().alabi_ens_group_plan(self._ens, plan)
            self.group_plan = dict(zip(("Q", "G", "NG", "RT", "tpm", "ltw", "KS", "lds_bytes"), (int(v) for v in plan)))
        self._stream.synchronize()
        torch.cuda.current_stream().wait_stream(self._stream)
        self.last_run_seconds = time.perf_counter() - t0
        self.iteration += nsteps
        self._rng_step += nsteps
        if nstore:
            self._chains.append(chain)
            self._chain_lps.append(chain_lp)
            self._thins.append(thin_by)
        return State(self._coords.cpu().numpy(), self._logp.cpu().numpy())

    def reset(self):
        """emcee's reset(): forget the chain and the acceptance counts.  The draw counter is NOT rewound."""
        self._chains, self._chain_lps, self._thins = [], [], []
        self._naccept.zero_()
        self.iteration = 0
        self._tau_memo = None

    # -------------------------------------------------------------------------- results
    def get_chain_device(self, discard=0, thin=1, flat=False, log_prob=False):
        src = self._chain_lps if log_prob else self._chains
        if not src:
            raise AttributeError("you must run the sampler with 'store == True' before accessing the results")
        v = src[0] if len(src) == 1 else torch.cat(src, dim=0)
        v = v[discard + thin - 1::thin]
        if flat:
            v = v.reshape((-1,) + tuple(v.shape[2:]))
        return v

    def get_chain(self, discard=0, thin=1, flat=False):
        return self.get_chain_device(discard, thin, flat).cpu().numpy()

    def get_log_prob(self, discard=0, thin=1, flat=False):
        return self.get_chain_device(discard, thin, flat, log_prob=True).cpu().numpy()

    def get_last_sample(self):
        if self._coords is None:
            raise AttributeError("you must run the sampler before accessing the results")
        return State(self._coords.cpu().numpy(), self._logp.cpu().numpy())

    @property
    def acceptance_fraction(self):
        return self._naccept.cpu().numpy() / float(max(self.iteration, 1))

    def get_autocorr_time(self, discard=0, thin=1, **kwargs):
        # run_emcee asks twice for the same numbers (estimate_burnin, then the summary: mcmc_utils.py:45, core.py:2387): remembered
        # until the chain grows or is reset
        key = (self.iteration, len(self._chains), int(discard), int(thin), tuple(sorted(kwargs.items())))
        memo = getattr(self, "_tau_memo", None)
        if memo is not None and memo[0] == key:
            return memo[1].copy()
        x = self.get_chain_device(discard=discard, thin=thin)
        tau = thin * integrated_time(x, **kwargs)
        self._tau_memo = (key, np.array(tau, copy=True))
        return tau
