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
                 n_ensembles=1, logp_affine=(1.0, 0.0), normal_prior=None, **unused):
        """``n_ensembles`` > 1 runs that many INDEPENDENT ensembles of ``nwalkers`` walkers in the same kernel
        launches (rows [e*nwalkers, (e+1)*nwalkers) of every array belong to ensemble e).
        ``logp_affine=(scale, shift)``: log-probability = scale * GP mean + shift inside the box (an affine y scaler).
        ``normal_prior=(mean[d], std[d])``: independent normal priors on top of the box (NaN / non-positive std = none on
        that coordinate), the reference's ``lnprior_normal``."""
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
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).view(np.uint64)[0])
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.iteration = 0
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
        self.gp.predict_device(self._y, torch.zeros((1, self.ndim), dtype=torch.float64, device=_dev()))  # alpha ready
        h = self.gp.handle
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
        for k in ("_coords", "_logp", "_naccept"):
            st[k] = None if st[k] is None else st[k].cpu()
        st["_chains"] = [c.cpu() for c in self._chains]
        st["_chain_lps"] = [c.cpu() for c in self._chain_lps]
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
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
        _lib.check(_lib.lib().alabi_ens_lnprob(self._ens, _lib.ptr(c), _lib.ptr(lp), _lib.current_stream()), "alabi_ens_lnprob")
        return lp

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
        nstore = nsteps // thin_by if store else 0
        dev = self._coords.device
        chain = torch.empty((nstore, self.total_walkers, self.ndim), dtype=torch.float64, device=dev) if nstore else None
        chain_lp = torch.empty((nstore, self.total_walkers), dtype=torch.float64, device=dev) if nstore else None
        t0 = time.perf_counter()
        self._stream.wait_stream(torch.cuda.current_stream())
        backup = (self._coords.clone(), self._logp.clone(), self._naccept.clone())

        def _run():
            with torch.cuda.stream(self._stream):
                return _lib.lib().alabi_ens_run(self._ens, _lib.ptr(self._coords), _lib.ptr(self._logp), self.iteration,
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
        self.last_path = "stream" if path.value == 1 else "launch-per-half-step"
        self._stream.synchronize()
        torch.cuda.current_stream().wait_stream(self._stream)
        self.last_run_seconds = time.perf_counter() - t0
        self.iteration += nsteps
        if nstore:
            self._chains.append(chain)
            self._chain_lps.append(chain_lp)
            self._thins.append(thin_by)
        return State(self._coords.cpu().numpy(), self._logp.cpu().numpy())

    def reset(self):
        self._chains, self._chain_lps, self._thins = [], [], []
        self._naccept.zero_()
        self.iteration = 0

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
        x = self.get_chain_device(discard=discard, thin=thin)
        return thin * integrated_time(x, **kwargs)
