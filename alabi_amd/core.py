"""SurrogateModel: drop-in for alabi's orchestration class on the GP-surrogate + MCMC hot path.

Keeps the public surface of ``alabi.core.SurrogateModel`` that BASELINE.json's north_star names
(``init_samples`` / ``init_gp`` / ``active_train`` / ``run_emcee`` a.k.a. ``run_mcmc``,
``surrogate_log_likelihood``, ``create_cached_surrogate_likelihood``, ``lnprob``, ``theta()``,
``y()``, ``training_results`` keys, result attributes) -- reference: alabi/core.py:248-251
(ctor), :542 (init_samples), :736-764 (init_gp), :1097 (_fit_gp), :1163 (_opt_gp), :1446
(surrogate_log_likelihood), :1535, :1587 (find_next_point), :1670 (active_train), :2073
(lnprob), :2108 (run_emcee) -- and swaps the two third-party engines behind it:
george.GP -> ``HipGP`` and emcee.EnsembleSampler -> ``alabi_amd.sampler.EnsembleSampler``.

Out of scope here: nested samplers, plotting, MPI / process pools, the
parallel-chain trainer.  Deliberate differences are listed in DESIGN.md ("Differences").
"""
from __future__ import annotations

import copy
import os
import pickle
import time
import warnings
from functools import partial

import numpy as np
import scipy.optimize as op
import torch

from . import gp_utils, mcmc_utils
from . import utility as ut
from .gp import HipGP, _dev
from .sampler import EnsembleSampler

__all__ = ["SurrogateModel", "CachedSurrogateLikelihood"]

_KERNELS = ("ExpSquaredKernel", "RationalQuadraticKernel", "Matern32Kernel", "Matern52Kernel")


def _is_identity(scaler, probe):
    try:
        return bool(np.array_equal(np.asarray(scaler.transform(probe)), np.asarray(probe)))
    except Exception:  # noqa: BLE001
        return False


def _affine_map(fn, box):
    """(mult, add) with fn(x) == mult * x + add per column on `box` ([d, 2] lower / upper), or None if fn is not an
    increasing-or-decreasing affine map per dimension there (checked at three interior points)."""
    try:
        box = np.asarray(box, dtype=np.float64).reshape(-1, 2)
        lo, hi = box[:, 0], box[:, 1]
        f_lo = np.asarray(fn(lo.reshape(1, -1)), dtype=np.float64).reshape(-1)
        f_hi = np.asarray(fn(hi.reshape(1, -1)), dtype=np.float64).reshape(-1)
        if f_lo.shape != lo.shape or np.any(hi == lo):
            return None
        mult = (f_hi - f_lo) / (hi - lo)
        add = f_lo - mult * lo
        if not (np.all(np.isfinite(mult)) and np.all(np.isfinite(add)) and np.all(mult != 0)):
            return None
        for frac in (0.25, 0.5, 0.8):
            x = lo + frac * (hi - lo)
            fx = np.asarray(fn(x.reshape(1, -1)), dtype=np.float64).reshape(-1)
            if not np.allclose(fx, mult * x + add, rtol=1e-11, atol=1e-11 * (np.abs(f_hi) + np.abs(f_lo) + 1e-300)):
                return None
        return mult, add
    except Exception:  # noqa: BLE001
        return None


class CachedSurrogateLikelihood:
    """Picklable callable: GP factorised once, then mean(-and-variance) predictions per call
    (alabi/core.py:28-122)."""

    def __init__(self, gp_iter, _y_cond, theta_scaler, y_scaler, ndim, return_var=False):
        self.gp_iter = gp_iter
        self._y_cond = _y_cond
        self.theta_scaler = theta_scaler
        self.y_scaler = y_scaler
        self.ndim = ndim
        self.return_var = return_var

    def __call__(self, theta_xs):
        theta_xs = np.asarray(theta_xs)
        one = theta_xs.ndim == 1
        if one:
            theta_xs = theta_xs.reshape(1, -1)
        elif theta_xs.ndim != 2:
            raise ValueError(f"theta_xs must be 1D or 2D array, got {theta_xs.ndim}D")
        _t = np.atleast_2d(self.theta_scaler.transform(theta_xs))
        if _t.shape[0] == 1 and _t.shape[1] != self.ndim and _t.size == self.ndim:
            _t = _t.reshape(1, -1)
        if not self.return_var:
            _yp = self.gp_iter.predict(self._y_cond, _t, return_var=False, return_cov=False)
            yp = self.y_scaler.inverse_transform(_yp.reshape(-1, 1)).flatten()
            return yp[0] if one else yp
        _yp, _vp = self.gp_iter.predict(self._y_cond, _t, return_var=True, return_cov=False)
        yp = self.y_scaler.inverse_transform(_yp.reshape(-1, 1)).flatten()
        if getattr(self.y_scaler, "scale_", None) is not None:
            vp = _vp * self.y_scaler.scale_[0] ** 2
        else:
            eps = 1e-6
            tr = self.y_scaler.inverse_transform(np.array([[0.0], [eps]]))
            vp = _vp * ((tr[1] - tr[0]) / eps) ** 2
        return (yp[0], vp[0]) if one else (yp, vp)


class SurrogateModel(object):
    def __init__(self, lnlike_fn=None, bounds=None, param_names=None, cache=True, savedir="results/",
                 model_name="surrogate_model", verbose=True, ncore=1, pool_method="forkserver",
                 ignore_warnings=True, random_state=None):
        if lnlike_fn is None:
            raise ValueError("Must supply lnlike_fn to train GP surrogate model.")
        if bounds is None:
            raise ValueError("Must supply prior bounds.")
        if random_state is None:
            random_state = int(time.time() * 1000000) % (2 ** 32)
        self.random_state = random_state
        self._rng = np.random.RandomState(random_state)
        self.lnlike_fn = lnlike_fn
        self.true_log_likelihood = lnlike_fn
        self.bounds = np.array(bounds)
        self.ndim = len(self.bounds)
        self.prior_sampler = partial(ut.prior_sampler, bounds=self.bounds, sampler="uniform", random_state=None)
        if param_names is not None:
            if len(param_names) != len(bounds):
                raise ValueError("Length of param_names must match length of bounds.")
            self.param_names = param_names
            self.labels = param_names
        else:
            self.param_names = [r"$\theta_%s$" % (i) for i in range(self.ndim)]
            self.labels = [f"theta_{i}" for i in range(self.ndim)]
        self.cache = cache
        self.savedir = savedir
        if not os.path.exists(self.savedir):
            os.makedirs(self.savedir)
        self.model_name = model_name
        self.verbose = verbose
        if ignore_warnings:
            warnings.filterwarnings("ignore", category=UserWarning)
            warnings.filterwarnings("ignore", category=FutureWarning)
        self.pool_method = pool_method
        self.ncore = max(int(ncore), 1)   # process pools are not used: the parallel axis is the GPU
        self.mpi_is_active = False
        self.emcee_run = False
        self.dynesty_run = False
        self.ultranest_run = False

    # ------------------------------------------------------------------------- persistence
    def __getstate__(self):
        state = self.__dict__.copy()
        state["pool"] = None
        if state.get("_emcee_full_src") is not None:          # the saved model carries the array itself, as the reference's does
            state["_emcee_full"] = self.emcee_samples_full
            state["_emcee_full_src"] = None
        return state

    @property
    def emcee_samples_full(self):
        """The whole chain [nsteps, nwalkers, ndim] in theta coordinates (core.py:2377).  Materialised on first access: at the
        reference's default 5e4 steps x 256 walkers it is a 1 GB copy out of HBM plus an un-scaling pass on the host (0.5 s, three
        times the sampling itself), which run_emcee should not charge to callers that never look at it."""
        if getattr(self, "_emcee_full", None) is None and getattr(self, "_emcee_full_src", None) is not None:
            sampler, t_add, t_mult = self._emcee_full_src
            self._emcee_full = (np.asarray(sampler.get_chain()) - t_add) / t_mult
            self._emcee_full_src = None
        if getattr(self, "_emcee_full", None) is None:
            raise AttributeError("emcee_samples_full: run_emcee has not been called")
        return self._emcee_full

    @emcee_samples_full.setter
    def emcee_samples_full(self, value):
        self._emcee_full, self._emcee_full_src = value, None

    def save(self):
        """Pickle the model (write-to-temp then rename, alabi/core.py:371-392)."""
        file = os.path.join(self.savedir, self.model_name)
        tmp = file + ".pkl.tmp"
        try:
            with open(tmp, "wb") as f:
                pickle.dump(self, f)
            os.rename(tmp, file + ".pkl")
        except Exception:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise

    def _seed(self):
        return int(self._rng.randint(0, 2 ** 31 - 1))

    # --------------------------------------------------------------------- data / scalers
    def _lnlike_fn(self, _theta):
        theta = self.theta_scaler.inverse_transform(_theta).flatten()
        y = np.asarray(self.true_log_likelihood(theta)).reshape(-1, 1)
        return self.y_scaler.transform(y).flatten()

    def theta(self):
        return self.theta_scaler.inverse_transform(self._theta)

    def y(self):
        return self.y_scaler.inverse_transform(self._y.reshape(-1, 1)).flatten()

    def refit_scalers(self, theta, y, theta_scaler=None, y_scaler=None):
        if theta_scaler is not None:
            self.theta_scaler = theta_scaler
        if y_scaler is not None:
            self.y_scaler = y_scaler
        # The identity scalers (``no_scaler``, the default) pass their input through: after their first fit the sklearn round trip
        # (parameter validation + check_array, ~0.15 ms per call) is skipped -- it was a tenth of a small active-learning iteration.
        def _identity(sc):
            return (getattr(sc, "func", None) is ut._ident and getattr(sc, "inverse_func", None) is ut._ident
                    and any(sc is f for f in self._scalers_fitted))
        if not hasattr(self, "_scalers_fitted"):
            self._scalers_fitted = []
        if _identity(self.theta_scaler):
            _theta = np.array(theta, dtype=np.float64)
        else:
            self.theta_scaler.fit(self.bounds.T)
            _theta = self.theta_scaler.transform(theta)
        if _identity(self.y_scaler):
            _y = np.array(y, dtype=np.float64).reshape(-1)
        else:
            _y = self.y_scaler.fit_transform(np.asarray(y).reshape(-1, 1)).flatten()
        for sc in (self.theta_scaler, self.y_scaler):
            if not any(sc is f for f in self._scalers_fitted):
                self._scalers_fitted.append(sc)
        for name, arr in (("theta_scaler", _theta), ("y_scaler", _y)):
            if np.any(np.isnan(arr)):
                raise ValueError(f"Refitted {name} produced NaN values!")
            if np.any(np.isinf(arr)):
                raise ValueError(f"Refitted {name} produced Inf values!")
        return _theta, _y

    def init_train(self, nsample=None, sampler="uniform", fname="initial_training_sample.npz"):
        if nsample is None:
            nsample = 50 * self.ndim
        theta = ut.prior_sampler(bounds=self.bounds, nsample=nsample, sampler=sampler, random_state=self._seed())
        y = np.array([self.true_log_likelihood(tt) for tt in theta], dtype=np.float64).reshape(-1, 1)
        for ii in range(len(y)):
            while not np.isfinite(y[ii, 0]):
                new_theta = ut.prior_sampler(bounds=self.bounds, nsample=1, sampler="uniform", random_state=self._seed())
                y[ii] = np.asarray(self.true_log_likelihood(new_theta[0])).reshape(-1)[0]
                theta[ii] = new_theta
        if self.cache:
            np.savez(f"{self.savedir}/{fname}", theta=theta, y=y)
        return theta, y

    def load_train(self, cache_file):
        sims = np.load(cache_file)
        theta, y = sims["theta"], sims["y"]
        if self.ndim != theta.shape[1]:
            raise ValueError(f"Dimension of bounds (n={self.ndim}) does not match dimension of training theta "
                             f"(n={theta.shape[1]})")
        return theta, y

    def _samples(self, n, sampler, file, default_name):
        if file is not None:
            path = file if os.path.exists(file) else f"{self.savedir}/{file}"
            try:
                theta, y = self.load_train(path)
                print(f"Loaded {len(theta)} samples from {path}.")
                return theta, y
            except Exception as e:  # noqa: BLE001
                print(f"Unable to reload {path} due to error: {e}. Computing new samples...")
                return self.init_train(nsample=n, sampler=sampler, fname=file)
        return self.init_train(nsample=n, sampler=sampler, fname=default_name)

    def init_samples(self, ntrain=100, ntest=0, sampler="uniform", train_file=None, test_file=None):
        theta, y = self._samples(ntrain, sampler, train_file, "initial_train_file_sample.npz")
        if ntest > 0:
            self.theta_test, self.y_test = self._samples(ntest, sampler, test_file, "initial_test_sample.npz")
            self.ntest = len(self.theta_test)
        else:
            self.theta_test, self.y_test, self.ntest = [], [], 0
        self.theta_train = theta
        self.y_train = y
        self.ninit_train = len(theta)
        self.ntrain = self.ninit_train
        self.nactive = 0

    # ------------------------------------------------------------------ hyper-parameters
    def set_hyperparam_prior_bounds(self):
        """Box for the hyper-parameter search (alabi/core.py:628-662).  The amplitude box is built from
        the LINEAR var(y) although the parameter is a log -- reference behaviour, kept."""
        pnames = self.param_names_optimized if self.uniform_scales else list(self.param_names_full)
        hp = [[None, None] for _ in pnames]
        if self.fit_mean:
            m, s = np.mean(self._y), np.std(self._y)
            hp[pnames.index("mean:value")] = [m - s, m + s]
        if self.fit_amp:
            v = np.var(self._y)
            hp[pnames.index(f"{self.kernel_amp_key}:log_constant")] = [v * 10 ** self.gp_amp_rng[0], v * 10 ** self.gp_amp_rng[1]]
        if self.fit_white_noise:
            hp[pnames.index("white_noise:value")] = [self.white_noise - 3, self.white_noise + 3]
        for ii, name in enumerate(pnames):       # RationalQuadraticKernel's shape parameter (the reference leaves it unbounded)
            if name.endswith(":log_alpha"):
                hp[ii] = [-3.0, 3.0]
        if self.uniform_scales:
            hp[pnames.index(f"{self.kernel_scale_key}:metric:log_M")] = list(self.gp_scale_rng)
        else:
            for ii in range(self.ndim):
                hp[pnames.index(f"{self.kernel_scale_key}:metric:log_M_{ii}_{ii}")] = list(self.gp_scale_rng)
        self.hp_bounds = np.array(hp)
        self.gp_hyper_prior = partial(ut.lnprior_uniform, bounds=self.hp_bounds)

    def expand_hyperparameter_vector(self, optimized_params):
        if optimized_params is None:
            raise ValueError("optimized_params cannot be None")
        if not self.uniform_scales:
            return optimized_params
        full = np.ones(len(self.param_names_full))
        names_f, names_o = list(self.param_names_full), self.param_names_optimized
        for key in ("mean:value", f"{getattr(self, 'kernel_amp_key', 'kernel:k1')}:log_constant", "white_noise:value"):
            if key in names_f and key in names_o:
                full[names_f.index(key)] = optimized_params[names_o.index(key)]
        for ii in range(self.ndim):
            full[names_f.index(f"{self.kernel_scale_key}:metric:log_M_{ii}_{ii}")] = \
                optimized_params[names_o.index(f"{self.kernel_scale_key}:metric:log_M")]
        return np.array(full)

    def set_hyperparameter_vector(self, tmp_gp, optimized_params):
        if optimized_params is None:
            raise ValueError("optimized_params cannot be None. Cannot set hyperparameters.")
        tmp_gp.set_parameter_vector(self.expand_hyperparameter_vector(optimized_params))
        return tmp_gp

    def get_hyperparameter_dict(self, gp):
        d = gp.get_parameter_dict()
        if self.uniform_scales:
            d[f"{self.kernel_scale_key}:metric:log_M"] = d.pop(f"{self.kernel_scale_key}:metric:log_M_0_0")
            for ii in range(1, self.ndim):
                del d[f"{self.kernel_scale_key}:metric:log_M_{ii}_{ii}"]
        return d

    def get_hyperparameter_vector(self, gp):
        return np.fromiter(self.get_hyperparameter_dict(gp).values(), dtype=float)

    # --------------------------------------------------------------------------- init_gp
    def init_gp(self, kernel="ExpSquaredKernel", fit_amp=True, fit_mean=True, fit_white_noise=True, white_noise=-12,
                gp_scale_rng=[-2, 2], gp_amp_rng=[-1, 1], uniform_scales=False, overwrite=False,
                theta_scaler=ut.no_scaler, y_scaler=ut.no_scaler, gp_opt_method="l-bfgs-b", gp_nopt=3,
                optimizer_kwargs={"maxiter": 100, "xatol": 1e-4, "fatol": 1e-3, "adaptive": True},
                hyperopt_method="cv", regularize=True, amp_0=1.0, mu_0=1.0, sigma_0=2.0, cv_folds=5,
                cv_scoring="mse", cv_n_candidates=100, cv_stage2_candidates=50, cv_stage2_width=0.5,
                cv_stage3_candidates=25, cv_stage3_width=0.25, cv_weighted_factor=1.0, multi_proc=True):
        if hasattr(self, "gp") and not overwrite:
            raise AssertionError("GP kernel already assigned. Use overwrite=True to re-assign the kernel.")
        if kernel not in _KERNELS:
            raise ValueError(f"Kernel '{kernel}' is not a valid option. Valid options: ExpSquaredKernel, "
                             "Matern32Kernel, Matern52Kernel, RationalQuadraticKernel")
        self.fit_amp, self.fit_mean, self.fit_white_noise = fit_amp, fit_mean, fit_white_noise
        self.white_noise = white_noise
        self.uniform_scales = uniform_scales
        self.gp_opt_method = gp_opt_method
        self.gp_nopt = gp_nopt
        self.opt_gp_kwargs = {"hyperopt_method": hyperopt_method, "regularize": regularize, "amp_0": amp_0,
                              "mu_0": mu_0, "sigma_0": sigma_0, "optimizer_kwargs": optimizer_kwargs,
                              "cv_folds": cv_folds, "cv_scoring": cv_scoring, "cv_n_candidates": cv_n_candidates,
                              "cv_stage2_candidates": cv_stage2_candidates, "cv_stage2_width": cv_stage2_width,
                              "cv_stage3_candidates": cv_stage3_candidates, "cv_stage3_width": cv_stage3_width,
                              "cv_weighted_factor": cv_weighted_factor, "multi_proc": multi_proc}
        self.theta_scaler = theta_scaler
        self.theta_scaler.fit(self.bounds.T)
        self._bounds = self.theta_scaler.transform(self.bounds.T).T
        self._prior_sampler = partial(ut.prior_sampler, bounds=self._bounds, sampler="uniform", random_state=None)
        self.y_scaler = y_scaler
        self._theta, self._y = self.refit_scalers(self.theta_train, self.y_train)
        if self.ntest > 0:
            self._theta_test = self.theta_scaler.transform(self.theta_test)
            self._y_test = self.y_scaler.transform(np.asarray(self.y_test).reshape(-1, 1)).flatten()
        self._theta_train, self._y_train = self._theta, self._y
        self.training_results = {k: [] for k in (
            "iteration", "gp_hyperparameters", "gp_hyperparameter_opt_iteration", "gp_hyperparam_opt_time",
            "training_mse", "test_mse", "training_scaled_mse", "test_scaled_mse", "gp_kl_divergence",
            "gp_train_time", "obj_fn_opt_time", "acquisition_optimizer_niter")}
        self.gp_scale_rng, self.gp_amp_rng = gp_scale_rng, gp_amp_rng
        self.kernel_name = kernel
        gp = None
        for attempt in range(1, 11):  # reference: up to 10 random initial length scales (core.py:980-1048)
            log_ls = self._rng.uniform(min(gp_scale_rng), max(gp_scale_rng), self.ndim)
            self.kernel = {"name": kernel, "log_M": log_ls.copy(), "log_constant": 0.0, "log_alpha": 1.0}
            gp = gp_utils.configure_gp(self._theta, self._y, self.kernel, fit_amp=fit_amp, fit_mean=fit_mean,
                                       fit_white_noise=fit_white_noise, white_noise=white_noise)
            if gp is not None:
                if self.verbose:
                    print(f"Successfully initialized GP on attempt {attempt}")
                break
            print("Warning: configure_gp returned None. Retrying with new initial scale length...")
        if gp is None:
            raise RuntimeError(f"Failed to initialize GP after 10 attempts. Check your data, kernel choice, and "
                               f"scale bounds. Current settings: kernel={kernel}, gp_scale_rng={gp_scale_rng}")
        self.gp = gp
        self.param_names_full = self.gp.get_parameter_names(include_frozen=False)
        self.kernel_scale_key = [x for x in self.param_names_full if "metric:log_M" in x][0].split(":metric:log_M")[0]
        self.param_names_optimized = []
        if fit_mean:
            self.param_names_optimized.append("mean:value")
        if fit_amp:      # without the amplitude the model has no log_constant parameter at all (core.py:1057-1059)
            self.kernel_amp_key = [x for x in self.param_names_full if "log_constant" in x][0].split(":log_constant")[0]
            self.param_names_optimized.append(f"{self.kernel_amp_key}:log_constant")
        if fit_white_noise:
            self.param_names_optimized.append("white_noise:value")
        if uniform_scales:
            self.param_names_optimized.append(f"{self.kernel_scale_key}:metric:log_M")
        else:
            self.param_names_optimized += [f"{self.kernel_scale_key}:metric:log_M_{ii}_{ii}" for ii in range(self.ndim)]
        self.hp_length_indices = [i for i, n in enumerate(self.param_names_full) if "metric:log_m" in n.lower()]
        self.hp_other_indices = [i for i, n in enumerate(self.param_names_full) if "metric:log_m" not in n.lower()]
        if uniform_scales:
            self.hp_length_index = [self.param_names_optimized.index(f"{self.kernel_scale_key}:metric:log_M")]
        self.initial_gp_hyperparameters = self.get_hyperparameter_vector(self.gp)
        self.gp, _ = self._opt_gp(**self.opt_gp_kwargs)
        if self.ntest > 0:
            _yt = self.gp.predict(self._y, self._theta_test, return_cov=False, return_var=False)
            yt = self.y_scaler.inverse_transform(_yt.reshape(-1, 1)).flatten()
            yt_true = self.y_scaler.inverse_transform(self._y_test.reshape(-1, 1)).flatten()
            return np.mean((yt_true - yt) ** 2)
        return None

    def _new_gp(self, _y):
        log_const = np.log(np.var(_y) / self.ndim) if self.fit_amp else self.kernel.get("log_constant", 0.0)
        return HipGP(self.ndim, mean=np.median(_y), white_noise=self.white_noise, log_constant=log_const,
                     log_M=self.kernel["log_M"], fit_mean=self.fit_mean, fit_white_noise=self.fit_white_noise,
                     kernel=self.kernel.get("name", "ExpSquaredKernel"), log_alpha=self.kernel.get("log_alpha", 1.0),
                     fit_amp=self.fit_amp)

    def _fit_gp(self, _theta=None, _y=None, hyperparameters=None):
        """New GP on (_theta, _y) with the carried hyper-parameter vector, factorised (core.py:1097-1160)."""
        _theta = self._theta if _theta is None else _theta
        _y = self._y if _y is None else _y
        t0 = time.time()
        self.set_hyperparam_prior_bounds()
        if not np.all(np.isfinite(_theta)):
            raise ValueError("_theta contains NaN or Inf values")
        if not np.all(np.isfinite(_y)):
            raise ValueError(f"_y contains NaN or Inf values: {_y[~np.isfinite(_y)]}")
        y_var = np.var(_y)
        if not np.isfinite(np.median(_y)):
            raise ValueError("median(_y) is not finite")
        if not np.isfinite(y_var) or y_var == 0:
            raise ValueError(f"var(_y) is not finite or zero: {y_var}")
        gp = self._new_gp(_y)
        if hyperparameters is not None and not np.all(np.isfinite(np.atleast_1d(hyperparameters))):
            print("Warning: Hyperparameters contain NaN or Inf. Reoptimizing hyperparameters from scratch...")
            gp, _ = self._opt_gp(**self.opt_gp_kwargs, _theta=_theta, _y=_y)
            if not np.all(np.isfinite(gp.get_parameter_vector())):
                raise ValueError("Reoptimized GP still has invalid parameters")
            return gp, time.time() - t0
        gp = self.set_hyperparameter_vector(gp, hyperparameters)
        # one more training point, same kernel hyper-parameters: extend the previous factor instead of refactorising
        gp.compute_from(getattr(self, "gp", None), _theta)
        return gp, time.time() - t0

    def _opt_gp(self, hyperopt_method="ml", regularize=True, amp_0=1.0, mu_0=1.0, sigma_0=2.0,
                optimizer_kwargs={"maxiter": 100, "xatol": 1e-4, "fatol": 1e-3, "adaptive": True},
                cv_folds=5, cv_scoring="mse", cv_n_candidates=20, multi_proc=True, cv_stage2_candidates=None,
                cv_stage2_width=0.5, cv_stage3_candidates=None, cv_stage3_width=0.2,
                cv_weighted_mse_method="exponential", cv_weighted_factor=1.0, _theta=None, _y=None,
                theta_scaler=None, y_scaler=None):
        """Hyper-parameter selection: marginal likelihood ("ml") or staged k-fold CV ("cv") (core.py:1163-1403)."""
        t0 = time.time()
        _theta = self._theta if _theta is None else _theta
        _y = self._y if _y is None else _y
        if hyperopt_method.lower() not in ("ml", "cv"):
            print(f"Invalid method '{hyperopt_method}'. Must be 'ml' or 'cv'. Defaulting to 'ml'.")
            hyperopt_method = "ml"
        self.set_hyperparam_prior_bounds()
        op_gp = None
        if hyperopt_method.lower() == "ml":
            gp = self.gp
            gp.compute(_theta)

            def make_objective(gp):
                """(nll, grad_nll) bound to one GP handle: restarts run concurrently on their own copies."""
                def nll(p_opt):
                    p = self.expand_hyperparameter_vector(p_opt)
                    self.set_hyperparameter_vector(gp, p_opt)
                    v = -gp.log_likelihood(_y, quiet=True)
                    if regularize:
                        v += gp_utils.regularization_term(p, self.hp_length_indices, amp_0=amp_0, mu_0=mu_0, sigma_0=sigma_0)
                    return v if np.isfinite(v) else 1e25

                def grad_nll(p_opt):
                    """-d logL/dp from the device's analytic gradient (+ the regulariser's), assembled exactly as the
                    reference does (core.py:1255-1277): with uniform_scales the shared length-scale entry receives the MEAN
                    of the per-dimension gradients."""
                    p = self.expand_hyperparameter_vector(p_opt)
                    self.set_hyperparameter_vector(gp, p_opt)
                    try:
                        grad_lnlike = -gp.grad_log_likelihood(_y, quiet=True)
                    except (np.linalg.LinAlgError, RuntimeError):
                        return np.zeros(len(p_opt))
                    if not np.all(np.isfinite(grad_lnlike)):
                        return np.zeros(len(p_opt))
                    if self.uniform_scales:
                        gll = np.zeros(len(p_opt))
                        gll[self.hp_length_index] = np.mean(grad_lnlike[self.hp_length_indices])
                        gll[self.hp_other_indices] = grad_lnlike[self.hp_other_indices]
                    else:
                        gll = grad_lnlike
                    if regularize:
                        reg_grad = gp_utils.regularization_gradient(p, self.hp_length_indices, amp_0=amp_0, mu_0=mu_0,
                                                                    sigma_0=sigma_0)
                        if self.uniform_scales:
                            gll[self.hp_length_index] += np.mean(reg_grad[self.hp_length_indices])
                        else:
                            gll = gll + reg_grad
                    return gll
                return nll, grad_nll

            nll, grad_nll = make_objective(gp)

            use_grad = self.gp_opt_method in ("newton-cg", "l-bfgs-b")
            opts = dict(optimizer_kwargs)
            if self.gp_opt_method == "l-bfgs-b":
                opts = {k: v for k, v in opts.items() if k in ("maxiter", "ftol", "gtol", "maxcor", "maxfun", "maxls")}

            def _run(x0, f=nll, g=grad_nll):
                return op.minimize(fun=f, x0=x0, jac=g if use_grad else None, method=self.gp_opt_method,
                                   bounds=self.hp_bounds, options=opts)

            current = self.get_hyperparameter_vector(gp)
            if self.gp_nopt <= 1:
                res = _run(current)
            else:
                p0 = ut.prior_sampler(bounds=self.hp_bounds, nsample=self.gp_nopt, sampler="lhs", random_state=self._seed())
                p0[0] = current
                nthreads = max(1, min(self.gp_nopt, int(os.environ.get("ALABI_ML_THREADS", self.gp_nopt))))
                if nthreads <= 1:
                    res = min((_run(p) for p in p0), key=lambda r: r.fun)
                else:
                    # The restarts are independent: each gets its own GP handle, HIP stream and host thread (every objective call
                    # is a chain of small launches with read-backs that fills only part of the chip; ctypes releases the GIL).
                    # The reference runs them one after the other (core.py:1287-1305); same optimum, ties broken by start order.
                    from concurrent.futures import ThreadPoolExecutor

                    def _restart(x0):
                        g_i = copy.deepcopy(gp)
                        with torch.cuda.stream(torch.cuda.Stream()):
                            g_i.compute(_theta)
                            f_i, d_i = make_objective(g_i)
                            r = _run(x0, f_i, d_i)
                            torch.cuda.current_stream().synchronize()
                        return r

                    torch.cuda.current_stream().synchronize()
                    with ThreadPoolExecutor(max_workers=nthreads) as pool:
                        results = list(pool.map(_restart, list(p0)))
                    res = min(results, key=lambda r: r.fun)
            best = res.x
            if not (np.all(np.isfinite(best)) and np.isfinite(res.fun) and res.fun < 1e25):
                # e.g. the incumbent lies outside the reference's amplitude box (built from the linear var(y),
                # core.py:654) and every point inside it overflows: keep the incumbent, as the CV branch does
                print("Warning: ML hyper-parameter search found no valid point; keeping current hyperparameters.")
                best = current
            op_gp = self.set_hyperparameter_vector(gp, best)
            if not op_gp.compute(_theta, quiet=True):
                op_gp = self.set_hyperparameter_vector(gp, current)
                op_gp.compute(_theta)
            if self.verbose:
                print(f"GP ML fit: -logL(+reg) {nll(current):.4f} -> {nll(best):.4f} in {res.nit} iterations")
                self.set_hyperparameter_vector(gp, best)
                gp.compute(_theta)
        else:
            if self.verbose:
                print(f"\nOptimizing GP hyperparameters using {cv_folds}-fold cross-validation...")
            try:
                cands = ut.prior_sampler(bounds=self.hp_bounds, nsample=cv_n_candidates, sampler="lhs",
                                         random_state=self._seed())
                if hasattr(self, "gp"):
                    cands[0] = self.get_hyperparameter_vector(self.gp)
                if self.uniform_scales:
                    cands = np.array([self.expand_hyperparameter_vector(c) for c in cands])
                # several ranks (one process per GPU, torch.distributed initialised) and multi_proc: the candidates of every
                # stage are dealt over the ranks -- the reference maps them over its process pool (gp_utils.py:640-700) -- and the
                # partial score vectors (+inf elsewhere) are combined with one MIN all-reduce; every rank then picks the same winner
                from . import dist as adist
                rank, world = adist.world_info()
                shard_kw = {}
                if world > 1 and multi_proc:
                    shard_kw = dict(ranks=(rank, world), reduce_scores=lambda sc: adist.allreduce_array(sc, "min"))
                op_gp = gp_utils.optimize_gp_kfold_cv(
                    self.gp, _theta, _y, cands, self.y_scaler, k_folds=cv_folds, scoring=cv_scoring,
                    stage2_candidates=cv_stage2_candidates, stage2_width=cv_stage2_width,
                    stage3_candidates=cv_stage3_candidates, stage3_width=cv_stage3_width,
                    weighted_mse_method=cv_weighted_mse_method, weighted_mse_factor=cv_weighted_factor,
                    verbose=self.verbose, random_state=self._seed(), **shard_kw)
            except Exception as e:  # noqa: BLE001
                print(f"Warning: CV hyperparameter optimization failed: {e}")
                op_gp = None
        if op_gp is None:
            if hasattr(self, "gp"):
                op_gp = self.gp
                op_gp.compute(_theta)
            else:
                op_gp = self._new_gp(_y)
                op_gp.compute(_theta)
        timing = time.time() - t0
        self.training_results["gp_hyperparam_opt_time"].append(timing)
        return op_gp, timing

    # ------------------------------------------------------------------ surrogate likelihood
    def eval_gp_at_iteration(self, iter, return_var=False):
        """predict-callable of the GP as it stood at active-learning iteration ``iter`` (core.py:1406-1443).

        The reference re-factorises on every call; here the current GP is reused for iter == -1 (it already
        holds the latest data and hyper-parameters) and other iterations are factorised once and memoised."""
        res = self.training_results
        n_it = len(res["iteration"])
        if iter == -1 or iter == n_it:
            if n_it > 0:
                want = np.asarray(res["gp_hyperparameters"][-1])
                if not np.array_equal(want, self.gp.get_parameter_vector()):
                    self.gp.set_parameter_vector(want)
                    self.gp.compute(self._theta)
            gp_iter, _y_cond = self.gp, self._y
        else:
            if iter == 0 or n_it == 0:
                n_cond = self.ninit_train
                hp = res["gp_hyperparameters"][0] if n_it > 0 else self.initial_gp_hyperparameters
            elif 0 < iter < n_it:
                n_cond = self.ninit_train + iter
                hp = res["gp_hyperparameters"][iter]
            else:
                raise ValueError(f"Iteration {iter} exceeds available training iterations ({res['iteration'][-1]}).")
            memo = self.__dict__.setdefault("_gp_iter_memo", {})
            key = (iter, self.ntrain)
            if key not in memo:
                g = self._new_gp(self._y[:n_cond])
                g.set_parameter_vector(hp)
                g.compute(self._theta[:n_cond])
                memo.clear()
                memo[key] = g
            gp_iter, _y_cond = memo[key], self._y[:n_cond]

        def gp_predict(x):
            x = np.atleast_2d(x)
            if x.shape[1] != self.ndim and x.size == self.ndim:
                x = x.reshape(1, -1)
            return gp_iter.predict(_y_cond, x, return_var=return_var, return_cov=False)

        return gp_predict

    def surrogate_log_likelihood(self, theta_xs, iter=-1, return_var=False):
        """GP surrogate of the log-likelihood at theta_xs ([d] -> scalar, [M,d] -> array) (core.py:1446-1508)."""
        theta_xs = np.asarray(theta_xs)
        one = theta_xs.ndim == 1
        if one:
            theta_xs = theta_xs.reshape(1, -1)
        elif theta_xs.ndim != 2:
            raise ValueError(f"theta_xs must be 1D or 2D array, got {theta_xs.ndim}D")
        _t = self.theta_scaler.transform(theta_xs)
        if hasattr(self, "training_results") and len(self.training_results["iteration"]) > 0:
            gp_ii = self.eval_gp_at_iteration(iter, return_var=return_var)
        else:
            gp_ii = lambda x: self.gp.predict(self._y, x, return_var=return_var, return_cov=False)  # noqa: E731
        if not return_var:
            yp = self.y_scaler.inverse_transform(gp_ii(_t).reshape(-1, 1)).flatten()
            return yp[0] if one else yp
        _yp, _vp = gp_ii(_t)
        yp = self.y_scaler.inverse_transform(_yp.reshape(-1, 1)).flatten()
        vp = self.y_scaler.inverse_transform(_vp.reshape(-1, 1)).flatten()   # reference quirk (core.py:1502)
        return (yp[0], vp[0]) if one else (yp, vp)

    def surrogate_likelihood(self, theta_xs):
        """Predictive probability (not log) of the GP at theta_xs (core.py:1511-1533)."""
        return np.exp(self.surrogate_log_likelihood(theta_xs))

    def create_cached_surrogate_likelihood(self, iter=-1, return_var=False):
        """Factorise once, return a picklable callable (core.py:1535-1584)."""
        if hasattr(self, "training_results") and len(self.training_results["iteration"]) > 0:
            n_cond = len(self._theta) if iter == -1 else self.ninit_train + iter
            hp = self.training_results["gp_hyperparameters"][-1]
        else:
            n_cond = len(self._theta)
            hp = self.gp.get_parameter_vector()
        _tc, _yc = self._theta[:n_cond], self._y[:n_cond]
        gp_iter = gp_utils.configure_gp(_tc, _yc, self.kernel, fit_amp=self.fit_amp, fit_mean=self.fit_mean,
                                        fit_white_noise=self.fit_white_noise, white_noise=self.white_noise,
                                        hyperparameters=hp)
        if gp_iter is None:
            raise np.linalg.LinAlgError("create_cached_surrogate_likelihood: GP factorisation failed")
        return CachedSurrogateLikelihood(gp_iter, _yc, self.theta_scaler, self.y_scaler, self.ndim, return_var=return_var)

    # ---------------------------------------------------------------------- active learning
    def find_next_point(self, nopt=3, optimizer_kwargs={}):
        """Next training point = arg-min of the acquisition function (core.py:1587-1667).

        obj_opt_method "scan" (default here): ``ncand`` uniform candidates in the scaled box are scored in
        one batched HIP pass (predict mean+variance -> utility -> arg-min), then ``refine`` zoom stages of
        ``nrefine`` candidates each around the ``ntop`` best; then the incumbent is polished by
        a projected L-BFGS with the closed-form GPU gradient (``optimizer_kwargs={"polish": maxiter}``, default 30, 0 = off;
        ``"polish_method": "scipy"`` runs scipy's L-BFGS-B around the same evaluations instead).  Any scipy method name keeps
        the reference's multistart local optimisation with one GP prediction per objective call."""
        t0 = time.time()
        y_best = float(np.max(self._y))
        method = str(self.obj_opt_method).lower()
        kw = dict(optimizer_kwargs or {})
        if method == "scan":
            ncand = int(kw.get("ncand", 16384))
            gen = torch.Generator(device=_dev())
            gen.manual_seed(self._seed())
            lo = torch.as_tensor(self._bounds[:, 0], device=_dev())
            hi = torch.as_tensor(self._bounds[:, 1], device=_dev())
            cand = lo + (hi - lo) * torch.rand((ncand, self.ndim), dtype=torch.float64, device=_dev(), generator=gen)
            nref, nper, ntop = int(kw.get("refine", 4)), int(kw.get("nrefine", 4096)), int(kw.get("ntop", 32))
            from . import dist as adist
            rank, world = adist.world_info()
            if world > 1 and getattr(self, "allow_opt_multiproc", True) and ncand >= world:
                # several ranks: every rank holds the same GP and draws the same candidates; it scores ITS contiguous slice, one
                # (value, index) pair and the slices' best `ntop` are exchanged, and every rank goes on from the same incumbent
                # and the same centres (BASELINE config C5: 1e6 candidates over 8 GPUs).  The reference
                # spreads its scipy restarts over a process pool instead (alabi/utility.py:1030-1163, allow_opt_multiproc).
                b0, e0 = adist.slice_bounds(ncand, world, rank)
                out = ut.utility_scan(self.gp, self._y, cand[b0:e0], self._bounds, algorithm=self.algorithm, y_best=y_best,
                                      return_all=nref > 0)
                gi_local = b0 + int(out[2]) if out[2] >= 0 and np.isfinite(out[1]) else -1
                u_best, idx = adist.reduce_min_index(out[1] if gi_local >= 0 else np.inf, gi_local)
                _thetaN = cand[idx].cpu().numpy() if idx >= 0 else np.nan
                centers = None
                if idx >= 0 and nref > 0:
                    u_loc = torch.where(torch.isfinite(out[3]), out[3], torch.full_like(out[3], float("inf")))
                    kk = min(ntop, e0 - b0)
                    top = torch.topk(-u_loc, kk)
                    mine = np.full((ntop, 2), [np.inf, -1.0])
                    mine[:kk, 0] = (-top.values).cpu().numpy()
                    mine[:kk, 1] = (top.indices + b0).cpu().numpy()
                    allp = adist.allgather_rows(mine)
                    allp = allp[allp[:, 1] >= 0]
                    keep = allp[np.lexsort((allp[:, 1], allp[:, 0]))][:min(ntop, ncand)]      # ascending value, ties to the lower index
                    centers = cand[torch.as_tensor(keep[:, 1].astype(np.int64), device=_dev())]
            else:
                out = ut.utility_scan(self.gp, self._y, cand, self._bounds, algorithm=self.algorithm, y_best=y_best,
                                      return_all=nref > 0, best_on_device=True)
                _thetaN, u_best, idx = out[:3]              # (the incumbent stays on the device through the zoom stages)
                centers = None
                if idx >= 0 and nref > 0:
                    u_all = torch.nan_to_num(out[3], nan=float("inf"), posinf=float("inf"), neginf=float("inf"))
                    centers = cand[torch.topk(u_all, min(ntop, ncand), largest=False).indices]
            # Zoom stages (stand in for the reference's local optimiser, utility.py:1030-1163, at batched-scan cost):
            # Gaussian clouds of shrinking width around the best candidates so far, scored in one pass each.
            if idx >= 0 and nref > 0:
                # (few launches per stage: at small N a stage is launch latency -- one fused multiply-add for the cloud, one clamp
                # against precomputed inner bounds, nan_to_num instead of isfinite / full_like / where, topk on the values themselves)
                width = 0.08 * (hi - lo)
                lo_in, hi_in = lo + 1e-12 * (hi - lo), hi - 1e-12 * (hi - lo)
                for _ in range(nref):
                    rep = centers[torch.randint(0, centers.shape[0], (nper,), device=_dev(), generator=gen)]
                    cloud = torch.addcmul(rep, torch.randn((nper, self.ndim), dtype=torch.float64, device=_dev(), generator=gen), width)
                    cloud = torch.clamp(cloud, min=lo_in, max=hi_in)
                    cloud[0] = torch.as_tensor(_thetaN, device=_dev())          # the incumbent can only be improved on
                    o2 = ut.utility_scan(self.gp, self._y, cloud, self._bounds, algorithm=self.algorithm, y_best=y_best,
                                         return_all=True, best_on_device=True)
                    if o2[2] >= 0 and o2[1] <= u_best:
                        _thetaN, u_best = o2[0].clone(), o2[1]
                    u2 = torch.nan_to_num(o2[3], nan=float("inf"), posinf=float("inf"), neginf=float("inf"))
                    centers = cloud[torch.topk(u2, min(ntop, nper), largest=False).indices]
                    width = 0.3 * width
            if isinstance(_thetaN, torch.Tensor):
                _thetaN = _thetaN.cpu().numpy()
            # optional continuous polish of the incumbent: L-BFGS-B with the closed-form GPU gradient
            npolish = int(kw.get("polish", 30))
            if idx >= 0 and npolish > 0:
                th_p, u_p = ut.polish_point(self.gp, self._y, _thetaN, self._bounds, algorithm=self.algorithm, y_best=y_best,
                                            maxiter=npolish, method=str(kw.get("polish_method", "native")))
                if u_p < u_best:
                    _thetaN, u_best = th_p, u_p
            self.last_acquisition_value = float(u_best) if idx >= 0 else np.nan
            if idx < 0:
                _thetaN = np.nan
        else:
            predict_gp = lambda _x: self.gp.predict(self._y, _x, return_var=True)  # noqa: E731
            if self.algorithm == "jones":
                obj_fn = partial(self.utility, predict_gp=predict_gp, bounds=self._bounds, y_best=y_best)
            else:
                obj_fn = partial(self.utility, predict_gp=predict_gp, bounds=self._bounds)
            for k in ("ncand", "polish", "polish_method", "refine", "nrefine", "ntop"):
                kw.pop(k, None)
            grad_obj_fn = None           # analytic gradient on the GPU (reference: core.py:1618-1625 passes grad_utility)
            if getattr(self, "use_grad_opt", True) and getattr(self, "grad_utility", None) is not None:
                grad_obj_fn = partial(self.grad_utility, gp=self.gp, bounds=self._bounds)
            _thetaN, _ = ut.minimize_objective(obj_fn, bounds=self._bounds, nopt=nopt, ps=self._prior_sampler,
                                               method=self.obj_opt_method, options=kw or None, grad_obj_fn=grad_obj_fn)
        opt_timing = time.time() - t0
        if not np.all(np.isfinite(_thetaN)):
            print("Warning: Acquisition function optimization failed. Falling back to random sampling.")
            _thetaN = ut.prior_sampler(bounds=self._bounds, nsample=1, random_state=self._seed()).flatten()
        thetaN = self.theta_scaler.inverse_transform(np.asarray(_thetaN).reshape(1, -1))
        yN = np.asarray(self.true_log_likelihood(thetaN.flatten()), dtype=np.float64).reshape(-1)
        if not np.all(np.isfinite(thetaN)) or not np.all(np.isfinite(yN)):
            print(f"New point is not finite: theta={thetaN}, y={yN}")
            return None, None, opt_timing
        theta_prop = np.append(self.theta(), thetaN, axis=0)
        y_prop = np.append(self.y(), yN)
        _theta_prop, _y_prop = self.refit_scalers(theta_prop, y_prop)
        if _theta_prop.shape[0] != _y_prop.shape[0]:
            return None, None, opt_timing
        return _theta_prop, _y_prop, opt_timing

    def active_train(self, niter=100, algorithm="bape", gp_opt_freq=20, save_progress=False,
                     obj_opt_method="scan", nopt=5, optimizer_kwargs={}, use_grad_opt=True,
                     show_progress=True, allow_opt_multiproc=True, max_attempts=10):
        """Active-learning loop: pick a point, evaluate the true function, refit (core.py:1670-1865)."""
        self.algorithm = str(algorithm).lower()
        self.utility, self.grad_utility = ut.assign_utility(self.algorithm)
        if not use_grad_opt:
            self.grad_utility = None
        if self.algorithm not in ("bape", "agp", "jones"):
            self.algorithm = "bape"
        self.gp_opt_freq = gp_opt_freq
        self.obj_opt_method = obj_opt_method
        self.use_grad_opt = bool(use_grad_opt)
        self.allow_opt_multiproc = bool(allow_opt_multiproc)    # several ranks: shard the candidate scan of find_next_point
        res = self.training_results
        first_iter = res["iteration"][-1] if len(res["iteration"]) else 0
        if self.verbose:
            print(f"Running {niter} active learning iterations using {self.algorithm}...")
        for ii in range(1, niter + 1):
            attempts, success = 0, False
            while not success:
                _theta_prop, _y_prop, opt_timing = self.find_next_point(nopt=nopt, optimizer_kwargs=optimizer_kwargs)
                if _theta_prop is None or _y_prop is None:
                    attempts += 1
                    if attempts >= max_attempts:
                        raise RuntimeError(f"Failed to find a valid training point after {max_attempts} attempts. "
                                           "Check your likelihood function and training data for issues or "
                                           "increase max_attempts.")
                    continue
                self.gp, fit_gp_timing = self._fit_gp(_theta=_theta_prop, _y=_y_prop,
                                                      hyperparameters=self.gp.get_parameter_vector())
                success = True
            self._theta, self._y = _theta_prop, _y_prop
            self.__dict__.pop("_gp_iter_memo", None)
            if (ii + first_iter) % self.gp_opt_freq == 0:
                self.gp, _ = self._opt_gp(**self.opt_gp_kwargs)
                res["gp_hyperparameter_opt_iteration"].append(ii + first_iter)
                if save_progress:
                    self.save()
            y_now = var_y_now = None                        # (self.y() un-scales the whole training set: once per iteration)
            try:
                _yp = self.gp.predict(_y_prop, _theta_prop, return_cov=False, return_var=False)
                yp = self.y_scaler.inverse_transform(_yp.reshape(-1, 1)).flatten()
                y_now = self.y(); var_y_now = np.var(y_now)
                training_mse = np.mean((y_now - yp) ** 2)
                training_scaled_mse = training_mse / var_y_now
            except Exception as e:  # noqa: BLE001
                print(f"Warning: Error evaluating GP training error at iteration {ii + first_iter}: {e}")
                training_mse = training_scaled_mse = np.nan
            test_mse = test_scaled_mse = np.nan
            if self.ntest > 0:
                try:
                    _yt = self.gp.predict(self._y, self._theta_test, return_cov=False, return_var=False)
                    yt = self.y_scaler.inverse_transform(_yt.reshape(-1, 1)).flatten()
                    yt_true = self.y_scaler.inverse_transform(self._y_test.reshape(-1, 1)).flatten()
                    test_mse = np.mean((yt_true - yt) ** 2)
                    test_scaled_mse = test_mse / (var_y_now if var_y_now is not None else np.var(self.y()))
                except Exception as e:  # noqa: BLE001
                    print(f"Warning: Error evaluating GP test error at iteration {ii + first_iter}: {e}")
            res["iteration"].append(ii + first_iter)
            res["gp_hyperparameters"].append(self.gp.get_parameter_vector())
            res["training_mse"].append(training_mse)
            res["test_mse"].append(test_mse)
            res["training_scaled_mse"].append(training_scaled_mse)
            res["test_scaled_mse"].append(test_scaled_mse)
            res["gp_kl_divergence"].append(np.nan)
            res["gp_train_time"].append(fit_gp_timing)
            res["obj_fn_opt_time"].append(opt_timing)
            self.ntrain = len(self._theta)
            self.nactive = self.ntrain - self.ninit_train
        if self.cache:
            self.save()

    # -------------------------------------------------------------------------------- MCMC
    def lnprob(self, theta):
        """log-posterior = like_fn(theta) + prior_fn(theta) (core.py:2073-2100)."""
        if getattr(self, "like_fn_name", "surrogate") == "surrogate" and not hasattr(self, "gp"):
            raise NameError("GP has not been trained")
        if not hasattr(self, "prior_fn"):
            raise NameError("prior_fn has not been specified")
        if not hasattr(self, "like_fn"):
            self.like_fn = self.surrogate_log_likelihood
        theta = np.asarray(theta).reshape(1, -1)
        return self.like_fn(theta) + self.prior_fn(theta)

    def find_map(self, theta0=None, prior_fn=None, method="nelder-mead", nRestarts=15, options=None):
        """Maximum of like_fn + prior_fn (alabi/core.py:2103; called by run_emcee(opt_init=True), core.py:2290-2294).

        The reference declares this entry point and raises NotImplementedError("Not implemented.") in its body, so
        ``opt_init=True`` cannot run there; here it works: ``nRestarts`` local optimisations (scipy ``method``) of
        -lnprob from the best points of a batched scan of the posterior (surrogate evaluated on the GPU, 4096 prior
        draws) and from ``theta0`` if given.  Sets ``self.map_theta`` / ``self.map_lnprob`` and returns the walker start
        positions run_emcee passes to the sampler: [nwalkers, ndim] in a ball of 1e-3 of the box width around the MAP,
        clipped into the open box."""
        prior_fn = prior_fn if prior_fn is not None else getattr(self, "prior_fn", None)
        if prior_fn is None:
            prior_fn = partial(ut.lnprior_uniform, bounds=self.bounds)
        like_fn = getattr(self, "like_fn", None) or self.surrogate_log_likelihood
        lo, hi = self.bounds[:, 0].astype(float), self.bounds[:, 1].astype(float)

        def lnp(th):
            th = np.asarray(th, dtype=np.float64).reshape(1, -1)
            pr = float(np.asarray(prior_fn(th)).reshape(-1)[0])
            if not np.isfinite(pr):
                return -np.inf
            v = float(np.asarray(like_fn(th)).reshape(-1)[0]) + pr
            return v if np.isfinite(v) else -np.inf

        cand = ut.prior_sampler(bounds=self.bounds, nsample=4096, sampler="uniform", random_state=self._seed())
        if like_fn == self.surrogate_log_likelihood:     # one batched GPU predict
            like_c = np.asarray(like_fn(cand), dtype=np.float64).reshape(-1)
        else:
            cand = cand[:256]
            like_c = np.array([float(np.asarray(like_fn(c.reshape(1, -1))).reshape(-1)[0]) for c in cand])
        post_c = like_c + np.array([float(np.asarray(prior_fn(c.reshape(1, -1))).reshape(-1)[0]) for c in cand])
        post_c = np.where(np.isfinite(post_c), post_c, -np.inf)
        starts = [cand[i] for i in np.argsort(-post_c)[:max(int(nRestarts), 1)]]
        if theta0 is not None:
            starts.insert(0, np.asarray(theta0, dtype=np.float64).reshape(-1))
        best_x, best_f = starts[0], lnp(starts[0])
        eps = 1e-9 * (hi - lo)
        use_bounds = method.lower() in ("nelder-mead", "l-bfgs-b", "tnc", "powell", "slsqp", "trust-constr")
        for x0 in starts[:max(int(nRestarts), 1)]:
            try:
                res = op.minimize(lambda x: (lambda v: -v if np.isfinite(v) else 1e25)(lnp(x)), x0, method=method, options=options,
                                  bounds=list(zip(lo + eps, hi - eps)) if use_bounds else None)
            except Exception:  # noqa: BLE001
                continue
            f = lnp(res.x)
            if np.isfinite(f) and f > best_f:
                best_x, best_f = np.asarray(res.x, dtype=np.float64), f
        self.map_theta, self.map_lnprob = best_x, best_f
        nw = int(getattr(self, "nwalkers", 10 * self.ndim))
        ball = best_x + 1e-3 * (hi - lo) * self._rng.standard_normal((nw, self.ndim))
        return np.minimum(np.maximum(ball, lo + 1e-9 * (hi - lo)), hi - 1e-9 * (hi - lo))

    def _y_unscale_kind(self):
        """How y_scaler.inverse_transform acts on a GP mean: ("affine", slope, offset), ("nlog",) / ("log",) for the two
        non-affine scalers the reference ships (alabi/utility.py:62-71), or None (anything else)."""
        y_lo, y_hi = float(np.min(self._y)), float(np.max(self._y))
        aff = _affine_map(self.y_scaler.inverse_transform, np.array([[y_lo - 1.0, y_hi + 1.0]]))
        if aff is not None and aff[0][0] > 0:
            return ("affine", float(aff[0][0]), float(aff[1][0]))
        try:
            probe = np.linspace(y_lo - 0.5, y_hi + 0.5, 7).reshape(-1, 1)
            got = np.asarray(self.y_scaler.inverse_transform(probe), dtype=np.float64).reshape(-1)
            p10 = 10.0 ** probe.reshape(-1)
            if np.allclose(got, -p10, rtol=1e-12, atol=0.0):
                return ("nlog",)
            if np.allclose(got, p10, rtol=1e-12, atol=0.0):
                return ("log",)
        except Exception:  # noqa: BLE001
            pass
        return None

    def run_emcee(self, like_fn=None, prior_fn=None, nwalkers=None, nsteps=int(5e4), sampler_kwargs={}, run_kwargs={},
                  opt_init=False, multi_proc=True, prior_fn_comment=None, burn=None, thin=None, samples_file=None,
                  min_ess=int(1e4)):
        """Ensemble MCMC on the posterior like_fn + prior_fn, on the GPU (core.py:2108-2414).

        * Fused path (the reference's defaults and the priors it ships): like_fn None / "surrogate" / "gp", prior_fn None,
          ``partial(lnprior_uniform, bounds=...)`` or ``partial(lnprior_normal, bounds=..., data=...)``, affine theta
          scaler, y scaler affine or one of the shipped ``nlog_scaler`` / ``log_scaler``: the whole log-probability is
          evaluated inside the ensemble kernels.
        * Any other ``prior_fn`` callable (core.py:2253-2280; docstring example :2236-2239): the device proposes and
          evaluates the surrogate part of every half step, the host adds ``prior_fn(theta)`` per proposal, the device
          does the accept test (alabi_ens_propose / alabi_ens_accept).
        * ``like_fn="true"`` or a callable, or scalers that are neither of the above: the same split with the
          likelihood evaluated on the host as well (walkers then move in the original theta coordinates).
        ``opt_init=True`` starts the walkers around ``find_map()``.

        Several GPUs (the reference's parallel axis is a process pool handed to emcee, core.py:2300, :2322, built at :349-369;
        ``ncore`` / ``pool_method`` have no meaning here): run one process per GPU under ``torch.distributed`` (e.g.
        ``python -m torch.distributed.run --nproc-per-node 8 script.py``, backend "nccl" = RCCL) and have every rank make the
        same calls.  With ``multi_proc=True`` and more than one rank
          * default -- REPLICAS: every rank runs its own ensemble of ``nwalkers`` walkers (sampler seed + rank, its own start
            points), no communication inside the run; after each run the flattened samples of all ranks are concatenated on
            every rank (``emcee_samples`` is identical everywhere) and ``min_ess`` counts the samples of all ranks;
          * ``sampler_kwargs={"shard": True}`` -- ONE ensemble whose active half is partitioned over the ranks, an RCCL
            all-gather of the new walker rows per half step (alabi_amd.dist.ShardedRun); every rank holds the same chain.
        Files (the .npz of samples, the cached model) are written by rank 0 only."""
        from . import dist as adist
        rank, world = adist.world_info()
        sampler_kwargs = dict(sampler_kwargs)
        shard = bool(sampler_kwargs.pop("shard", False)) and bool(multi_proc) and world > 1
        replicas = bool(multi_proc) and world > 1 and not shard
        # ---- likelihood
        like_host = None                          # host callable on theta [n,d] -> [n], or None for the device surrogate
        if like_fn is None or (isinstance(like_fn, str) and like_fn.lower() in ("surrogate", "gp")):
            self.like_fn_name = "surrogate"
            self.like_fn = self.surrogate_log_likelihood
            if not hasattr(self, "gp"):
                raise NameError("GP has not been trained")
        elif isinstance(like_fn, str) and like_fn.lower() == "true":
            self.like_fn_name = "true"
            self.like_fn = self.true_log_likelihood
            like_host = self.like_fn
        elif callable(like_fn):
            self.like_fn_name = "likelihood"
            self.like_fn = like_fn
            like_host = like_fn
        else:
            raise ValueError("like_fn must be None, 'surrogate', 'gp', 'true' or a callable")
        # ---- prior: None (uniform box) or one of the two shipped priors fuse into the kernel; anything else is a host call
        prior_bounds, prior_data, prior_host = None, None, None
        if prior_fn is not None:
            f = getattr(prior_fn, "func", None)
            kwp = dict(getattr(prior_fn, "keywords", None) or {})
            argp = tuple(getattr(prior_fn, "args", ()) or ())
            name = getattr(f, "__name__", "")
            if name == "lnprior_uniform" and f is ut.lnprior_uniform and ("bounds" in kwp or len(argp) >= 1):
                prior_bounds = kwp.get("bounds", argp[0] if argp else None)
            elif name == "lnprior_normal" and f is ut.lnprior_normal and (("bounds" in kwp and "data" in kwp) or len(argp) >= 2):
                prior_bounds = kwp.get("bounds", argp[0] if argp else None)
                prior_data = kwp.get("data", argp[1] if len(argp) > 1 else None)
            else:
                prior_host = prior_fn
        # ---- scalers
        t_aff = _affine_map(self.theta_scaler.transform, self.bounds) if hasattr(self, "theta_scaler") else \
            (np.ones(self.ndim), np.zeros(self.ndim))
        y_kind = self._y_unscale_kind() if (like_host is None) else ("affine", 1.0, 0.0)
        if like_host is None and (t_aff is None or y_kind is None):
            # exotic scalers: the surrogate is evaluated through surrogate_log_likelihood (batched GPU predict) on the host side
            like_host = self.surrogate_log_likelihood
        if like_host is not None:
            t_aff = (np.ones(self.ndim), np.zeros(self.ndim))     # walkers move in theta itself
            y_kind = ("affine", 1.0, 0.0)
        t_mult, t_add = t_aff                      # scaled = t_mult * theta + t_add, per dimension
        logp_affine = (y_kind[1], y_kind[2]) if y_kind[0] == "affine" else (1.0, 0.0)
        logp_map = None if y_kind[0] == "affine" else y_kind[0]
        box = self.bounds if prior_bounds is None else np.asarray(prior_bounds, dtype=np.float64).reshape(self.ndim, 2)
        _box = np.sort(box * t_mult[:, None] + t_add[:, None], axis=1)          # the prior box in scaled coordinates
        normal_prior = None
        if prior_data is not None:
            pm = np.array([np.nan if dd[0] is None else float(dd[0]) for dd in prior_data])
            ps = np.array([np.nan if dd[0] is None else float(dd[1]) for dd in prior_data])
            # N(m, s) on theta_k is N(mult m + add, |mult| s) on the scaled coordinate; the density stays the theta-space one,
            # so log|mult| per normal coordinate goes back into the log-probability through the constant shift
            normal_prior = (pm * t_mult + t_add, ps * np.abs(t_mult))
            if logp_map is None:
                logp_affine = (logp_affine[0], logp_affine[1] + float(np.sum(np.log(np.abs(t_mult[np.isfinite(pm)])))))
            elif np.any(t_mult[np.isfinite(pm)] != 1.0):
                prior_host, normal_prior = prior_fn, None        # cannot fold the Jacobian behind a non-affine map: host prior
        to_theta = lambda c: (np.asarray(c) - t_add) / t_mult  # noqa: E731

        if like_host is not None and prior_fn is not None and prior_data is not None:
            prior_host, normal_prior = prior_fn, None            # the fused normal prior lives in the device likelihood path

        def _rows(fn):
            """Host callable on sampler coordinates [n,d] -> [n]: one call per point with a (1, d) argument, exactly what
            lnprob hands to like_fn / prior_fn (core.py:2097-2098)."""
            def call(q):
                th = to_theta(q)
                return np.array([float(np.asarray(fn(row.reshape(1, -1))).reshape(-1)[0]) for row in th], dtype=np.float64)
            return call

        def _batched(fn):                                        # the surrogate takes the whole batch in one GPU predict
            return lambda q: np.asarray(fn(to_theta(q)), dtype=np.float64).reshape(-1)
        sampler_extra = {}
        if prior_host is not None or like_host is not None:
            like_call = None
            if like_host is not None:
                like_call = _batched(like_host) if like_host == self.surrogate_log_likelihood else _rows(like_host)
            sampler_extra = dict(prior_fn=_rows(prior_host) if prior_host is not None else None, like_fn=like_call,
                                 gate_box=prior_host is None)
        self.prior_fn = partial(ut.lnprior_uniform, bounds=self.bounds) if prior_fn is None else prior_fn
        self.prior_fn_comment = ("Default uniform prior. \nPrior function: ut.prior_fn_uniform\n"
                                 f"\twith bounds {self.bounds}") if prior_fn_comment is None else prior_fn_comment
        self.nwalkers = int(10 * self.ndim) if nwalkers is None else int(nwalkers)
        self.nsteps = int(nsteps)
        if hasattr(self, "gp") and len(self.training_results["iteration"]) > 0:
            self.eval_gp_at_iteration(-1)   # makes self.gp carry the latest hyper-parameters / data
        if shard and (prior_host is not None or like_host is not None):
            raise ValueError('sampler_kwargs={"shard": True} needs the fused log-probability (surrogate likelihood, shipped priors '
                             'and scalers); host callables run as replicas')
        if opt_init:
            p0 = self.find_map(prior_fn=self.prior_fn)           # core.py:2290-2292
        else:
            p0_seed = self._seed()                               # (drawn on every rank alike: the model's stream stays in step)
            p0 = ut.prior_sampler(nsample=self.nwalkers, bounds=box, sampler="uniform",
                                  random_state=p0_seed + (1000003 * rank if replicas else 0))
        p0 = p0 * t_mult + t_add                   # walkers live in scaled coordinates
        if hasattr(self, "gp"):
            gp_obj, y_obj = self.gp, self._y
        else:                                      # like_fn="true" before any GP exists: the handle needs an owner only
            gp_obj, y_obj = HipGP(self.ndim), np.zeros(1)
        if self.verbose:
            print(f"Running emcee-style ensemble on the GPU with {self.nwalkers} walkers for {self.nsteps} steps...")
        all_chains, all_times, accumulated, run_number = [], [], 0, 1
        kw = dict(sampler_kwargs)
        kw.setdefault("seed", self._seed())
        if replicas:
            kw["seed"] = int(kw["seed"]) + rank
        if shard:
            kw["shard"] = True
        while True:
            t0 = time.time()
            self.emcee_sampler = EnsembleSampler(self.nwalkers, self.ndim, gp_obj, y_obj, _box, logp_affine=logp_affine,
                                                 normal_prior=normal_prior, logp_map=logp_map, **sampler_extra, **kw)
            self.emcee_sampler.run_mcmc(p0, self.nsteps, **run_kwargs)
            all_times.append(time.time() - t0)
            cur_iburn, cur_ithin = mcmc_utils.estimate_burnin(self.emcee_sampler, verbose=self.verbose)
            cur_burn = burn if burn is not None else cur_iburn
            cur_thin = thin if thin is not None else cur_ithin
            cur = to_theta(self.emcee_sampler.get_chain(discard=cur_burn, thin=cur_thin, flat=True))
            if replicas:
                cur = adist.gather_replicas(cur)                 # every rank's kept samples, in rank order, on every rank
            all_chains.append(cur)
            accumulated += cur.shape[0]
            if self.verbose and min_ess > 0:
                print(f"Run {run_number} complete: {cur.shape[0]} samples (total {accumulated})")
            if accumulated >= min_ess:
                break
            run_number += 1
            if run_number > 10:
                print(f"WARNING: Reached maximum of 10 runs, stopping with {accumulated} samples")
                break
            p0 = self.emcee_sampler.get_last_sample().coords
            kw["seed"] = self._seed() + (rank if replicas else 0)
        self.emcee_samples = np.vstack(all_chains) if len(all_chains) > 1 else all_chains[0]
        self._emcee_full, self._emcee_full_src = None, (self.emcee_sampler, t_add, t_mult)     # emcee_samples_full: on first access
        self.iburn, self.ithin = cur_iburn, cur_ithin
        self.burn, self.thin = cur_burn, cur_thin
        self.emcee_runtime = sum(all_times)
        if self.like_fn_name == "true":
            self.emcee_samples_true = self.emcee_samples
        elif self.like_fn_name == "surrogate":
            self.emcee_samples_gp = self.emcee_samples
        self.acc_frac = np.mean(self.emcee_sampler.acceptance_fraction)
        self.autcorr_time = np.mean(self.emcee_sampler.get_autocorr_time(tol=0))
        if replicas:                                             # the same numbers on every rank: the mean over the ensembles
            both = adist.allreduce_array([self.acc_frac, self.autcorr_time], "sum") / world
            self.acc_frac, self.autcorr_time = float(both[0]), float(both[1])
        self.emcee_ranks = world if (replicas or shard) else 1
        self.emcee_mode = "replicas" if replicas else ("sharded" if shard else "single")
        if self.verbose:
            print(f"Total samples: {self.emcee_samples.shape[0]}")
            print("Mean acceptance fraction: {0:.3f}".format(self.acc_frac))
            print("Mean autocorrelation time: {0:.3f} steps".format(self.autcorr_time))
        self.emcee_run = True
        if (replicas or shard) and rank != 0:
            return                                               # files are rank 0's
        if self.cache:
            try:
                self.save()
            except Exception:  # noqa: BLE001
                pass
        if samples_file is not None:
            fname = f"{self.savedir}/{samples_file}"
        elif self.like_fn_name == "true":
            fname = f"{self.savedir}/emcee_samples_final_{self.like_fn_name}.npz"
        else:
            res = getattr(self, "training_results", {"iteration": []})
            it = res["iteration"][-1] if len(res["iteration"]) else 0
            fname = f"{self.savedir}/emcee_samples_final_{self.like_fn_name}_iter_{it}.npz"
        np.savez(fname, samples=self.emcee_samples)

    run_mcmc = run_emcee  # BASELINE.json's name for the same entry point
