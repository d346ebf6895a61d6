#!/usr/bin/env python3
"""bench.py -- surrogate-MCMC samples/s (+ GP-predict points/s) at N_train=2000, d=10, 256 walkers per GPU.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  One bench "step" = one pass of the hot path over one batch:
`--mcmc-steps` stretch-move steps of the whole ensemble (W * mcmc_steps walker updates), every state stored
to the chain in HBM.  W untimed warm-up steps (this is where the hipGraph is captured), then exactly K steps
bracketed by barrier + torch.cuda.synchronize() on both sides; the time is the MAX over ranks; rank 0 prints
ONE JSON line.

Multi-GPU: `--mode replicas` (default) = one independent 256-walker ensemble per GPU, different seeds, no
data-path collective (weak scaling).  `--mode shard` = ONE ensemble of 256*N walkers, active half partitioned
over the ranks, one RCCL all-gather per half step (alabi_amd/dist.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = fp64 matrix peak (spec); MI355X_MICROARCH.md lists no fp64 row


_CPU = {}   # worker-side state of the CPU baseline (inherited by fork: no pickling of the GP)


def _cpu_setup(cfg):
    from oracle.gp_oracle import OracleGP
    h = cfg["hyper"]
    gp = OracleGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"])
    _CPU.update(alpha=gp._compute_alpha(cfg["y"]), X=cfg["X"], bounds=cfg["bounds"], h=h)


def _cpu_lnprob(theta):
    """alabi's lnprob (core.py:2073-2100): uniform box prior + single-point mean-only predict, factorisation cached."""
    from oracle.gp_oracle import sqexp_kernel
    from oracle.utility_oracle import lnprior_uniform
    lp = lnprior_uniform(theta, _CPU["bounds"])
    if not np.isfinite(lp):
        return -np.inf
    h = _CPU["h"]
    k = sqexp_kernel(theta.reshape(1, -1), _CPU["X"], h["log_amp"], h["log_M"])
    return float((k @ _CPU["alpha"])[0] + h["mean"]) + lp


def cpu_baseline_vectorized(cfg, budget_s=5.0, keep_chain=None):
    """A stronger CPU statement than the reference's own call shape: the whole half-ensemble's proposals evaluated in one
    NumPy call (what emcee's vectorize=True would allow).  Reported beside the reference-shaped baselines so the GPU/CPU
    ratio is not read off the per-walker Python-call overhead alone."""
    from oracle import stretch_oracle as so
    from oracle.gp_oracle import sqexp_kernel
    _cpu_setup(cfg)
    h, alpha, X, bounds = _CPU["h"], _CPU["alpha"], _CPU["X"], _CPU["bounds"]

    isd = np.exp(-0.5 * np.asarray(h["log_M"]))
    Xs = X * isd; x2 = np.sum(Xs * Xs, axis=1); amp = np.exp(h["log_amp"])

    def lnprob_batch(q):                      # r2 through one GEMM (BLAS, all threads), one exp pass, one GEMV
        lp = so.box_lnprior_batch(q, bounds)
        out = np.full(len(q), -np.inf)
        ok = np.isfinite(lp)
        if ok.any():
            qs = q[ok] * isd
            r2 = np.maximum(np.sum(qs * qs, axis=1)[:, None] + x2[None, :] - 2.0 * (qs @ Xs.T), 0.0)
            out[ok] = amp * (np.exp(-0.5 * r2) @ alpha) + h["mean"]
        return out

    W = cfg["W"]
    assert np.allclose(lnprob_batch(cfg["p0"][:8]), sqexp_kernel(cfg["p0"][:8], X, h["log_amp"], h["log_M"]) @ alpha + h["mean"],
                       rtol=1e-6, atol=1e-6)
    coords = cfg["p0"].copy(); logp = lnprob_batch(coords)
    rs = np.random.RandomState(7); ids = np.arange(W)
    t0 = time.perf_counter(); nsteps = 0
    while time.perf_counter() - t0 < budget_s:
        lab = ids % 2; rs.shuffle(lab)
        order = np.concatenate([ids[lab == 0], ids[lab == 1]]); n0 = W - int(lab.sum())
        partner = np.where(lab == 0, rs.randint(W - n0, size=W), rs.randint(max(n0, 1), size=W))
        coords, logp, _ = so.stretch_step_arrays(coords, logp, order, n0, rs.rand(W), partner, rs.rand(W), lnprob_batch)
        nsteps += 1
        if keep_chain is not None:
            keep_chain.append(coords.copy())
    dt = time.perf_counter() - t0
    return {"value": cfg["W"] * nsteps / dt, "unit": "samples/s", "cores": "numpy default threads", "kind": "port",
            "sample": f"{nsteps} stretch-move steps x {cfg['W']} walkers ({dt:.1f} s), half-ensemble proposals evaluated "
                      "in one NumPy call (not the reference's call shape)"}


def cpu_predict_baseline(cfg):
    """GP-predict points/s of the NumPy/SciPy oracle (BLAS threads = library default): mean-only on 10^4 points and
    mean+variance (cho_solve, as george does) on 2048 points of the same workload."""
    from oracle.gp_oracle import OracleGP
    h = cfg["hyper"]
    gp = OracleGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"])
    rs = np.random.RandomState(5)
    lo, hi = cfg["bounds"][:, 0], cfg["bounds"][:, 1]
    out = {}
    for label, M, var in (("mean_pts_per_s", 10000, False), ("meanvar_pts_per_s", 2048, True)):
        Xs = lo + (hi - lo) * rs.rand(M, cfg["d"])
        gp.predict(cfg["y"], Xs[:64], return_var=var)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.5:
            gp.predict(cfg["y"], Xs, return_var=var); n += 1
        out[label] = M * n / (time.perf_counter() - t0)
    out["sample"] = "oracle.OracleGP.predict: 10^4 points mean-only, 2048 points mean+variance; NumPy/SciPy default threads"
    return out


def cpu_baseline(cfg, budget_s=10.0, cores=1):
    """Reference-shaped CPU path (BASELINE.md B-ref): emcee's red-blue stretch move calling lnprob once per walker per
    half step (CachedSurrogateLikelihood route, alabi/core.py:53-122).  cores > 1 hands the per-walker calls to a
    process pool with pool.map, as run_emcee(multi_proc=True) does (core.py:2300, :2322).  Must run BEFORE this process
    touches the GPU (the pool forks)."""
    from oracle import stretch_oracle as so
    _cpu_setup(cfg)
    pool = None
    map_fn = map
    if cores > 1:
        import multiprocessing as mp
        pool = mp.get_context("fork").Pool(cores)
        map_fn = pool.map
    coords = cfg["p0"].copy()
    logp = np.array(list(map_fn(_cpu_lnprob, list(coords))))
    rs = np.random.RandomState(12345)
    t0 = time.perf_counter()
    nsteps = 0
    while time.perf_counter() - t0 < budget_s:
        coords, logp, _ = so.emcee_literal_step(coords, logp, _cpu_lnprob, rs, map_fn=map_fn)
        nsteps += 1
    dt = time.perf_counter() - t0
    if pool is not None:
        pool.close(); pool.join()
    return {"value": cfg["W"] * nsteps / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{nsteps} stretch-move steps x {cfg['W']} walkers ({dt:.1f} s), one Python lnprob call per "
                      "walker per half step" + (" mapped over a process pool" if cores > 1 else "") +
                      ", single-point mean-only predict with cached factorisation"}


def pmc_file():
    """The committed PMC passes of this bench command (tools/collect_profiles.sh): the newest round's file."""
    for name in ("r04_pmc_by_kernel.json", "r03_pmc_by_kernel.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            return path
    return None


def cpu_end_to_end_baseline(cfg, budget_s=12.0):
    """Reference-shaped CPU legs of the end-to-end calls (bounded samples, extrapolated where stated), on the oracle:
    * one active-learning iteration = find_next_point as the reference runs it -- `nopt` = 5 scipy L-BFGS-B starts on the BAPE
      utility, one single-point predict(return_var=True) per objective call (alabi/core.py:1587-1667, utility.py:1030-1163;
      gradients by scipy's finite differences, i.e. use_grad_opt=False: the reference's own gradient route inverts K per call)
      -- plus the from-scratch refit of the N + 1 points (core.py:1780 -> :1158);
    * one fold fit of the k-fold CV search (gp_utils.py:568-600: compute on 0.8 N rows, log-likelihood, predict the held-out
      rows), extrapolated to the 875 fits of init_gp(hyperopt_method="cv") with the default 100 + 50 + 25 candidates x 5 folds."""
    from scipy.optimize import minimize
    from oracle.gp_oracle import OracleGP
    from oracle.utility_oracle import bape_utility
    h, d, X, y, b = cfg["hyper"], cfg["d"], cfg["X"], cfg["y"], cfg["bounds"]
    out = {"kind": "port", "cores": "numpy / scipy default threads"}
    best = None
    for _ in range(2):                                                  # (the first call also warms the BLAS threads up)
        t0 = time.perf_counter()
        gp = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X)
        gp._compute_alpha(y)
        dt0 = 1e3 * (time.perf_counter() - t0)
        best = dt0 if best is None else min(best, dt0)
    out["refit_from_scratch_ms"] = best
    predict = lambda t: gp.predict(y, np.atleast_2d(t), return_var=True)  # noqa: E731
    obj = lambda t: float(bape_utility(t, predict, b))  # noqa: E731
    rs = np.random.RandomState(3)
    t0 = time.perf_counter(); nstart = 0; ncall = [0]

    def counted(t):
        ncall[0] += 1
        return obj(t)
    while nstart < 5 and time.perf_counter() - t0 < budget_s:
        x0 = b[:, 0] + (b[:, 1] - b[:, 0]) * rs.rand(d)
        minimize(counted, x0, method="L-BFGS-B", bounds=[tuple(r) for r in b], options={"maxiter": 15})
        nstart += 1
    dt = time.perf_counter() - t0
    out["find_next_point_ms"] = 1e3 * dt * 5.0 / max(nstart, 1)
    out["active_train_iter_ms"] = out["find_next_point_ms"] + out["refit_from_scratch_ms"]
    out["sample_active"] = (f"{nstart} of 5 L-BFGS-B starts (maxiter 15, {ncall[0]} single-point predict(return_var) calls, "
                            f"{dt:.1f} s) scaled to 5, + one from-scratch refit at N = {len(X)}")
    n = len(X)
    perm = np.random.RandomState(4).permutation(n)
    val, train = np.sort(perm[: n // 5]), np.sort(perm[n // 5:])
    t0 = time.perf_counter(); nfit = 0
    while nfit < 3 or (time.perf_counter() - t0 < 3.0 and nfit < 40):
        g = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(X[train])
        g.log_likelihood(y[train]); g.predict(y[train], X[val])
        nfit += 1
    per_fit = (time.perf_counter() - t0) / nfit
    out["cv_fold_fit_ms"] = 1e3 * per_fit
    out["init_gp_cv_s_extrapolated"] = 875 * per_fit
    out["sample_cv"] = f"{nfit} fold fits (compute on {len(train)} rows + log-likelihood + predict {len(val)} rows), x 875 / {nfit}"
    return out


def end_to_end_extras(cfg):
    """What a user of SurrogateModel waits for besides the sampler (reference: 6.29 it/s active_train in
    docs/source/gp_tutorial.ipynb:202, on a smaller problem): init_gp with the default k-fold CV search (100 + 50 + 25 candidates
    x 5 folds = 875 fits of 0.8 N rows) and with the ML fit, one re-optimisation of the hyper-parameters as active_train runs it
    every gp_opt_freq iterations, and an active-learning iteration (scan + zoom + polish, true-function call, append) at C3 and
    with 10^6 candidates at the C5 size.  Wall-clock with a device synchronisation on both sides."""
    import tempfile
    import torch
    from alabi_amd import SurrogateModel
    from alabi_amd.workloads import make_config
    res = {}
    tmp = tempfile.mkdtemp(prefix="alabi_bench_")

    def model(c, seed=0):
        f = os.path.join(tmp, f"{c['name']}_train.npz")
        if not os.path.exists(f):
            np.savez(f, theta=c["X"], y=c["y"].reshape(-1, 1))
        sm = SurrogateModel(lnlike_fn=c["fn"], bounds=c["bounds"], savedir=tmp, verbose=False, random_state=seed, cache=False)
        sm.init_samples(train_file=f)
        return sm

    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r
    try:
        for method, key in (("cv", "init_gp_cv_s"), ("ml", "init_gp_ml_s")):
            times = []
            for rep in range(3):
                sm = model(cfg, seed=rep)
                times.append(timed(lambda: sm.init_gp(hyperopt_method=method))[0])
            res[key] = min(times[1:])
            res[key + "_first_call"] = times[0]
        res["init_gp_cv_fits"] = 875
        res["init_gp_cv_tflops"] = 875 * (0.8 * cfg["N"]) ** 3 / 3.0 / res["init_gp_cv_s"] / 1e12
        # sm now holds the ML-fitted GP: one hyper-parameter re-optimisation each way, then plain iterations
        sm.opt_gp_kwargs["hyperopt_method"] = "cv"
        res["opt_gp_cv_ms"] = 1e3 * min(timed(lambda: sm._opt_gp(**sm.opt_gp_kwargs))[0] for _ in range(2))
        sm.opt_gp_kwargs["hyperopt_method"] = "ml"
        res["opt_gp_ml_ms"] = 1e3 * min(timed(lambda: sm._opt_gp(**sm.opt_gp_kwargs))[0] for _ in range(2))
        sm.active_train(niter=2, gp_opt_freq=10 ** 6)                                   # warm-up (L^-1 cache, first append)
        niter = 12
        dt, _ = timed(lambda: sm.active_train(niter=niter, gp_opt_freq=10 ** 6))
        res["active_train_iter_ms"] = 1e3 * dt / niter
        res["active_train_iter_detail"] = "C3 + 14 points: scan 16384 candidates + 4 zoom stages + polish, lnlike call, append; no hyper-fit"
        res["active_train_appended"] = int(getattr(sm.gp, "appended", 0))
        # the reference's default sampling call (nsteps = 5e4, core.py:2108) with 256 walkers: sampling + burn-in / thinning estimate
        # (integrated autocorrelation time of the 1 GB chain on the device) + flattening + the samples file
        times = []
        for rep in range(2):
            times.append(timed(lambda: sm.run_emcee(nwalkers=256, nsteps=50_000, min_ess=0))[0])
        res["run_emcee_5e4_steps_s"] = min(times)
        res["run_emcee_5e4_steps_s_first_call"] = times[0]
        res["run_emcee_sampling_s"] = float(sm.emcee_sampler.last_run_seconds)
        res["run_emcee_kept_samples"] = int(sm.emcee_samples.shape[0])
        del sm
        torch.cuda.empty_cache()
    except Exception as ex:  # noqa: BLE001
        res["error_C3"] = repr(ex)[:300]
    try:
        c5 = make_config("C5")
        sm = model(c5, seed=5)
        sm.init_gp(hyperopt_method="ml", gp_nopt=1, optimizer_kwargs={"maxiter": 1})
        h = c5["hyper"]
        sm.gp.set_parameter_vector(np.concatenate([[h["mean"], h["log_white_noise"], h["log_amp"]], h["log_M"]]))
        sm.gp.compute(sm._theta, quiet=True)
        kw = {"ncand": 1_000_000, "refine": 1, "nrefine": 4096, "polish": 5}
        sm.active_train(niter=1, gp_opt_freq=10 ** 6, optimizer_kwargs=kw)              # warm-up: builds L^-1 (N = 10000)
        dt, _ = timed(lambda: sm.active_train(niter=2, gp_opt_freq=10 ** 6, optimizer_kwargs=kw))
        res["active_train_iter_ms_C5_1e6_candidates"] = 1e3 * dt / 2
        del sm
        torch.cuda.empty_cache()
    except Exception as ex:  # noqa: BLE001
        res["error_C5"] = repr(ex)[:300]
    return res


def pmc_summary(prefix, which="max"):
    """Counters of the kernel whose name starts with `prefix` from the committed PMC passes of this round
    (profiles/rNN_pmc_by_kernel.json: rocprofv3 --pmc runs of this same bench command, tools/collect_profiles.sh).
    FETCH_SIZE is doubled (16-byte-per-lane streaming reads are tallied at half their bytes on gfx950,
    MI355X_MICROARCH.md section HBM) unless the kernel's fetches are 8-byte polls.  None when the file is absent."""
    path = pmc_file()
    if path is None:
        return None
    try:
        allk = json.load(open(path))
        v = next(v for k, v in allk.items() if k.replace("void ", "").startswith("alabi::" + prefix))
        out = {"source": os.path.relpath(path, ROOT), "statistic": which + " over the launches of the profiled run"}
        for c, key in (("FETCH_SIZE", "fetch_KB"), ("WRITE_SIZE", "write_KB"), ("SQ_INSTS_VALU_MFMA_F64", "mfma_f64_instructions"),
                       ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles_summed_over_simds"), ("SQ_VALU_MFMA_COEXEC_CYCLES", "valu_mfma_coexec_cycles"),
                       ("SQ_INSTS_VALU", "valu_instructions"), ("SQ_WAVE_CYCLES", "wave_quad_cycles"), ("SQ_WAIT_ANY", "wait_any_quad_cycles"),
                       ("SQ_WAIT_INST_ANY", "wait_inst_quad_cycles"), ("SQ_ACTIVE_INST_VALU", "active_valu_quad_cycles")):
            if c in v:
                out[key] = v[c][which]
        return out
    except Exception:  # noqa: BLE001
        return None


def parity_gate(cfg, gp, y_dev, cpu_chain, args):
    """SURVEY.md section 8(d)'s correctness gate, outside the timed region: GPU predict against the CPU oracle on 256
    points, and the two-sample KS distance per marginal between a fresh GPU chain and the CPU (vectorised oracle) chain of
    this same run.  The CPU chain is short (12 s of host time), so its effective sample size -- not the GPU -- sets the KS
    noise floor: tests/test_gpu_configs.py::test_C3_ks_distance_gpu_vs_cpu_chain holds the < 0.01 assertion with a long one."""
    import torch
    from scipy.stats import ks_2samp
    from alabi_amd import EnsembleSampler
    from alabi_amd.mcmc_utils import integrated_time
    from oracle.gp_oracle import OracleGP
    h, d, W = cfg["hyper"], cfg["d"], cfg["W"]
    o = OracleGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"])
    b = cfg["bounds"]
    Xs = np.random.RandomState(11).uniform(b[:, 0], b[:, 1], (256, d))
    mu, var = gp.predict_device(y_dev, torch.as_tensor(Xs, device="cuda"), return_var=True)
    mu_o, var_o = o.predict(cfg["y"], Xs, return_var=True)
    gate = {"points": 256,
            "max_dmu_rel": float(np.max(np.abs(mu.cpu().numpy() - mu_o) / (np.abs(mu_o) + 1.0))), "tol_dmu_rel": 1e-8,
            "max_dvar_over_amp": float(np.max(np.abs(var.cpu().numpy() - var_o)) / np.exp(h["log_amp"])), "tol_dvar_over_amp": 1e-6}
    gate["predict_ok"] = bool(gate["max_dmu_rel"] <= 1e-8 and gate["max_dvar_over_amp"] <= 1e-6)
    if cpu_chain:
        burn = min(1500, len(cpu_chain) // 4)
        cpu = np.asarray(cpu_chain[burn:])                                   # [steps, W, d]
        s = EnsembleSampler(W, d, gp, cfg["y"], b, seed=4711)
        s.run_mcmc(cfg["p0"], 2000, store=False)
        s.run_mcmc(None, 100_000, thin_by=4)
        gpu = s.get_chain(flat=True)
        tau = float(np.max(integrated_time(torch.as_tensor(cpu, device="cuda"), tol=0))) if cpu.shape[0] > 50 else float("nan")
        ks = [float(ks_2samp(gpu[:, k], cpu[:, :, k].ravel()).statistic) for k in range(d)]
        n_eff = cpu.shape[0] * W / tau if tau == tau and tau > 0 else float("nan")
        gate.update({"ks_max": max(ks), "ks_per_marginal": ks, "ks_target": 0.01, "ks_gpu_samples": int(gpu.shape[0]),
                     "ks_cpu_samples": int(cpu.shape[0] * W), "ks_cpu_autocorr_steps": tau, "ks_cpu_n_eff": n_eff,
                     "ks_noise_floor_1sigma": float(0.6 / np.sqrt(n_eff)) if n_eff == n_eff else None,
                     "ks_ok": bool(max(ks) < 0.01)})
    return gate


def config_extras():
    """BASELINE.json's other configurations, each on its own workload, so that the driver's line (not only profiles/) holds
    their numbers: ensemble throughput of C1, C2, C4 and the C5-sized ensemble (N=10000, d=20, 2048 walkers) with the chain
    stored, and the C5 BAPE scan over 10^6 candidates.  A few hundred steps each: ~10 s in all."""
    import torch
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.utility import utility_scan
    from alabi_amd.workloads import make_config
    res = {}
    for name, steps in (("C1", 2048), ("C2", 2048), ("C4", 1024), ("C5", 512)):
        try:
            cfg = make_config(name)
            h = cfg["hyper"]
            gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gp.compute(cfg["X"]); torch.cuda.synchronize()
            fit_ms = 1e3 * (time.perf_counter() - t0)
            refit_ms = None                                    # K assembly + Cholesky once everything is allocated (best of 3)
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                gp.compute(cfg["X"]); torch.cuda.synchronize()
                dt_ms = 1e3 * (time.perf_counter() - t0)
                refit_ms = dt_ms if refit_ms is None else min(refit_ms, dt_ms)
            s = EnsembleSampler(cfg["W"], cfg["d"], gp, cfg["y"], cfg["bounds"], seed=5)
            s.run_mcmc(cfg["p0"], 32, store=False)
            torch.cuda.synchronize()
            best = None
            for _ in range(2):
                ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
                s._stream.wait_stream(torch.cuda.current_stream())
                ev0.record(s._stream)
                s.run_mcmc(None, steps, store=True)
                ev1.record(s._stream); torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1)
                best = ms if best is None else min(best, ms)
                s.reset()
            entry = {"N_train": cfg["N"], "d": cfg["d"], "walkers": cfg["W"], "steps": steps, "path": s.last_path,
                     "kernel": s.last_stream_kernel or "ens_half_kernel / ens_half_multi_kernel",
                     "samples_per_s": cfg["W"] * steps / (best * 1e-3), "us_per_half_step": 1e3 * best / (2 * steps),
                     "fit_ms_first_call": fit_ms, "cholesky_assemble_ms": refit_ms,
                     "cholesky_tflops": (cfg["N"] ** 3 / 3.0) / (refit_ms * 1e-3) / 1e12, "acceptance_fraction": None}
            # algorithmic fp64 work of the kernel sums: W/2 proposals x N points x (2d + 3) per half step
            entry["tflops"] = (cfg["W"] / 2.0) * cfg["N"] * (2 * cfg["d"] + 3) / (entry["us_per_half_step"] * 1e-6) / 1e12
            entry["roofline_frac_fp64"] = entry["tflops"] / FP64_PEAK_TFLOPS
            if s.last_stream_kernel == "ens_group_kernel":     # counters of the committed PMC passes (same bench command)
                entry["counters"] = pmc_summary("ens_group_kernel<%d," % ((cfg["d"] + 2 + 3) // 4))
            if name == "C5":
                gen = torch.Generator(device="cuda"); gen.manual_seed(6)
                lo = torch.as_tensor(cfg["bounds"][:, 0], device="cuda"); hi = torch.as_tensor(cfg["bounds"][:, 1], device="cuda")
                cand = lo + (hi - lo) * torch.rand((1_000_000, cfg["d"]), dtype=torch.float64, device="cuda", generator=gen)
                utility_scan(gp, cfg["y"], cand[:4096], cfg["bounds"], "bape")            # builds L^-1 once (as active_train does)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                utility_scan(gp, cfg["y"], cand, cfg["bounds"], "bape")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                entry["bape_scan_1e6_candidates_s"] = dt
                entry["bape_scan_candidates_per_s"] = 1e6 / dt
                entry["bape_scan_tflops"] = 1e6 * (cfg["N"] ** 2 + cfg["N"] * (2 * cfg["d"] + 3)) / dt / 1e12
                del cand
            res[name] = entry
            del s, gp
            torch.cuda.empty_cache()
        except Exception as ex:  # noqa: BLE001  (a side measurement must never cost the headline line)
            res[name] = {"error": repr(ex)[:300]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mcmc-steps", type=int, default=2048, help="stretch-move steps per bench step")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--walkers", type=int, default=None)
    ap.add_argument("--ntrain", type=int, default=None)
    ap.add_argument("--mode", choices=["replicas", "shard"], default="replicas")
    ap.add_argument("--ensembles", type=int, default=1,
                    help="independent ensembles of --walkers walkers sharing every launch on each GPU (default 1 = the "
                         "headline configuration: ONE 256-walker ensemble per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the predict / Cholesky side measurements")
    ap.add_argument("--shard-extra", action="store_true",
                    help="N > 1: also time ONE ensemble of 256*N walkers sharded over the ranks (all-gather per half step); "
                         "opt-in so that an untested collective path can never cost the headline line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks here, BEFORE anything in this process touches the GPU (no
        # torch import yet), as a child torch.distributed.run (one process per GPU, RCCL), and pass its exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.workloads import make_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # CPU baselines first: the multi-core one forks a pool, which must happen before the GPU is initialised
    cpu_base = cpu_base_all = cpu_base_vec = cpu_pred = cpu_e2e = None
    cpu_chain = None
    if world == 1 and not args.no_cpu_baseline:
        cfg0 = make_config(args.config, N=args.ntrain, W=args.walkers)
        cpu_base = cpu_baseline(cfg0, budget_s=10.0, cores=1)
        cpu_chain = []
        cpu_base_vec = cpu_baseline_vectorized(cfg0, budget_s=12.0, keep_chain=cpu_chain)
        cpu_pred = cpu_predict_baseline(cfg0)
        try:
            cpu_e2e = cpu_end_to_end_baseline(cfg0)
        except Exception as ex:  # noqa: BLE001
            cpu_e2e = {"error": repr(ex)[:200]}
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncores = max(1, min(ncores, 64))
        if ncores > 1:
            try:
                cpu_base_all = cpu_baseline(cfg0, budget_s=8.0, cores=ncores)
            except Exception as ex:  # noqa: BLE001
                cpu_base_all = {"error": repr(ex)[:200], "cores": ncores}
    # ALABI_DIST_BACKEND=gloo is a TEST rig (several ranks on one GPU, which RCCL refuses); the driver's runs use nccl
    backend = os.environ.get("ALABI_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    # ALABI_BENCH_FORCE_DIST=1 (under torch.distributed.run with ONE rank): initialise RCCL and take the collective code
    # paths (barrier, MAX all-reduce, the sharded run) although world == 1 -- the rehearsal a one-GPU box allows
    use_dist = world > 1 or (os.environ.get("ALABI_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    cfg = make_config(args.config, N=args.ntrain, W=args.walkers)
    h = cfg["hyper"]
    W, d, N = cfg["W"], cfg["d"], cfg["N"]
    gp = HipGP(d, h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"])
    t0 = time.perf_counter()
    gp.compute(cfg["X"])
    torch.cuda.synchronize()
    y_dev = torch.as_tensor(cfg["y"], device="cuda")
    gp.predict_device(y_dev, torch.as_tensor(cfg["X"][:1], device="cuda"))
    torch.cuda.synchronize()
    first_fit_s = time.perf_counter() - t0

    shard = args.mode == "shard" and use_dist
    if shard:
        from alabi_amd.dist import HipBackend, ShardedEnsemble, ShardedRun
        Wtot = W * world
        rngp = np.random.RandomState(1000 + d)
        p0 = rngp.uniform(cfg["bounds"][:, 0] * 0.5, cfg["bounds"][:, 1] * 0.5, (Wtot, d))
        sampler = EnsembleSampler(Wtot, d, gp, cfg["y"], cfg["bounds"], seed=2026)
        # the whole step loop in the library, one in-place ncclAllGather per half step on the stream; ALABI_BENCH_SHARD_IMPL=python
        # selects the Python loop around alabi_ens_half_step + torch.distributed (the implementation the gloo tests exercise)
        if os.environ.get("ALABI_BENCH_SHARD_IMPL") == "python":
            _py = ShardedEnsemble(HipBackend(sampler))

            class _Ens:
                def run(self, coords, nsteps, step0=0, store=True):
                    return _py.run(coords, nsteps, step0=step0)
            ens = _Ens()
        else:
            ens = ShardedRun(sampler)
        state = {"coords": torch.as_tensor(p0, device="cuda"), "step": 0}

        def one_step():
            _, c, _, _ = ens.run(state["coords"], args.mcmc_steps, step0=state["step"], store=True)
            state["coords"] = c
            state["step"] += args.mcmc_steps
        samples_per_step = Wtot * args.mcmc_steps
        launches_per_step = 2 * args.mcmc_steps
    else:
        E = args.ensembles
        sampler = EnsembleSampler(W, d, gp, cfg["y"], cfg["bounds"], seed=2026 + rank, n_ensembles=E)
        p0 = cfg["p0"] if E == 1 else np.random.RandomState(77).uniform(
            cfg["bounds"][:, 0] * 0.5, cfg["bounds"][:, 1] * 0.5, (W * E, d))
        sampler.run_mcmc(p0, 1, store=False)

        def one_step():
            sampler.run_mcmc(None, args.mcmc_steps, store=True)
            sampler._chains.clear(); sampler._chain_lps.clear(); sampler._thins.clear()
        samples_per_step = W * E * args.mcmc_steps * world
        launches_per_step = 2 * args.mcmc_steps

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    stream = sampler._stream
    stream.wait_stream(torch.cuda.current_stream())
    ev0.record(stream)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    ev1.record(stream)
    fence()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = samples_per_step * args.steps / dt

    # Secondary measurement on N > 1 GPUs (every rank takes part): ONE ensemble of 256*N walkers sharded over the ranks
    # with an RCCL all-gather of the updated half after every half step (alabi_amd/dist.py).  Reported next to the
    # replica number; never the headline value.  Opt-in (--shard-extra).
    shard_info = None
    if use_dist and not shard and args.shard_extra:
        try:
            from alabi_amd.dist import ShardedRun
            Wtot = W * world
            p0s = np.random.RandomState(5).uniform(cfg["bounds"][:, 0] * 0.5, cfg["bounds"][:, 1] * 0.5, (Wtot, d))
            s2 = EnsembleSampler(Wtot, d, gp, cfg["y"], cfg["bounds"], seed=99)
            ens2 = ShardedRun(s2)
            c0 = torch.as_tensor(p0s, device="cuda")
            ens2.run(c0, 8, store=False)
            fence()
            t1 = time.perf_counter()
            nst = 128
            ens2.run(c0, nst, store=True)
            fence()
            dt2 = time.perf_counter() - t1
            shard_info = {"walkers": Wtot, "steps": nst, "samples_per_s": Wtot * nst / dt2,
                          "us_per_half_step": 1e6 * dt2 / (2 * nst), "collective": "ncclAllGather per half step, enqueued by alabi_ens_run_sharded"}
        except Exception as ex:  # noqa: BLE001
            shard_info = {"error": repr(ex)[:300]}

    out = {
        "metric": "surrogate_mcmc_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{cfg['name']}: {cfg['description']}", "N_train": N, "d": d,
                   "walkers_per_gpu": W * (1 if shard else args.ensembles), "ensembles_per_gpu": 1 if shard else args.ensembles,
                   "mcmc_steps_per_bench_step": args.mcmc_steps,
                   "parallelism": ("sharded-ensemble+allgather" if shard else "replicas") if world > 1 else "single-gpu",
                   "chain_stored": True},
    }
    if rank == 0:
        # dominant kernel = ens_half_kernel (2 launches per stretch-move step).  Algorithmic flops per launch:
        # (W/2) proposals x N training points x (2d + 2 flops + 1 exp counted as 1 flop).
        n_prop = (W * world if shard else W * args.ensembles) / 2.0 / (world if shard else 1)
        path = getattr(sampler, "last_path", "launch-per-half-step")
        if path == "stream":
            # persistent dataflow kernel: ONE launch covers up to 1024 whole steps (the draw-buffer chunk)
            chunk = min(args.mcmc_steps, 1024)
            launches_per_step = -(-args.mcmc_steps // chunk)
            flops_per_launch = 2.0 * chunk * n_prop * N * (2 * d + 3)
            npad = -(-N // 64) * 64
            lanes = 256 if npad // 2 <= 1024 else 1024            # alabi_ens_create's choice (api.hip)
            lanes = int(os.environ.get("ALABI_ENS_THREADS", lanes))
            kernel_name = f"ens_stream_kernel<{d},{-(-(npad // 2) // lanes)},{lanes + 128}>"
        elif path == "group":
            chunk = min(args.mcmc_steps, 1024)
            launches_per_step = -(-args.mcmc_steps // chunk)
            flops_per_launch = 2.0 * chunk * n_prop * N * (2 * d + 3)
            kernel_name = f"ens_group_kernel<{(d + 2 + 3) // 4},...>"
        else:
            flops_per_launch = n_prop * N * (2 * d + 3)
            kernel_name = f"ens_half_kernel<{d}>"
        us_per_launch = 1e3 * ev_ms / (args.steps * launches_per_step)
        achieved = flops_per_launch / (us_per_launch * 1e-6) / 1e12
        # HBM/fabric bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE, then
        # --pmc WRITE_SIZE, same command; FETCH_SIZE doubled for 16-B-per-lane streaming reads per
        # MI355X_MICROARCH.md section HBM).  A PMC pass cannot run inside this process, so the number is read from
        # profiles/ and is null when the file is absent or the workload differs from the profiled one.
        traffic = None
        pmc_path = pmc_file()
        if pmc_path is not None and args.config == "C3" and args.ensembles == 1 and not shard:
            try:
                allk = json.load(open(pmc_path))
                pmc = next(v for k, v in allk.items() if k.startswith("void alabi::" + kernel_name.split("<")[0] + "<"))
                # FETCH_SIZE is doubled only where the reads are 16-B-per-lane streams (the per-launch X loads of
                # ens_half_kernel); the persistent kernel's fetches are 8-byte polls / row reads, counted as reported
                fetch_corr = 1.0 if path == "stream" else 2.0
                # the persistent kernel's full launches are the largest ones (1024 steps); shorter first/last launches
                # would dilute a mean
                key = "max_KB" if (path == "stream" and "max_KB" in pmc["FETCH_SIZE"]) else "mean_KB"
                traffic = (fetch_corr * pmc["FETCH_SIZE"][key] + pmc["WRITE_SIZE"][key]) * 1024.0
                if path == "stream":   # PMC launches cover `steps_per_launch` steps (1024 unless stated): scale to this launch
                    traffic *= min(args.mcmc_steps, 1024) / float(pmc.get("steps_per_launch", 1024))
            except Exception:  # noqa: BLE001
                traffic = None
        out["roofline"] = {"bound": "fp64-valu", "regime": "latency (dependent half-step chain, no MFMA in this kernel)",
                           "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                           "kernel": kernel_name, "path": path, "us_per_launch_incl_boundary": us_per_launch,
                           "us_per_half_step": 1e3 * ev_ms / (args.steps * 2 * args.mcmc_steps),
                           "flops_per_launch": flops_per_launch,
                           "note": "fp64 vector peak == fp64 matrix peak on MI355X; latency-bound kernel, see DESIGN.md"}
        if path == "stream":
            out["roofline"]["counters"] = pmc_summary("ens_stream_kernel<")
        out["acceptance_fraction"] = float(sampler.acceptance_fraction.mean()) if not shard else None
        if not args.no_extras:
            extras = {"first_fit_incl_init_s": first_fit_s}
            for rep in range(2):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                gp.compute(cfg["X"]); torch.cuda.synchronize()
                extras["cholesky_assemble_ms"] = 1e3 * (time.perf_counter() - t1)
            extras["cholesky_gflops"] = (N ** 3 / 3.0) / (extras["cholesky_assemble_ms"] * 1e-3) / 1e9
            gp.predict_device(y_dev, torch.as_tensor(cfg["X"][:1], device="cuda"))
            gen = torch.Generator(device="cuda"); gen.manual_seed(1)
            lo = torch.as_tensor(cfg["bounds"][:, 0], device="cuda"); hi = torch.as_tensor(cfg["bounds"][:, 1], device="cuda")
            for label, M, var in (("predict_mean_pts_per_s_M1e6", 1_000_000, False), ("predict_mean_pts_per_s_M1e4", 10_000, False),
                                  ("predict_mean_pts_per_s_M256", 256, False), ("predict_meanvar_pts_per_s_M1e6", 1_000_000, True),
                                  ("predict_meanvar_pts_per_s_M65536", 65536, True), ("predict_meanvar_pts_per_s_M1e4", 10_000, True),
                                  ("predict_meanvar_pts_per_s_M256", 256, True)):
                Xs = lo + (hi - lo) * torch.rand((M, d), dtype=torch.float64, device="cuda", generator=gen)
                gp.predict_device(y_dev, Xs, return_var=var); gp.predict_device(y_dev, Xs, return_var=var); torch.cuda.synchronize()
                reps = (2 if var else 10) if M > 100000 else (20 if M > 1000 else 200)
                # HIP events on the stream the kernels are launched on (torch's current stream): GPU time of `reps` calls
                ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(reps):
                    gp.predict_device(y_dev, Xs, return_var=var)
                ev1.record(); torch.cuda.synchronize()
                extras[label] = M * reps / (ev0.elapsed_time(ev1) * 1e-3)
            # secondary rooflines (HIP-event time over the calls, launch gaps included; kernel-only times are in profiles/).
            # predict_var: per query point N^2 flops for the triangular solve L^-1 k* (N^2/2 fma) + N(2d+3) for k*.
            pv_flops = 65536.0 * (N * N + N * (2 * d + 3))
            pv_tf = pv_flops * extras["predict_meanvar_pts_per_s_M65536"] / 65536.0 / 1e12
            extras["roofline_predict_var"] = {"bound": "mfma", "achieved": pv_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                              "frac": pv_tf / FP64_PEAK_TFLOPS, "kernel": "predict_kstar_tile_kernel + predict_var_w2_kernel", "sustained_mfma_peak_measured": {"one_wave_per_simd": 59.1, "two_waves_per_simd": 68.0},
                                              "counters": pmc_summary("predict_var_w2_kernel")}
            pm_tf = extras["predict_mean_pts_per_s_M1e6"] * N * (2 * d + 3) / 1e12
            extras["roofline_predict_mean"] = {"bound": "fp64-valu", "achieved": pm_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": pm_tf / FP64_PEAK_TFLOPS, "kernel": "predict_mean_mfma_kernel (M >= 2048, training points split over workgroups below ~2e5 queries; predict_mean_rowwise_kernel below)",
                                               "flops_per_point": N * (2 * d + 3), "counters": pmc_summary("predict_mean_mfma_kernel<")}
            ch_tf = extras["cholesky_gflops"] / 1e3
            extras["roofline_cholesky"] = {"bound": "mfma", "achieved": ch_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": ch_tf / FP64_PEAK_TFLOPS, "kernel": "chol_tasks8_kernel (one launch, tile tasks with slab-wise hand-over, panel solves on the matrix cores, eight waves per workgroup; N^3/3 flops, assembly included in the time)",
                                           "counters": pmc_summary("chol_tasks8_kernel", "min"),
                                           "note": "N=2000 is latency-bound on the chain of 31 diagonal factorisations; 13.2 us per chain step, DESIGN.md par. 4.1; the larger sizes are timed in extras.configs (C4: N=5000, C5: N=10000: 44.8 TFLOP/s; N=16000: 53.4 TFLOP/s; profiles/r04_cholesky_sizes.txt)"}
            if args.config == "C3" and not shard:
                t_cfg = time.perf_counter()
                extras["configs"] = config_extras()
                extras["configs_wall_s"] = time.perf_counter() - t_cfg
                if world == 1:
                    # (one rank only: under a process group SurrogateModel's init_gp / active_train are COLLECTIVE calls -- candidates
                    # dealt over ranks, sharded candidate scan -- and the other ranks of a multi-GPU bench run do not make them)
                    t_cfg = time.perf_counter()
                    extras["end_to_end"] = end_to_end_extras(cfg)
                    extras["end_to_end"]["wall_s"] = time.perf_counter() - t_cfg
                    if cpu_e2e is not None:
                        extras["end_to_end"]["cpu_baseline"] = cpu_e2e
            out["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            out["parity_gate"] = parity_gate(cfg, gp, y_dev, cpu_chain, args)
        if shard_info is not None:
            out["sharded_ensemble"] = shard_info
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
            out["speedup_vs_cpu_baseline"] = value / cpu_base["value"]
            if cpu_pred is not None:
                out["cpu_baseline_predict"] = cpu_pred
            if cpu_base_vec is not None:
                out["cpu_baseline_vectorized"] = cpu_base_vec
                out["speedup_vs_cpu_baseline_vectorized"] = value / cpu_base_vec["value"]
            if cpu_base_all is not None:
                out["cpu_baseline_all_cores"] = cpu_base_all
                if "value" in cpu_base_all:
                    out["speedup_vs_cpu_baseline_all_cores"] = value / cpu_base_all["value"]
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
