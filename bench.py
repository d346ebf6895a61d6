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


def cpu_baseline(cfg, budget_s=12.0):
    """Reference-shaped CPU path (BASELINE.md B-ref): emcee's red-blue stretch move calling lnprob once per
    walker per half step, each call = box prior + single-point mean-only GP predict with the factorisation
    cached (the reference's CachedSurrogateLikelihood route, alabi/core.py:53-122, :2073-2100).  1 core."""
    from oracle.gp_oracle import OracleGP, sqexp_kernel
    from oracle import stretch_oracle as so
    from oracle.utility_oracle import lnprior_uniform
    h = cfg["hyper"]
    gp = OracleGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]).compute(cfg["X"])
    alpha = gp._compute_alpha(cfg["y"])
    X, bounds = cfg["X"], cfg["bounds"]

    def lnprob(theta):
        lp = lnprior_uniform(theta, bounds)
        if not np.isfinite(lp):
            return -np.inf
        k = sqexp_kernel(theta.reshape(1, -1), X, h["log_amp"], h["log_M"])
        return float(k @ alpha + h["mean"]) + lp

    coords = cfg["p0"].copy()
    logp = np.array([lnprob(c) for c in coords])
    rs = np.random.RandomState(12345)
    t0 = time.perf_counter()
    nsteps = 0
    while time.perf_counter() - t0 < budget_s:
        coords, logp, _ = so.emcee_literal_step(coords, logp, lnprob, rs)
        nsteps += 1
    dt = time.perf_counter() - t0
    return {"value": cfg["W"] * nsteps / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{nsteps} stretch-move steps x {cfg['W']} walkers ({dt:.1f} s), one Python lnprob call per "
                      "walker per half step, single-point mean-only predict with cached factorisation"}


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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from alabi_amd import EnsembleSampler, HipGP
    from alabi_amd.workloads import make_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # ALABI_DIST_BACKEND=gloo is a TEST rig (several ranks on one GPU, which RCCL refuses); the driver's runs use nccl
    backend = os.environ.get("ALABI_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
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

    shard = args.mode == "shard" and world > 1
    if shard:
        from alabi_amd.dist import HipBackend, ShardedEnsemble
        Wtot = W * world
        rngp = np.random.RandomState(1000 + d)
        p0 = rngp.uniform(cfg["bounds"][:, 0] * 0.5, cfg["bounds"][:, 1] * 0.5, (Wtot, d))
        sampler = EnsembleSampler(Wtot, d, gp, cfg["y"], cfg["bounds"], seed=2026)
        ens = ShardedEnsemble(HipBackend(sampler))
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
        if world > 1:
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
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = samples_per_step * args.steps / dt

    # Secondary measurement on N > 1 GPUs (every rank takes part): ONE ensemble of 256*N walkers sharded over the ranks
    # with an RCCL all-gather of the updated half after every half step (alabi_amd/dist.py).  Reported next to the
    # replica number; never the headline value.
    shard_info = None
    if world > 1 and not shard and not args.no_extras:
        try:
            from alabi_amd.dist import HipBackend, ShardedEnsemble
            Wtot = W * world
            p0s = np.random.RandomState(5).uniform(cfg["bounds"][:, 0] * 0.5, cfg["bounds"][:, 1] * 0.5, (Wtot, d))
            s2 = EnsembleSampler(Wtot, d, gp, cfg["y"], cfg["bounds"], seed=99)
            ens2 = ShardedEnsemble(HipBackend(s2))
            c0 = torch.as_tensor(p0s, device="cuda")
            ens2.run(c0, 8, store=False)
            fence()
            t1 = time.perf_counter()
            nst = 128
            ens2.run(c0, nst, store=True)
            fence()
            dt2 = time.perf_counter() - t1
            shard_info = {"walkers": Wtot, "steps": nst, "samples_per_s": Wtot * nst / dt2,
                          "us_per_half_step": 1e6 * dt2 / (2 * nst), "collective": "all_gather_into_tensor per half step"}
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
            kernel_name = f"ens_stream_kernel<{d},1,1024>"
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
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_by_kernel.json")
        if os.path.exists(pmc_path) and args.config == "C3" and args.ensembles == 1 and not shard:
            try:
                pmc = json.load(open(pmc_path))["void alabi::" + kernel_name.split("<")[0] + ("<10, 1, 1024>" if path == "stream" else "<10>")]
                # FETCH_SIZE is doubled only where the reads are 16-B-per-lane streams (the per-launch X loads of
                # ens_half_kernel); the persistent kernel's fetches are 8-byte polls / row reads, counted as reported
                fetch_corr = 1.0 if path == "stream" else 2.0
                traffic = (fetch_corr * pmc["FETCH_SIZE"]["mean_KB"] + pmc["WRITE_SIZE"]["mean_KB"]) * 1024.0
                if path == "stream":   # the PMC pass profiled launches of `pmc_steps` steps: scale to this launch
                    traffic *= min(args.mcmc_steps, 1024) / float(pmc.get("steps_per_launch", min(args.mcmc_steps, 1024)))
            except Exception:  # noqa: BLE001
                traffic = None
        out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                           "kernel": kernel_name, "path": path, "us_per_launch_incl_boundary": us_per_launch,
                           "us_per_half_step": 1e3 * ev_ms / (args.steps * 2 * args.mcmc_steps),
                           "flops_per_launch": flops_per_launch,
                           "note": "fp64 vector peak == fp64 matrix peak on MI355X; latency-bound kernel, see DESIGN.md"}
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
            for label, M, var in (("predict_mean_pts_per_s_M1e6", 1_000_000, False), ("predict_mean_pts_per_s_M256", 256, False),
                                  ("predict_meanvar_pts_per_s_M65536", 65536, True)):
                Xs = lo + (hi - lo) * torch.rand((M, d), dtype=torch.float64, device="cuda", generator=gen)
                gp.predict_device(y_dev, Xs, return_var=var); torch.cuda.synchronize()
                reps = 3 if M > 1000 else 200
                t1 = time.perf_counter()
                for _ in range(reps):
                    gp.predict_device(y_dev, Xs, return_var=var)
                torch.cuda.synchronize()
                extras[label] = M * reps / (time.perf_counter() - t1)
            out["extras"] = extras
        if shard_info is not None:
            out["sharded_ensemble"] = shard_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
