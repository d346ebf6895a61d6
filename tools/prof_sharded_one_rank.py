"""The sharded ensemble's step loop (alabi_ens_run_sharded) rehearsed on ONE GPU: microseconds per half step for one rank
with the memcpy stand-in and with a real one-rank RCCL communicator (ncclAllGather per half step, captured in the chunk's
hipGraph), C3-sized (256 walkers) and C4-sized (1024 walkers at N=5000).  UNMEASURED on multi-GPU hardware."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import EnsembleSampler, HipGP
from alabi_amd.dist import ShardedRun
from alabi_amd.workloads import make_config

for name in ("C3", "C4"):
    cfg = make_config(name); h = cfg["hyper"]
    gp = HipGP(cfg["d"], h["mean"], h["log_white_noise"], h["log_amp"], h["log_M"]); gp.compute(cfg["X"])
    for rccl in ("0", "1"):
        for graph in ("1", "0"):
            os.environ["ALABI_DIST_FORCE_RCCL"] = rccl; os.environ["ALABI_ENS_SHARD_GRAPH"] = graph
            s = EnsembleSampler(cfg["W"], cfg["d"], gp, cfg["y"], cfg["bounds"], seed=3)
            run = ShardedRun(s)
            c0 = torch.as_tensor(cfg["p0"], device="cuda")
            steps = 2048 if name == "C3" else 1024
            run.run(c0, steps, store=False); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); run.run(c0, steps, store=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            st = run.stats()
            print(f"{name} W={cfg['W']} one rank, RCCL communicator={rccl} graph={graph}: {1e6 * best / (2 * steps):.2f} us per half step, "
                  f"{cfg['W'] * steps / best:.3e} samples/s (chunks replayed {st['graph_replays']}, eager {st['eager_chunks']})", flush=True)
            del run, s
