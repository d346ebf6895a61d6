"""How many evaluations the polish of find_next_point makes (alabi_utility_polish against scipy's L-BFGS-B around the same evaluations), and
the time per polish, on the 2-D demo problem while it grows from 50 to 90 points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alabi_amd import SurrogateModel, utility as ut
from alabi_amd.benchmarks import rosenbrock_fn
sm = SurrogateModel(lnlike_fn=rosenbrock_fn, bounds=[(-5, 5), (-5, 5)], savedir="/tmp/alabi_pe", verbose=False, random_state=1, cache=False)
sm.init_samples(ntrain=50); sm.init_gp(hyperopt_method="ml", gp_nopt=1)
ne, tn, ts, un, us = [], [], [], [], []
rng = np.random.RandomState(0)
for it in range(40):
    sm.active_train(niter=1, algorithm="bape", gp_opt_freq=1000)
    x0 = rng.uniform(sm._bounds[:, 0], sm._bounds[:, 1])
    yb = float(np.max(sm._y))
    t0 = time.perf_counter(); xa, ua = ut.polish_point(sm.gp, sm._y, x0, sm._bounds, algorithm="bape", y_best=yb, maxiter=30, method="native"); tn.append(time.perf_counter() - t0)
    ne.append(sm.gp._last_polish_nevals)
    t0 = time.perf_counter(); xb, ub = ut.polish_point(sm.gp, sm._y, x0, sm._bounds, algorithm="bape", y_best=yb, maxiter=30, method="scipy"); ts.append(time.perf_counter() - t0)
    un.append(ua); us.append(ub)
un, us = np.array(un), np.array(us)
print(f"native: {np.mean(ne):.1f} evaluations per polish (max {max(ne)}), {np.mean(tn)*1e3:.2f} ms; scipy {np.mean(ts)*1e3:.2f} ms")
print(f"value reached: native better {(un < us - 1e-6 * (abs(us) + 1)).sum()}, equal {(abs(un - us) <= 1e-6 * (abs(us) + 1)).sum()}, scipy better {(us < un - 1e-6 * (abs(us) + 1)).sum()} of {len(un)}")
