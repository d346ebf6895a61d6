"""Simulation of the stretch-move dataflow of ens_stream_kernel at the headline size (256 walkers, 128 workgroups): what a
producer-affine assignment of proposals to workgroups could gain.  Every proposal reads two rows (its own walker's and its
partner's); a row produced in the half step just before costs a cross-CU publish -> detect hand-off L unless the proposal runs
on the workgroup that produced it; older rows are assumed visible.  C = everything else in a half step.
  static : list position b -> workgroup b (what the kernel does)                    -> reproduces the measured 1.97-1.99 us
  greedy : proposals claim the producer of a fresh input, the rest fill free groups -> 1.45-1.5 us (+33 %) IF old rows cost nothing
The second number was not reached on hardware (tools/experiments/ens_producer_affine_schedule.patch)."""
import numpy as np
rng = np.random.default_rng(1)
W=256; H=128; G=128
C=0.9; L=1.09        # compute per item (everything but the hand-off), cross-CU hand-off latency
def sim(T, policy):
    avail = np.zeros(W)            # time at which walker's current row was produced
    prod = -np.ones(W, int)        # WG that produced it
    prod_h = -np.ones(W, int)*5    # half-step index at which it was produced
    wg_free = np.zeros(G)          # time WG finished its previous item
    h = 0
    nlocal = 0; nitems = 0
    for t in range(T):
        perm = rng.permutation(W)
        halves = [perm[:H], perm[H:]]
        for sp in (0,1):
            S = halves[sp]; Cset = halves[1-sp]
            cw = Cset[rng.integers(0, H, H)]
            # candidates
            assign = -np.ones(H, int)
            taken = np.zeros(G, bool)
            if policy == 'static':
                assign = np.arange(H)
            else:
                cand1 = np.where(prod_h[S] == h-1, prod[S], -1)      # own row fresh
                cand2 = np.where(prod_h[cw] == h-1, prod[cw], -1)    # partner row fresh
                # prefer the later-available input
                first = np.where((cand2 >= 0) & ((cand1 < 0) | (avail[cw] >= avail[S])), cand2, cand1)
                second = np.where(first == cand2, cand1, cand2)
                for cands in (first, second):
                    for i in range(H):
                        if assign[i] < 0 and cands[i] >= 0 and not taken[cands[i]]:
                            assign[i] = cands[i]; taken[cands[i]] = True
                free = [g for g in range(G) if not taken[g]]
                k = 0
                for i in range(H):
                    if assign[i] < 0:
                        assign[i] = free[k]; k += 1
            new_avail = np.zeros(H)
            for i in range(H):
                X = assign[i]; w = S[i]; c = cw[i]
                t_own = avail[w] + (0.0 if prod[w] == X and prod_h[w] == h-1 else L)
                t_par = avail[c] + (0.0 if prod[c] == X and prod_h[c] == h-1 else L)
                # older rows (produced >= 2 half steps ago) have long been visible: the L is already elapsed unless times are tight
                start = max(t_own, t_par, wg_free[X])
                if (prod[w] == X and prod_h[w] == h-1) or (prod[c] == X and prod_h[c] == h-1): nlocal += 1
                nitems += 1
                new_avail[i] = start + C
                wg_free[X] = start + C
            for i in range(H):
                avail[S[i]] = new_avail[i]; prod[S[i]] = assign[i]; prod_h[S[i]] = h
            h += 1
    return avail.max() / (2*T), nlocal / nitems
for pol in ('static', 'greedy'):
    per, fl = sim(1500, pol)
    print(pol, 'us per half step', round(per,3), 'samples/s', round(128/per*1e6/1e7,2), 'e7', 'items with a local input', round(fl,3))
