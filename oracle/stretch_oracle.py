"""fp64 restatement of the emcee red-blue stretch move that alabi's run_emcee drives.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED at this
boundary: emcee (>= 3.0, /root/reference/setup.py:16) is an un-vendored third-party
dependency, not installed here; the reference has no tests.  Restated from emcee 3.x
``moves/red_blue.py`` (RedBlueMove.propose) and ``moves/stretch.py``
(StretchMove.get_proposal), anchored on the reference call site
alabi/core.py:2319-2325 (EnsembleSampler(nwalkers, ndim, self.lnprob).run_mcmc) and
alabi/core.py:2073-2100 (lnprob = like_fn(theta) + prior_fn(theta)).

Two statements of the same step are given:

* ``emcee_literal_step`` follows emcee's code order draw by draw, taking the random
  numbers from a ``numpy.random.RandomState`` exactly as emcee does (shuffle of the
  0/1 label vector, ``rand(Ns)``, ``randint(Nc, size=Ns)``, one ``rand()`` per
  accept test).  It can also record the draws.
* ``stretch_step_arrays`` consumes pre-drawn arrays keyed by WALKER id
  (order / u_z / partner / u_acc), which is the contract of the HIP half-step
  kernel (include/alabi_hip.h: alabi_ens_step_with_randoms).  Given the draws
  recorded from the literal step it must reproduce it bit for bit
  (tests/test_oracle_crosscheck.py::test_array_step_reproduces_emcee_literal_step_bit_for_bit), which is what "walker index arithmetic
  bit-exact given the same uniforms" means in this repo.

``draw_step_randoms`` is the counter-based generator (Philox4x32-10) the device
uses in production; the device kernel and this function agree bit for bit.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "philox4x32_10", "u53", "draw_step_randoms", "stretch_step_arrays",
    "emcee_literal_step", "literal_draws_to_arrays", "run_ensemble", "box_lnprior_batch",
]

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint64(0x9E3779B9)
_W1 = np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)

STREAM_SPLIT, STREAM_PROPOSE, STREAM_ACCEPT = 0, 1, 2


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al. 2011).  ctr: (...,4) uint32-valued, key: (2,)."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) & _MASK for i in range(4)]
    k0 = np.uint64(int(key[0]) & 0xFFFFFFFF)
    k1 = np.uint64(int(key[1]) & 0xFFFFFFFF)
    for r in range(10):
        if r > 0:
            k0 = (k0 + _W0) & _MASK
            k1 = (k1 + _W1) & _MASK
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
    return np.stack(c, axis=-1).astype(np.uint32)


def u53(a, b):
    """53-bit uniform in [0,1) from two 32-bit words (same recipe as numpy's random_sample)."""
    a = np.asarray(a, dtype=np.uint64) >> np.uint64(5)
    b = np.asarray(b, dtype=np.uint64) >> np.uint64(6)
    return (a.astype(np.float64) * 67108864.0 + b.astype(np.float64)) / 9007199254740992.0


def _ctr(step, walkers, stream):
    w = np.asarray(walkers, dtype=np.uint64)
    ctr = np.empty(w.shape + (4,), dtype=np.uint64)
    ctr[..., 0] = np.uint64(int(step) & 0xFFFFFFFF)
    ctr[..., 1] = np.uint64((int(step) >> 32) & 0xFFFFFFFF)
    ctr[..., 2] = w
    ctr[..., 3] = np.uint64(stream)
    return ctr


def draw_step_randoms(seed, step, W, id0=0):
    """Counter-based draws for one step; identical on every rank and on the device.

    ``id0`` is the global id of this ensemble's walker 0 (e * W for the e-th of several independent
    ensembles sharing a launch); the returned ids stay LOCAL (0..W-1).

    Returns (order[W] int32, n0, u_z[W], partner[W] int32, u_acc[W]).
    ``order`` lists the label-0 walkers in index order then the label-1 walkers in
    index order, the labels being (rank of a 64-bit Philox key) % 2, i.e. a uniformly
    random balanced 0/1 labelling as emcee's ``random.shuffle(arange(W) % 2)`` gives.
    u_z / partner / u_acc are indexed by WALKER id; ``partner`` indexes the
    complementary list of the walker's own set.
    """
    key = (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    ids = np.arange(W)
    gids = ids + int(id0)
    r = philox4x32_10(_ctr(step, gids, STREAM_SPLIT), key).astype(np.uint64)
    k64 = (r[:, 0] << _S32) | r[:, 1]
    rank = np.empty(W, dtype=np.int64)
    rank[np.lexsort((ids, k64))] = ids  # ties broken by walker id
    label = rank % 2
    order = np.concatenate([ids[label == 0], ids[label == 1]]).astype(np.int32)
    n0 = int(np.sum(label == 0))
    rp = philox4x32_10(_ctr(step, gids, STREAM_PROPOSE), key)
    u_z = u53(rp[:, 0], rp[:, 1])
    nc = np.where(label == 0, W - n0, n0).astype(np.uint64)
    partner = ((rp[:, 2].astype(np.uint64) * nc) >> _S32).astype(np.int32)
    ra = philox4x32_10(_ctr(step, gids, STREAM_ACCEPT), key)
    u_acc = u53(ra[:, 0], ra[:, 1])
    return order, n0, u_z, partner, u_acc


def box_lnprior_batch(theta, bounds):
    """Vectorised lnprior_uniform: 0 strictly inside the open box else -inf (utility.py:268-275)."""
    b = np.asarray(bounds, dtype=np.float64)
    inside = np.all((theta > b[:, 0]) & (theta < b[:, 1]), axis=1)
    return np.where(inside, 0.0, -np.inf)


def stretch_step_arrays(coords, logp, order, n0, u_z, partner, u_acc, lnprob_batch, a=2.0):
    """One full red-blue step from pre-drawn arrays keyed by walker id.

    Follows RedBlueMove.propose / StretchMove.get_proposal:
        zz = ((a-1) u + 1)^2 / a ; factors = (ndim-1) log zz ; q = c[r] - (c[r]-s) zz
        accept iff factors + logp(q) - logp(s) > log(u')
    """
    coords = np.array(coords, dtype=np.float64, copy=True)
    logp = np.array(logp, dtype=np.float64, copy=True)
    W, ndim = coords.shape
    accepted = np.zeros(W, dtype=bool)
    sets = [np.asarray(order[:n0]), np.asarray(order[n0:])]
    for split in range(2):
        S = sets[split]
        C = sets[1 - split]
        if len(S) == 0:
            continue
        s = coords[S]
        c = coords[C[partner[S]]]
        zz = ((a - 1.0) * u_z[S] + 1.0) ** 2.0 / a
        factors = (ndim - 1.0) * np.log(zz)
        q = c - (c - s) * zz[:, None]
        new_logp = np.asarray(lnprob_batch(q), dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            lnpdiff = factors + new_logp - logp[S]
            acc = lnpdiff > np.log(u_acc[S])
        coords[S[acc]] = q[acc]
        logp[S[acc]] = new_logp[acc]
        accepted[S[acc]] = True
    return coords, logp, accepted


def emcee_literal_step(coords, logp, lnprob_one, random, a=2.0, record=None, map_fn=map):
    """emcee 3.x RedBlueMove.propose with StretchMove.get_proposal, draw for draw.

    ``lnprob_one(theta[d]) -> float`` is called once per walker, as emcee's
    ``map(log_prob_fn, q)`` does (this is the call structure of the reference's CPU
    path, alabi/core.py:2319-2325 -> :2073).  ``record`` (dict) receives the draws.
    """
    coords = np.array(coords, dtype=np.float64, copy=True)
    logp = np.array(logp, dtype=np.float64, copy=True)
    nwalkers, ndim = coords.shape
    accepted = np.zeros(nwalkers, dtype=bool)
    all_inds = np.arange(nwalkers)
    inds = all_inds % 2
    random.shuffle(inds)
    if record is not None:
        record.update(inds=inds.copy(), u_z=[], rint=[], u_acc=[])
    for split in range(2):
        S1 = inds == split
        sets = [coords[inds == j] for j in range(2)]
        s = sets[split]
        c = np.concatenate(sets[:split] + sets[split + 1:], axis=0)
        Ns, Nc = len(s), len(c)
        u = random.rand(Ns)
        zz = ((a - 1.0) * u + 1) ** 2.0 / a
        factors = (ndim - 1.0) * np.log(zz)
        rint = random.randint(Nc, size=(Ns,))
        q = c[rint] - (c[rint] - s) * zz[:, None]
        new_log_probs = np.array([float(v) for v in map_fn(lnprob_one, list(q))])   # emcee: pool.map when a pool is given
        uacc = np.empty(Ns)
        for i, (j, f, nlp) in enumerate(zip(all_inds[S1], factors, new_log_probs)):
            lnpdiff = f + nlp - logp[j]
            uacc[i] = random.rand()
            with np.errstate(divide="ignore"):
                if lnpdiff > np.log(uacc[i]):
                    accepted[j] = True
        upd = all_inds[S1][accepted[S1]]
        m = accepted[S1]
        coords[upd] = q[m]
        logp[upd] = new_log_probs[m]
        if record is not None:
            record["u_z"].append(u); record["rint"].append(rint); record["u_acc"].append(uacc)
    return coords, logp, accepted


def literal_draws_to_arrays(record):
    """Re-key the draws recorded by ``emcee_literal_step`` by walker id."""
    inds = record["inds"]
    W = len(inds)
    ids = np.arange(W)
    order = np.concatenate([ids[inds == 0], ids[inds == 1]]).astype(np.int32)
    n0 = int(np.sum(inds == 0))
    u_z = np.empty(W); u_acc = np.empty(W); partner = np.empty(W, dtype=np.int32)
    for split in range(2):
        S = ids[inds == split]
        u_z[S] = record["u_z"][split]
        u_acc[S] = record["u_acc"][split]
        partner[S] = record["rint"][split]
    return order, n0, u_z, partner, u_acc


def run_ensemble(p0, nsteps, lnprob_batch, seed, a=2.0, thin_by=1, step0=0, logp0=None, id0=0):
    """Array-driven ensemble run with the counter-based draws (device production contract).

    Returns chain[nsteps//thin_by, W, d], chain_logp[.., W], n_accept[W], coords, logp.
    """
    coords = np.array(p0, dtype=np.float64, copy=True)
    W, d = coords.shape
    logp = np.asarray(lnprob_batch(coords), dtype=np.float64) if logp0 is None else np.array(logp0, dtype=np.float64)
    nstore = nsteps // thin_by
    chain = np.empty((nstore, W, d))
    chain_lp = np.empty((nstore, W))
    nacc = np.zeros(W, dtype=np.int64)
    for t in range(nsteps):
        order, n0, u_z, partner, u_acc = draw_step_randoms(seed, step0 + t, W, id0)
        coords, logp, acc = stretch_step_arrays(coords, logp, order, n0, u_z, partner, u_acc, lnprob_batch, a)
        nacc += acc
        if (t + 1) % thin_by == 0:
            chain[(t + 1) // thin_by - 1] = coords
            chain_lp[(t + 1) // thin_by - 1] = logp
    return chain, chain_lp, nacc, coords, logp
