// On-GPU affine-invariant ensemble sampler (red-blue stretch move) for gfx950.
//
// Replaces emcee.EnsembleSampler(W, d, sm.lnprob).run_mcmc(p0, nsteps) as driven by the
// reference at alabi/core.py:2319-2325, with the log-probability of alabi/core.py:2073-2100:
// surrogate GP mean (core.py:1486 / :85) + uniform box prior (utility.py:218-275).
// Semantics restated from emcee 3 moves/red_blue.py + moves/stretch.py (see
// oracle/stretch_oracle.py, which is the CPU statement these kernels are tested against):
//   per step: random balanced 0/1 labelling; for split in {0,1}: for every walker k of the
//   active set S: z = ((a-1)u+1)^2/a, partner r = randint(|C|), q = C[r] - (C[r]-S_k) z,
//   accept iff (d-1) ln z + lnp(q) - lnp(S_k) > ln u'.
//
// Kernels
//   ens_draw_kernel   one workgroup per step: Philox4x32-10 keys -> rank -> label -> the two
//                     ordered walker lists, plus (u_z, partner, u_acc) per walker.  Draws
//                     depend only on (seed, step, walker): every rank of a multi-GPU run and
//                     the CPU oracle generate identical values.
//   ens_half_kernel   one workgroup per proposal of a half step: the first d lanes build the
//                     proposal and the box test; then all 256 lanes evaluate the GP mean with
//                     coalesced SoA loads of the training set and a shuffle+LDS reduction;
//                     lane 0 does the accept test; the walker's state (and its chain row) is
//                     written in place.  Walkers of S write, walkers of C are only read, so a
//                     half step needs no intra-kernel synchronisation; the two half steps are
//                     ordered by the stream (a kernel boundary is cheaper than a grid barrier
//                     on this chip).
// The per-step work at the headline size (W=256, N=2000, d=10) is 2 x 128 workgroups x
// 2000 kernel evaluations: latency bound, not HBM bound (X and alpha, 176 KB, stay in L2).
#include "gp_device.hpp"

namespace alabi {

__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                     uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ inline double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// grid = steps of the chunk; dynamic LDS = W * (8 + 4) bytes.
__global__ void __launch_bounds__(256)
ens_draw_kernel(unsigned long long seed, const long long* __restrict__ run_state, int W,
                int* __restrict__ order, int* __restrict__ partner, double* __restrict__ u_z,
                double* __restrict__ u_acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    int* label = reinterpret_cast<int*>(smem + (size_t)W * 8);
    const long long step = run_state[0] + blockIdx.x;
    const uint32_t s_lo = (uint32_t)step, s_hi = (uint32_t)((unsigned long long)step >> 32);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const size_t base = (size_t)blockIdx.x * W;
    uint32_t r[4];
    for (int i = threadIdx.x; i < W; i += 256) {
        philox4x32_10(s_lo, s_hi, (uint32_t)i, 0u, k0, k1, r);
        keys[i] = ((uint64_t)r[0] << 32) | r[1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += 256) {
        const uint64_t ki = keys[i];
        int rank = 0;
        for (int j = 0; j < W; ++j) {
            const uint64_t kj = keys[j];
            rank += (kj < ki) || (kj == ki && j < i);
        }
        label[i] = rank & 1;
    }
    __syncthreads();
    const int n0 = (W + 1) / 2;
    for (int i = threadIdx.x; i < W; i += 256) {
        const int li = label[i];
        int pos = 0;
        for (int j = 0; j < i; ++j) pos += (label[j] == li);
        order[base + (li ? n0 : 0) + pos] = i;
        philox4x32_10(s_lo, s_hi, (uint32_t)i, 1u, k0, k1, r);
        u_z[base + i] = u53(r[0], r[1]);
        const uint64_t nc = li ? (uint64_t)n0 : (uint64_t)(W - n0);
        partner[base + i] = (int)(((uint64_t)r[2] * nc) >> 32);
        philox4x32_10(s_lo, s_hi, (uint32_t)i, 2u, k0, k1, r);
        u_acc[base + i] = u53(r[0], r[1]);
    }
}


template <int D>
__global__ void __launch_bounds__(256)
ens_half_kernel(HalfArgs p, DimVec inv_len, DimVec lo, DimVec hi) {
    __shared__ double q_s[ALABI_MAX_DIM], qs_s[ALABI_MAX_DIM], old_s[ALABI_MAX_DIM];
    __shared__ double scratch[4];
    __shared__ int acc_s;
    const int tid = threadIdx.x;
    const int* S = p.split == 0 ? p.order : p.order + p.n0;
    const int* C = p.split == 0 ? p.order + p.n0 : p.order;
    const int w = S[p.part_begin + blockIdx.x];
    const int nC = p.split == 0 ? p.W - p.n0 : p.n0;
    // caller-supplied lists (test entry) are range-checked: a bad index must not become a wild access
    if ((unsigned)w >= (unsigned)p.W) return;
    const int pr = p.partner[w];
    if ((unsigned)pr >= (unsigned)nC) return;
    const int cw = C[pr];
    if ((unsigned)cw >= (unsigned)p.W) return;
    // z = ((a-1) u + 1)^2 / a, evaluated in numpy's operation order (no contraction)
    const double t1 = (p.a - 1.0) * p.u_z[w] + 1.0;
    const double zz = (t1 * t1) / p.a;
    int ok = 1;
    if (tid < D) {
        double qv = 0.0, sv = 0.0;
        if (tid < p.d) {
            const double cv = p.coords[(size_t)cw * p.d + tid];
            sv = p.coords[(size_t)w * p.d + tid];
            qv = cv - (cv - sv) * zz;
            ok = (qv > lo.v[tid]) && (qv < hi.v[tid]);
            q_s[tid] = qv; old_s[tid] = sv;
        }
        qs_s[tid] = (tid < p.d) ? qv * inv_len.v[tid] : 0.0;
    }
    const int inb = __syncthreads_and(ok);
    double lp_new = -INFINITY;
    if (inb) {  // workgroup-uniform
        const double s = gp_kernel_dot_block<D>(p.Xt, p.alpha, p.Npad, qs_s, scratch);
        lp_new = fma(p.amp, s, p.mean);
    }
    const double lp_old = p.logp[w];
    if (tid == 0) {
        const double lnpdiff = ((double)p.d - 1.0) * log(zz) + lp_new - lp_old;
        acc_s = (lnpdiff > log(p.u_acc[w])) ? 1 : 0;
    }
    __syncthreads();
    const int acc = acc_s;
    if (acc) {
        if (tid < p.d) p.coords[(size_t)w * p.d + tid] = q_s[tid];
        if (tid == 0) {
            p.logp[w] = lp_new;
            if (p.n_accept) p.n_accept[w] += 1;
        }
    }
    if (p.chain || p.chain_logp) {
        const long long done = p.run_state[1] + p.local_t + 1;
        if (done % p.thin_by == 0) {
            const long long slot = done / p.thin_by - 1;
            if (p.chain && tid < p.d) p.chain[((size_t)slot * p.W + w) * p.d + tid] = acc ? q_s[tid] : old_s[tid];
            if (p.chain_logp && tid == 0) p.chain_logp[(size_t)slot * p.W + w] = acc ? lp_new : lp_old;
        }
    }
}

template <int D>
__global__ void __launch_bounds__(256)
ens_lnprob_kernel(const double* __restrict__ coords, int d, const double* __restrict__ Xt,
                  const double* __restrict__ alpha, int Npad, double amp, double mean, DimVec inv_len,
                  DimVec lo, DimVec hi, double* __restrict__ logp) {
    __shared__ double qs_s[ALABI_MAX_DIM];
    __shared__ double scratch[4];
    const int tid = threadIdx.x, w = blockIdx.x;
    int ok = 1;
    if (tid < D) {
        double qv = 0.0;
        if (tid < d) {
            qv = coords[(size_t)w * d + tid];
            ok = (qv > lo.v[tid]) && (qv < hi.v[tid]);
        }
        qs_s[tid] = qv * ((tid < d) ? inv_len.v[tid] : 0.0);
    }
    const int inb = __syncthreads_and(ok);
    double lp = -INFINITY;
    if (inb) lp = fma(amp, gp_kernel_dot_block<D>(Xt, alpha, Npad, qs_s, scratch), mean);
    if (tid == 0) logp[w] = lp;
}

__global__ void ens_advance_kernel(long long* run_state, long long n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { run_state[0] += n; run_state[1] += n; }
}

int launch_ens_draw(alabi_ens* e, int nsteps, hipStream_t s) {
    const size_t lds = (size_t)e->W * 12;
    hipLaunchKernelGGL(ens_draw_kernel, dim3(nsteps), dim3(256), lds, s, e->seed, e->run_state, e->W, e->order,
                       e->partner, e->u_z, e->u_acc);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_half_args(alabi_ens* e, const HalfArgs& args, int nblocks, hipStream_t s) {
    if (nblocks <= 0) return ALABI_OK;
    const int db = dim_bucket(e->d);
    ALABI_DISPATCH_DIM(db, hipLaunchKernelGGL(ens_half_kernel<D>, dim3(nblocks), dim3(256), 0, s, args,
                                              e->gp->inv_len, e->lo, e->hi));
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_lnprob(alabi_ens* e, const double* coords, int W, double* logp, hipStream_t s) {
    const int db = dim_bucket(e->d);
    alabi_gp* gp = e->gp;
    ALABI_DISPATCH_DIM(db, hipLaunchKernelGGL(ens_lnprob_kernel<D>, dim3(W), dim3(256), 0, s, coords, e->d, gp->Xt,
                                              gp->alpha, gp->Npad, exp(gp->log_amp), gp->mean, gp->inv_len,
                                              e->lo, e->hi, logp));
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_advance(alabi_ens* e, long long n, hipStream_t s) {
    hipLaunchKernelGGL(ens_advance_kernel, dim3(1), dim3(64), 0, s, e->run_state, n);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
