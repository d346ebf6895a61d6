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
// E independent ensembles of W walkers ("independent chains") can share every launch: walker
// ids are global (e*W + i), lists are per-ensemble segments, blockIdx.y is the ensemble.
//
// Kernels
//   ens_draw_kernel   one workgroup per (step, ensemble): Philox4x32-10 keys -> rank -> label ->
//                     the two ordered walker lists, then per list position one proposal record
//                     (walker id, partner walker id, z, (d-1) ln z, ln u').  Draws depend only on
//                     (seed, step, global walker id): every rank of a multi-GPU run and the CPU
//                     oracle generate identical values.
//   ens_prep_kernel   the same record construction from caller-supplied draws (test entry).
//   ens_half_kernel   one workgroup per proposal of a half step.  The record is ONE dependent
//                     load away from blockIdx (no list -> partner -> list chasing); the first d
//                     lanes build the proposal and the box test; all lanes evaluate the GP mean
//                     with coalesced SoA loads of the training set and a shuffle+LDS reduction;
//                     lane 0 does the accept test; the walker's state (and its chain row) is
//                     written in place.  Walkers of S write, walkers of C are only read, so a
//                     half step needs no intra-kernel synchronisation; the two half steps are
//                     ordered by the stream (a kernel boundary is cheaper than a grid barrier
//                     on this chip).
// The per-step work at the headline size (W=256, N=2000, d=10) is 2 x 128 workgroups x
// 2000 kernel evaluations: latency bound, not HBM bound (X and alpha, 176 KB, stay in L2).
#include <cstdlib>
#include <vector>
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

// Proposal record of list position `pos` (walker id wid = order[pos]) from raw draws keyed by
// walker id.  z = ((a-1) u + 1)^2 / a in NumPy's operation order (no contraction).
__device__ inline void write_record(const DrawBuffers& b, size_t pos, int wid, int cw, double u_z, double u_acc,
                                    double a, int d) {
    const double t1 = (a - 1.0) * u_z + 1.0;
    const double zz = (t1 * t1) / a;
    const double lnfac = ((double)d - 1.0) * log(zz), lnu = log(u_acc);
    b.order[pos] = wid;
    b.cw[pos] = cw;
    b.zz[pos] = zz;
    b.lnfac[pos] = lnfac;
    b.lnu[pos] = lnu;
    unsigned long long* pk = b.packed + 4 * pos;   // the same record in one 32-byte line for the persistent kernel
    pk[0] = (unsigned long long)(unsigned)wid | ((unsigned long long)(unsigned)cw << 32);
    pk[1] = (unsigned long long)__double_as_longlong(zz);
    pk[2] = (unsigned long long)__double_as_longlong(lnfac);
    pk[3] = (unsigned long long)__double_as_longlong(lnu);
}

// grid = (steps of the chunk, ensembles); dynamic LDS = Wp * 12 + 1024 bytes, Wp = W rounded up to a power of two.
// The rank of a walker's key among the step's keys (ties by walker id) decides its label: a bitonic sort of (key, id) pairs in
// LDS -- O(W log^2 W) compare-exchanges instead of the W^2 comparisons of the first version (0.21 ms per 1024-step chunk at
// 1024 walkers, 0.62 ms at 2048: 4 % of the C4 run) -- then the two ordered lists by a prefix count over the labels.
__global__ void __launch_bounds__(256)
ens_draw_kernel(unsigned long long seed, const long long* __restrict__ run_state, int W, int Wp, int d, double a,
                DrawBuffers b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                       // [Wp]; after the sort: label[W], olist[W]
    uint32_t* ids = reinterpret_cast<uint32_t*>(smem + (size_t)Wp * 8);       // [Wp]
    int* scan = reinterpret_cast<int*>(smem + (size_t)Wp * 12);               // [256]
    const int E = gridDim.y, e = blockIdx.y, tid = threadIdx.x;
    const long long step = run_state[0] + blockIdx.x;
    const uint32_t s_lo = (uint32_t)step, s_hi = (uint32_t)((unsigned long long)step >> 32);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const size_t base = ((size_t)blockIdx.x * E + e) * W;
    const uint32_t g0 = (uint32_t)e * (uint32_t)W;  // global id of this ensemble's walker 0
    uint32_t r[4];
    for (int i = tid; i < Wp; i += 256) {
        uint64_t key = ~0ull;                                                 // padding sorts behind every walker (ties by id)
        if (i < W) {
            philox4x32_10(s_lo, s_hi, g0 + (uint32_t)i, 0u, k0, k1, r);
            key = ((uint64_t)r[0] << 32) | r[1];
        }
        keys[i] = key; ids[i] = (uint32_t)i;
    }
    __syncthreads();
    for (int k = 2; k <= Wp; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < Wp; t += 256) {
                const int x = t ^ j;
                if (x > t) {
                    const uint64_t ka = keys[t], kb = keys[x];
                    const uint32_t ia = ids[t], ib = ids[x];
                    const bool gt = (ka > kb) || (ka == kb && ia > ib);
                    if (gt == ((t & k) == 0)) { keys[t] = kb; keys[x] = ka; ids[t] = ib; ids[x] = ia; }
                }
            }
            __syncthreads();
        }
    // rank of walker ids[t] is t: label = rank & 1 (`label` and `olist` reuse the keys' memory: the sort's last barrier is behind us)
    int* label = reinterpret_cast<int*>(smem);
    int* olist = label + W;   // local ids in list order (set 0 then set 1)
    for (int t = tid; t < Wp; t += 256) {
        const uint32_t i = ids[t];
        if (i < (uint32_t)W) label[i] = t & 1;
    }
    __syncthreads();
    // position of walker i inside its list = number of walkers j < i with the same label: every thread owns a run of
    // consecutive walkers, the runs' label-0 counts are scanned by one wave
    const int n0 = (W + 1) / 2;
    const int per = (W + 255) / 256, i0 = tid * per, i1 = (i0 + per < W) ? i0 + per : W;
    int zeros = 0;
    for (int i = i0; i < i1; ++i) zeros += (label[i] == 0);
    scan[tid] = zeros;
    __syncthreads();
    if (tid < 64) {
        int v[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = scan[4 * tid + q]; tot += v[q]; }
        int inc = tot;                                                        // inclusive scan over the 64 lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (tid >= off) inc += o; }
        int run = inc - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q) { scan[4 * tid + q] = run; run += v[q]; }
    }
    __syncthreads();
    {
        int z = scan[tid], o = i0 - z;                                        // label-0 / label-1 walkers before i0
        for (int i = i0; i < i1; ++i) {
            const int li = label[i];
            const int pos = li ? o++ : z++;
            olist[(li ? n0 : 0) + pos] = i;
        }
    }
    __syncthreads();
    for (int pos = tid; pos < W; pos += 256) {
        const int i = olist[pos];
        const int li = pos >= n0;
        philox4x32_10(s_lo, s_hi, g0 + (uint32_t)i, 1u, k0, k1, r);
        const double uz = u53(r[0], r[1]);
        const uint64_t nc = li ? (uint64_t)n0 : (uint64_t)(W - n0);
        const int pr = (int)(((uint64_t)r[2] * nc) >> 32);
        const int cw = olist[(li ? 0 : n0) + pr];
        philox4x32_10(s_lo, s_hi, g0 + (uint32_t)i, 2u, k0, k1, r);
        const double ua = u53(r[0], r[1]);
        b.partner[base + pos] = pr;
        b.u_z[base + pos] = uz;
        b.u_acc[base + pos] = ua;
        b.pos_of[base + i] = pos;
        write_record(b, base + pos, (int)g0 + i, (int)g0 + cw, uz, ua, a, d);
    }
}

// Records from caller-supplied draws (E = 1): order[W] lists set 0 then set 1, u_z / partner / u_acc are
// keyed by WALKER id, partner indexes the complementary list.  Bad indices yield an inert record (w = -1).
__global__ void __launch_bounds__(256)
ens_prep_kernel(const int* __restrict__ order, int n0, int W, const double* __restrict__ u_z,
                const int* __restrict__ partner, const double* __restrict__ u_acc, double a, int d,
                DrawBuffers b) {
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= W) return;
    const int w = order[pos];
    const int li = pos >= n0;
    const int nC = li ? n0 : W - n0;
    int cw = -1;
    double uz = 0.5, ua = 0.5;
    int wid = -1;
    if ((unsigned)w < (unsigned)W) {
        const int pr = partner[w];
        if ((unsigned)pr < (unsigned)nC) {
            const int c = order[(li ? 0 : n0) + pr];
            if ((unsigned)c < (unsigned)W) { wid = w; cw = c; uz = u_z[w]; ua = u_acc[w]; }
        }
    }
    write_record(b, pos, wid, cw, uz, ua, a, d);
}

typedef double f64x2 __attribute__((ext_vector_type(2)));

// Inverse of the y scaler applied to the GP mean (alabi/core.py:1483-1502 un-scales every prediction; the two non-affine
// scalers the reference ships are alabi/utility.py:62-71): 0 identity (affine scalers are folded into amp / mean),
// 1 nlog_scaler (y = -10^x), 2 log_scaler (y = 10^x).  Evaluated once per proposal by the deciding wave.
__device__ inline double apply_ymap(double x, int kind) {
    if (kind == 0) return x;
    const double v = pow(10.0, x);
    return kind == 1 ? -v : v;
}

// Normal-prior term of coordinate `lane` (< d) of a proposal, summed over the wave: lanes >= d contribute 0.
// consts rows 3 / 4: prior mean, 1 / std (0 where there is no normal prior).  Result valid in every lane.
// ---- squared exponential in the half-step kernels (round 3) -------------------------------------------------------------
// alpha_n exp(-|x_n - q|^2 / 2) = sgn(alpha_n) exp(q.x_n - h_n - |q|^2 / 2) with h_n = |x_n|^2 / 2 - ln|alpha_n| resident instead of
// alpha_n (sign in its lowest mantissa bit) and all coordinates relative to the mean of the training inputs (the products lose
// eps (|x|^2 + |q|^2) / 2 absolutely -- negligible near the centre): ONE fma per point and coordinate instead of a subtraction and
// an fma, no multiply by alpha -- 33 instead of 41 fp64 instructions per kernel evaluation in kernels whose time is their
// instruction count (ens_stream_kernel: 1.96 -> 1.89 us per half step at N = 2000, d = 10, without a register more: h takes
// alpha's place).  ens_stream_kernel, ens_half_kernel and ens_half_multi_kernel share these functions and the accumulation order
// (acc = 0; acc += term_a; acc += term_b per pair), so their chains stay bit-identical.  The other kernel families keep the
// difference form.
template <int D>
__device__ inline double se_neg_half_norm(const double (&q)[D]) {
    double n = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) n = fma(q[k], q[k], n);
    return -0.5 * n;
}
template <int D>
__device__ inline void se_pair_terms(const f64x2 (&x)[D], f64x2 h, const double (&q)[D], double nhq, double& fa, double& fb) {
    double da = nhq - h.x, db = nhq - h.y;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        da = fma(x[k].x, q[k], da);
        db = fma(x[k].y, q[k], db);
    }
    fa = exp_direct(da);
    fb = exp_direct(db);
    fa = __hiloint2double(__double2hiint(fa) ^ (__double2loint(h.x) << 31), __double2loint(fa));
    fb = __hiloint2double(__double2hiint(fb) ^ (__double2loint(h.y) << 31), __double2loint(fb));
}
#define ALABI_SE_PAD 1000.0   // h of a point that contributes nothing: exp(-1000 + ...) underflows to exactly 0

// h and the centred inputs of the squared-exponential half-step kernels (alabi_gp::Xc, ::ens_h)
__global__ void __launch_bounds__(256)
ens_se_prepare_kernel(const double* __restrict__ Xt, const double* __restrict__ centre, const double* __restrict__ alpha, int N, int Npad,
                      int d, int rows, double* __restrict__ Xc, double* __restrict__ h) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= Npad) return;
    // a point that contributes nothing (padding, alpha exactly 0 or NaN) gets h = ALABI_SE_PAD AND a zero row: with its real
    // coordinates the exponent q.x - |q|^2/2 - 1000 of se_pair_terms could still be large for a point tens of length scales from
    // the centre (round-3 advisor), with a zero row it is -|q|^2/2 - 1000 and the term underflows to exactly 0
    const double a = n < N ? alpha[n] : 0.0;
    const bool live = a != 0.0 && a == a;
    double hh = 0.0;
    for (int k = 0; k < rows; ++k) {
        const double v = (k < d && live) ? Xt[(size_t)k * Npad + n] - centre[k] : 0.0;
        Xc[(size_t)k * Npad + n] = v;
        hh = fma(v, v, hh);
    }
    double out = ALABI_SE_PAD;
    if (live) {
        long long bits = __double_as_longlong(0.5 * hh - log(fabs(a)));
        bits = (bits & ~1LL) | (a < 0.0 ? 1LL : 0LL);
        out = __longlong_as_double(bits);
    }
    h[n] = out;
}

__device__ inline double normal_prior_sum(const double* pmean, const double* pistd, int lane, int d, double x) {
    double t = 0.0;
    if (lane < d) { t = (x - pmean[lane]) * pistd[lane]; t = -0.5 * t * t; }
    return lane_bcast(wave_sum_dpp(t), 63);
}

template <int D, bool GENERIC>
__global__ void __launch_bounds__(1024)
ens_half_kernel(HalfArgs p) {
    __shared__ double q_s[ALABI_MAX_DIM], qs_s[ALABI_MAX_DIM], old_s[ALABI_MAX_DIM];
    __shared__ double scratch[16];
    const int tid = threadIdx.x, T = blockDim.x;
    const int e = blockIdx.y;
    const size_t pos = (size_t)e * p.W + (p.split ? p.n0 : 0) + p.part_begin + blockIdx.x;
    // (1) Issue this thread's first training-point loads right away (16 B per lane: points 2*tid and
    //     2*tid+1 of every coordinate row): they do not depend on the proposal, so their latency -- every
    //     kernel starts with a cold L2 on this chip, and a CU ingests only ~64 B/clk -- overlaps the
    //     record -> coords chain below.
    const int half = p.Npad >> 1;                 // Npad is a multiple of 64
    const bool vA = tid < half;
    const double* Xsrc = GENERIC ? p.Xt : p.Xc;        // squared exponential: centred inputs and h instead of alpha (se_pair_terms)
    const double* Asrc = GENERIC ? p.alpha : p.ens_h;
    f64x2 xa[D];
#pragma unroll
    for (int k = 0; k < D; ++k)
        xa[k] = vA ? reinterpret_cast<const f64x2*>(Xsrc + (size_t)k * p.Npad)[tid] : f64x2{0.0, 0.0};
    const f64x2 aa = vA ? reinterpret_cast<const f64x2*>(Asrc)[tid] : (GENERIC ? f64x2{0.0, 0.0} : f64x2{ALABI_SE_PAD, ALABI_SE_PAD});
    // (2) proposal record: one dependent load away from blockIdx
    const int w = p.rec.order[pos];
    if (w < 0) return;  // inert record (range-checked test input); workgroup-uniform
    const int cw = p.rec.cw[pos];
    const double zz = p.rec.zz[pos];
    // where the two rows live: in place (coords / logp), or -- sharded ensemble -- in the history of published rows
    const double* own_c = p.coords + (size_t)w * p.d;
    const double* own_lp = p.logp + w;
    const double* par_c = p.coords + (size_t)cw * p.d;
    if (p.shist) {
        const unsigned long long lw = p.rec.link[2 * pos];
        const int so = (int)(unsigned)(lw & 0xffffffffull), sp = (int)(unsigned)(lw >> 32);
        if (so >= 0) { own_c = p.shist + so; own_lp = own_c + p.d; }
        if (sp >= 0) par_c = p.shist + sp;
    }
    double lp_old = 0.0, lnfac = 0.0, lnu = 0.0;
    if (tid < 64) { lp_old = *own_lp; lnfac = p.rec.lnfac[pos]; lnu = p.rec.lnu[pos]; }  // wave 0 decides
    const double* inv_len = p.consts;
    const double* lo = p.consts + ALABI_MAX_DIM;
    const double* hi = p.consts + 2 * ALABI_MAX_DIM;
    int ok = 1;
    if (tid < D) {
        double qv = 0.0;
        if (tid < p.d) {
            const double cv = par_c[tid];
            const double sv = own_c[tid];
            qv = cv - (cv - sv) * zz;
            ok = (qv > lo[tid]) && (qv < hi[tid]);
            q_s[tid] = qv; old_s[tid] = sv;
            qv *= inv_len[tid];
            if (!GENERIC) qv -= p.centre[tid];
        }
        qs_s[tid] = qv;
    }
    const int inb = __syncthreads_and(ok);
    double lp_new = -INFINITY;
    if (inb) {  // workgroup-uniform
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = qs_s[k];
        double acc;
        if (!GENERIC) {
            const double nhq = se_neg_half_norm<D>(q);
            double fa, fb;
            se_pair_terms<D>(xa, aa, q, nhq, fa, fb);
            acc = 0.0; acc += fa; acc += fb;
            for (int j = tid + T; j < half; j += T) {
                f64x2 x[D];
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = reinterpret_cast<const f64x2*>(Xsrc + (size_t)k * p.Npad)[j];
                se_pair_terms<D>(x, reinterpret_cast<const f64x2*>(Asrc)[j], q, nhq, fa, fb);
                acc += fa; acc += fb;
            }
        } else {
            double r2a = 0.0, r2b = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double da = xa[k].x - q[k], db = xa[k].y - q[k];
                r2a = fma(da, da, r2a);
                r2b = fma(db, db, r2b);
            }
            acc = aa.x * radial<GENERIC>(r2a, p.kf);
            acc = fma(aa.y, radial<GENERIC>(r2b, p.kf), acc);
            for (int j = tid + T; j < half; j += T) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const f64x2 x = reinterpret_cast<const f64x2*>(p.Xt + (size_t)k * p.Npad)[j];
                    const double d0 = x.x - q[k], d1 = x.y - q[k];
                    s0 = fma(d0, d0, s0);
                    s1 = fma(d1, d1, s1);
                }
                const f64x2 al = reinterpret_cast<const f64x2*>(p.alpha)[j];
                acc = fma(al.x, radial<GENERIC>(s0, p.kf), acc);
                acc = fma(al.y, radial<GENERIC>(s1, p.kf), acc);
            }
        }
        // (3) wave totals by DPP, one LDS word per wave, ONE barrier; only wave 0 goes on
        const double wsum = wave_sum_dpp(acc);
        if ((tid & 63) == 63) scratch[tid >> 6] = wsum;
        __syncthreads();
        if (tid >= 64) return;
        const int nw = T >> 6;
        double part = (tid < nw) ? scratch[tid] : 0.0;
        part = wave_sum_dpp(part);   // fixed order: bit-reproducible
        const double s = lane_bcast(part, 63);
        lp_new = fma(p.amp, s, p.mean);
        if (p.ymap) lp_new = apply_ymap(lp_new, p.ymap);
        if (p.has_prior)
            lp_new += normal_prior_sum(p.consts + 3 * ALABI_MAX_DIM, p.consts + 4 * ALABI_MAX_DIM, tid, p.d, q_s[tid]) + p.prior_const;
    } else if (tid >= 64) {
        return;
    }
    // (4) wave 0: accept test in every lane (same inputs), first d lanes write the state
    const int acc_flag = (lnfac + lp_new - lp_old > lnu) ? 1 : 0;
    if (p.sout) {                                          // sharded ensemble: the new row goes to the history, nothing else is written
        double* o = p.sout + (size_t)blockIdx.x * (p.d + 2);
        if (tid < p.d) o[tid] = acc_flag ? q_s[tid] : old_s[tid];
        if (tid == 0) { o[p.d] = acc_flag ? lp_new : lp_old; o[p.d + 1] = acc_flag ? 1.0 : 0.0; }
        return;
    }
    if (acc_flag) {
        if (tid < p.d) p.coords[(size_t)w * p.d + tid] = q_s[tid];
        if (tid == 0) {
            p.logp[w] = lp_new;
            if (p.n_accept) p.n_accept[w] += 1;
        }
    }
    if (p.chain || p.chain_logp) {
        const long long done = p.run_state[1] + p.local_t + 1;
        if (done % p.thin_by == 0) {
            const size_t slot = (size_t)(done / p.thin_by - 1);
            const size_t WT = (size_t)p.W * gridDim.y;
            if (p.chain && tid < p.d)
                __builtin_nontemporal_store(acc_flag ? q_s[tid] : old_s[tid], &p.chain[(slot * WT + w) * p.d + tid]);
            if (p.chain_logp && tid == 0)
                __builtin_nontemporal_store(acc_flag ? lp_new : lp_old, &p.chain_logp[slot * WT + w]);
        }
    }
}

// NP proposals of the same half step per workgroup: the training set is streamed ONCE per workgroup and used for all of them.
// With more proposals than CUs (W/2 > 256, or E ensembles) ens_half_kernel is bound by L2 -> CU bandwidth: every workgroup
// pulls the whole X (440 KB at N = 5000, d = 10) for one proposal.  Per proposal the arithmetic, the lane -> point map and the
// reduction order are those of ens_half_kernel with the same block size, so the two kernels agree bit for bit.
template <int D, bool GENERIC, int NP>
__global__ void __launch_bounds__(512)
ens_half_multi_kernel(HalfArgs p) {
    __shared__ double q_s[NP][ALABI_MAX_DIM], qs_s[NP][ALABI_MAX_DIM], old_s[NP][ALABI_MAX_DIM];
    __shared__ double scratch[NP][16];
    __shared__ int ok_s[NP], w_s[NP];
    __shared__ double lpold_s[NP], lnfac_s[NP], lnu_s[NP];
    const int tid = threadIdx.x, T = blockDim.x;
    const int e = blockIdx.y;
    const int first = blockIdx.x * NP;
    const double* inv_len = p.consts;
    const double* lo = p.consts + ALABI_MAX_DIM;
    const double* hi = p.consts + 2 * ALABI_MAX_DIM;
    if (tid < NP) ok_s[tid] = 1;
    __syncthreads();
    // (1) the NP proposals: thread (pp, k) forms coordinate k of proposal pp
    for (int idx = tid; idx < NP * D; idx += T) {
        const int pp = idx / D, k = idx % D;
        int w = -1;
        if (first + pp < p.count) {
            const size_t pos = (size_t)e * p.W + (p.split ? p.n0 : 0) + p.part_begin + first + pp;
            w = p.rec.order[pos];
            if (w >= 0) {
                double qv = 0.0;
                if (k < p.d) {
                    const int cw = p.rec.cw[pos];
                    const double zz = p.rec.zz[pos];
                    const double cv = p.coords[(size_t)cw * p.d + k];
                    const double sv = p.coords[(size_t)w * p.d + k];
                    qv = cv - (cv - sv) * zz;
                    if (!((qv > lo[k]) && (qv < hi[k]))) ok_s[pp] = 0;
                    q_s[pp][k] = qv; old_s[pp][k] = sv;
                    qv *= inv_len[k];
                    if (!GENERIC) qv -= p.centre[k];
                }
                qs_s[pp][k] = qv;
                if (k == 0) { lpold_s[pp] = p.logp[w]; lnfac_s[pp] = p.rec.lnfac[pos]; lnu_s[pp] = p.rec.lnu[pos]; }
            }
        }
        if (k == 0) w_s[pp] = w;
    }
    __syncthreads();
    // (2) kernel sums: X pairs streamed once, NP accumulators
    double q[NP][D];
    bool live[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        live[pp] = w_s[pp] >= 0 && ok_s[pp];
#pragma unroll
        for (int k = 0; k < D; ++k) q[pp][k] = qs_s[pp][k];
    }
    double acc[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) acc[pp] = 0.0;
    const int half = p.Npad >> 1;
    bool first_pair = true;
    const double* Xsrc = GENERIC ? p.Xt : p.Xc;
    const double* Asrc = GENERIC ? p.alpha : p.ens_h;
    double nhq[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) nhq[pp] = GENERIC ? 0.0 : se_neg_half_norm<D>(q[pp]);
    for (int j = tid; j < half; j += T) {
        f64x2 x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = reinterpret_cast<const f64x2*>(Xsrc + (size_t)k * p.Npad)[j];
        const f64x2 al = reinterpret_cast<const f64x2*>(Asrc)[j];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            if (!GENERIC) {                                   // (se_pair_terms: the order of ens_half_kernel, acc += term_a; acc += term_b)
                double fa, fb;
                se_pair_terms<D>(x, al, q[pp], nhq[pp], fa, fb);
                acc[pp] += fa; acc[pp] += fb;
                continue;
            }
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double d0 = x[k].x - q[pp][k], d1 = x[k].y - q[pp][k];
                s0 = fma(d0, d0, s0);
                s1 = fma(d1, d1, s1);
            }
            // ens_half_kernel's order: the lane's first pair enters by a multiply, every later term by an fma
            acc[pp] = first_pair ? al.x * radial<GENERIC>(s0, p.kf) : fma(al.x, radial<GENERIC>(s0, p.kf), acc[pp]);
            acc[pp] = fma(al.y, radial<GENERIC>(s1, p.kf), acc[pp]);
        }
        first_pair = false;
    }
    // a lane without any pair (tid >= half) contributes aa = 0: 0 * f + 0 * f = 0 exactly, as in ens_half_kernel
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        const double wsum = wave_sum_dpp(acc[pp]);
        if ((tid & 63) == 63) scratch[pp][tid >> 6] = wsum;
    }
    __syncthreads();
    if (tid >= 64) return;
    // (3) wave 0: accept tests and state updates, one proposal after the other
    const int nw = T >> 6;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        const int w = w_s[pp];
        if (w < 0) continue;                              // beyond the launch's proposals, or an inert record
        double lp_new = -INFINITY;
        if (ok_s[pp]) {
            double part = (tid < nw) ? scratch[pp][tid] : 0.0;
            part = wave_sum_dpp(part);
            lp_new = fma(p.amp, lane_bcast(part, 63), p.mean);
            if (p.ymap) lp_new = apply_ymap(lp_new, p.ymap);
            if (p.has_prior)
                lp_new += normal_prior_sum(p.consts + 3 * ALABI_MAX_DIM, p.consts + 4 * ALABI_MAX_DIM, tid, p.d, q_s[pp][tid]) +
                          p.prior_const;
        }
        const double lp_old = lpold_s[pp];
        const int acc_flag = (lnfac_s[pp] + lp_new - lp_old > lnu_s[pp]) ? 1 : 0;
        if (acc_flag) {
            if (tid < p.d) p.coords[(size_t)w * p.d + tid] = q_s[pp][tid];
            if (tid == 0) {
                p.logp[w] = lp_new;
                if (p.n_accept) p.n_accept[w] += 1;
            }
        }
        if (p.chain || p.chain_logp) {
            const long long done = p.run_state[1] + p.local_t + 1;
            if (done % p.thin_by == 0) {
                const size_t slot = (size_t)(done / p.thin_by - 1);
                const size_t WT = (size_t)p.W * gridDim.y;
                if (p.chain && tid < p.d)
                    __builtin_nontemporal_store(acc_flag ? q_s[pp][tid] : old_s[pp][tid], &p.chain[(slot * WT + w) * p.d + tid]);
                if (p.chain_logp && tid == 0)
                    __builtin_nontemporal_store(acc_flag ? lp_new : lp_old, &p.chain_logp[slot * WT + w]);
            }
        }
    }
}

template <int D>
__global__ void __launch_bounds__(1024)
ens_lnprob_kernel(const double* __restrict__ coords, int d, const double* __restrict__ Xt,
                  const double* __restrict__ alpha, int Npad, double amp, double mean, KernelFn kf,
                  const double* __restrict__ consts, int has_prior, double prior_const, int ymap, int gate_box,
                  double* __restrict__ logp) {
    __shared__ double qs_s[ALABI_MAX_DIM];
    __shared__ double scratch[16];
    __shared__ double prior_s;
    const int tid = threadIdx.x, w = blockIdx.x;
    int ok = 1;
    double qraw = 0.0;
    if (tid < D) {
        double qv = 0.0;
        if (tid < d) {
            qv = coords[(size_t)w * d + tid];
            qraw = qv;
            ok = (qv > consts[ALABI_MAX_DIM + tid]) && (qv < consts[2 * ALABI_MAX_DIM + tid]);
            qv *= consts[tid];
        }
        qs_s[tid] = qv;
    }
    if (tid < 64) {
        const double pr = has_prior ? normal_prior_sum(consts + 3 * ALABI_MAX_DIM, consts + 4 * ALABI_MAX_DIM, tid, d, qraw) + prior_const
                                    : 0.0;
        if (tid == 0) prior_s = pr;
    }
    const int inb = __syncthreads_and(ok) || !gate_box;
    double lp = -INFINITY;
    if (inb) lp = apply_ymap(fma(amp, gp_kernel_dot_block<D>(Xt, alpha, Npad, qs_s, scratch, kf), mean), ymap) + prior_s;
    if (tid == 0) logp[w] = lp;
}

// ---------------------------------------------------------------------------------------------------
// Generic log-probability path: the reference's lnprob is like_fn(theta) + prior_fn(theta) with ARBITRARY Python
// callables (alabi/core.py:2073-2100, :2253-2280).  A half step is then split in two launches around the host call:
//   ens_propose_kernel  one workgroup per proposal of the half: forms q (same arithmetic as ens_half_kernel), writes it
//                       in list order, and -- when `like` is given -- the surrogate part y_scaler^-1(GP mean) at q
//                       (box-gated to -inf only if gate_box, i.e. when the prior is the uniform box itself);
//   (host)              lp_new = like + prior_fn(q)   [or like_fn(q) + prior_fn(q)]
//   ens_accept_kernel   one thread per proposal: accept test with the record's (d-1) ln z and ln u', state update.
// The ensemble, the draws and the accept decisions stay on the device; only the proposals of a half step travel.
template <int D, bool GENERIC>
__global__ void __launch_bounds__(1024)
ens_propose_kernel(HalfArgs p, int gate_box, double* __restrict__ q_out, double* __restrict__ like_out) {
    __shared__ double qs_s[ALABI_MAX_DIM];
    __shared__ double scratch[16];
    const int tid = threadIdx.x;
    const size_t pos = (size_t)(p.split ? p.n0 : 0) + p.part_begin + blockIdx.x;
    const int w = p.rec.order[pos];
    if (w < 0) {                                       // inert record: a NaN proposal is rejected by the accept kernel
        if (tid < p.d) q_out[(size_t)blockIdx.x * p.d + tid] = __longlong_as_double(0x7FF8000000000000ll);
        if (tid == 0 && like_out) like_out[blockIdx.x] = -INFINITY;
        return;
    }
    const int cw = p.rec.cw[pos];
    const double zz = p.rec.zz[pos];
    int ok = 1;
    if (tid < D) {
        double qv = 0.0;
        if (tid < p.d) {
            const double cv = p.coords[(size_t)cw * p.d + tid];
            const double sv = p.coords[(size_t)w * p.d + tid];
            qv = cv - (cv - sv) * zz;
            ok = (qv > p.consts[ALABI_MAX_DIM + tid]) && (qv < p.consts[2 * ALABI_MAX_DIM + tid]);
            q_out[(size_t)blockIdx.x * p.d + tid] = qv;
            qv *= p.consts[tid];
        }
        qs_s[tid] = qv;
    }
    const int inb = __syncthreads_and(ok) || !gate_box;
    if (!like_out) return;
    double lp = -INFINITY;
    if (inb) lp = apply_ymap(fma(p.amp, gp_kernel_dot_block<D, GENERIC>(p.Xt, p.alpha, p.Npad, qs_s, scratch, p.kf), p.mean), p.ymap);
    if (tid == 0) like_out[blockIdx.x] = lp;
}

__global__ void __launch_bounds__(256)
ens_accept_kernel(HalfArgs p, int count, const double* __restrict__ q, const double* __restrict__ lp_new) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const size_t pos = (size_t)(p.split ? p.n0 : 0) + p.part_begin + i;
    const int w = p.rec.order[pos];
    if (w < 0) return;
    const double lpn = lp_new[i], lpo = p.logp[w];
    if (p.rec.lnfac[pos] + lpn - lpo > p.rec.lnu[pos]) {      // false for NaN
        for (int k = 0; k < p.d; ++k) p.coords[(size_t)w * p.d + k] = q[(size_t)i * p.d + k];
        p.logp[w] = lpn;
        if (p.n_accept) p.n_accept[w] += 1;
    }
}

__global__ void ens_advance_kernel(long long* run_state, long long n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { run_state[0] += n; run_state[1] += n; }
}


// ---------------------------------------------------------------------------------------------------
// Persistent dataflow variant ("stream" kernel).  One workgroup per list position b (and ensemble) lives for
// K whole steps.  Its share of the training set (2 points x (D + 1) doubles per lane) is loaded ONCE and stays
// in VGPRs, so a proposal costs the distance/exp/reduction only.  There is no barrier between half steps: the
// state of every walker after every step is a row of the version history `hist[v][walker] = (coords, logp)`
// (v = number of steps that walker has completed; immutable once written), and a proposal waits only for the
// two rows it reads -- its own walker at version t and its partner at version t (+1 if the partner belongs to
// the half already updated in this step).  Hand-off follows cdna_hip_programming.md Guideline 16 form R2 (the
// data is the flag): rows of versions 1..K start as a sentinel NaN; every word is written by ONE aligned 8-byte
// write-through store (sc1 = relaxed agent-scope atomic store) and the consumer's lanes poll their own words
// with sc1 loads (L1 bypassed) until none is the sentinel -- no release fence, no drain, no separate flag word.
// Correctness never depends on placement or timing; every spin is bounded
// (a timeout sets *err and every workgroup leaves, the host then falls back to the launch-per-half-step path).
// All workgroups must be co-resident: the host launches at most one per CU.
#define ALABI_HIST_EMPTY 0x7FF8A1AB1D15EA5Eull   // quiet NaN with a payload no computation produces

struct StreamArgs {
    unsigned long long* hist;    // [(K+1)][E*W][d+1] as raw 64-bit words; rows 1..K pre-filled with ALABI_HIST_EMPTY
    int* err;                    // [1], zeroed before the launch
    DrawBuffers rec;             // chunk base
    const double* consts;
    const double* Xt;            // squared exponential: gp->Xc (inputs relative to their mean) ...
    const double* alpha;         // ... and gp->ens_h (se_pair_terms)
    const double* centre;        // mean of the scaled training inputs (squared exponential)
    double* chain;
    double* chain_logp;
    unsigned long long* n_accept;
    const long long* run_state;
    int K, W, n0, d, Npad, thin_by, spin_limit, has_prior;
    double amp, mean, prior_const;
    KernelFn kf;
};

__device__ inline unsigned long long ld_sc1(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void st_sc1(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef ALABI_STREAM_PROF
__device__ long long g_stream_prof[16];
#endif

// Sum of the wave partials s[0..15] (zeros beyond the last compute wave) in exactly the order wave_sum_dpp adds
// lanes 0..15 of a row -- a balanced binary tree -- so this path and ens_half_kernel agree bit for bit.
__device__ inline double wave_partials_tree(const double* s, int nw) {
    const f64x2 a = reinterpret_cast<const f64x2*>(s)[0], b = reinterpret_cast<const f64x2*>(s)[1];
    const f64x2 c = reinterpret_cast<const f64x2*>(s)[2], d = reinterpret_cast<const f64x2*>(s)[3];
    const double lo8 = ((d.y + d.x) + (c.y + c.x)) + ((b.y + b.x) + (a.y + a.x));
    if (nw <= 8) return lo8;
    const f64x2 e = reinterpret_cast<const f64x2*>(s)[4], f = reinterpret_cast<const f64x2*>(s)[5];
    const f64x2 g = reinterpret_cast<const f64x2*>(s)[6], h = reinterpret_cast<const f64x2*>(s)[7];
    const double hi8 = ((h.y + h.x) + (g.y + g.x)) + ((f.y + f.x) + (e.y + e.x));
    return hi8 + lo8;
}

// blockDim.x = 64 + compute threads (a multiple of 64) + 64, three roles:
//   wave 0          the ONLY wave on the hand-off chain: polls the two rows, forms the proposal, publishes it in LDS,
//                   and after the reduction does the accept test and the one row store.  It computes no kernel values
//                   and issues no other memory operation: gfx950 returns a wave's vector memory operations in issue
//                   order (one vmcnt), so any ordinary load or store would put its latency in front of the next poll.
//   waves 1..nwc    the training-set share of each lane lives in VGPRs for the whole launch; between the two barriers
//                   of a proposal they evaluate the kernel sum and leave one partial per wave in LDS.
//   last wave       fetches the packed proposal records (one 32-byte load) three proposals ahead into an LDS ring.
// The chain, the thinning and the acceptance counters are NOT written here: every version of every walker is a row of
// `hist` (coords, logp, accepted), and ens_hist_chain_kernel copies it out after the launch at HBM speed.
template <int D, int PPT, int TMAX, bool GENERIC>
__global__ void __launch_bounds__(TMAX)
ens_stream_kernel(StreamArgs p) {
    __shared__ __attribute__((aligned(16))) double scratch[2][16];   // wave partials, by proposal parity
    __shared__ unsigned long long rec_s[4][4];                       // proposal-record ring (last wave -> wave 0)
    __shared__ __attribute__((aligned(16))) double qs_s[2][ALABI_MAX_DIM];   // scaled proposal, by proposal parity
    __shared__ double consts_s[6][ALABI_MAX_DIM];                    // 1/length scale, lower, upper bound, prior mean, prior 1/std, centre
    __shared__ int ctl_s[2][2];                                      // [parity][0] proposal inside the box; [0][1] abort
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int TC = blockDim.x - 128, nwc = TC >> 6;
    const bool comm = wv == 0, service = wv == nwc + 1, compute = !comm && !service;
    const int b = blockIdx.x, e = blockIdx.y, E = gridDim.y;
    const int WT = p.W * E, row = p.d + 2;
#ifdef ALABI_STREAM_PROF
    long long prof[5] = {0, 0, 0, 0, 0};
    const long long prof_begin = clock64();
#endif
    // training-set share of this lane, resident for the whole launch (same lane -> point map as ens_half_kernel)
    const int half = p.Npad >> 1, ct = tid - 64;
    f64x2 xa[PPT][D], aa[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int idx = ct + j * TC;
        const bool v = compute && idx < half;
#pragma unroll
        for (int k = 0; k < D; ++k)
            xa[j][k] = v ? reinterpret_cast<const f64x2*>(p.Xt + (size_t)k * p.Npad)[idx] : f64x2{0.0, 0.0};
        aa[j] = v ? reinterpret_cast<const f64x2*>(p.alpha)[idx] : (GENERIC ? f64x2{0.0, 0.0} : f64x2{ALABI_SE_PAD, ALABI_SE_PAD});
    }
    if (tid < ALABI_MAX_DIM) {
        consts_s[0][tid] = (tid < p.d) ? p.consts[tid] : 0.0;
        consts_s[1][tid] = (tid < p.d) ? p.consts[ALABI_MAX_DIM + tid] : 0.0;
        consts_s[2][tid] = (tid < p.d) ? p.consts[2 * ALABI_MAX_DIM + tid] : 0.0;
        consts_s[3][tid] = (tid < p.d) ? p.consts[3 * ALABI_MAX_DIM + tid] : 0.0;
        consts_s[4][tid] = (tid < p.d) ? p.consts[4 * ALABI_MAX_DIM + tid] : 0.0;
        consts_s[5][tid] = (!GENERIC && tid < p.d) ? p.centre[tid] : 0.0;
    }
    if (tid < 32) scratch[tid >> 4][tid & 15] = 0.0;
    if (tid < 4) ctl_s[tid >> 1][tid & 1] = 0;

    // This workgroup's proposals: list positions b, b+G, b+2G, ... of every half step, in (step, split, position)
    // order.  Every dependency points to an EARLIER half step, so the globally oldest unfinished proposal can always
    // run: no deadlock as long as all G x E workgroups are resident.
    const int G = gridDim.x;
    auto next_item = [&](int& t, int& split, int& bb) {
        bb += G;
        for (;;) {
            if (t >= p.K) return;
            if (bb < (split ? p.W - p.n0 : p.n0)) return;
            bb = b;
            if (split == 0) split = 1; else { split = 0; ++t; }
        }
    };
    // Proposal records depend on nothing: the last wave fetches them three proposals ahead (lane l < 4 loads word l).
    const unsigned long long* packed = p.rec.packed;
    unsigned long long pend = 0;
    int pt = 0, psplit = 0, pbb = b, pslot = 0;
    auto record_load = [&](int t, int split, int bb) -> unsigned long long {
        if (t >= p.K || lane >= 4) return 0ull;
        const size_t pos = ((size_t)t * E + e) * p.W + (split ? p.n0 : 0) + bb;
        return packed[4 * pos + lane];
    };
    if (service) {                                    // prologue: items 0 and 1 into the ring, item 2 in flight
        for (int i = 0; i < 2; ++i) {
            const unsigned long long v = record_load(pt, psplit, pbb);
            if (lane < 4) rec_s[pslot & 3][lane] = v;
            next_item(pt, psplit, pbb); ++pslot;
        }
        pend = record_load(pt, psplit, pbb);
    }
    __syncthreads();
    int t = 0, split = 0, bb = b, item = 0;           // b < G <= n0: the first item is valid
    // Wave 0 idles while the compute waves work: it uses that time to decode the NEXT proposal's record and to form the
    // addresses it will poll and store to, so that nothing but the accept test separates barrier B from the row store and
    // the store from the next poll.
    int n_w = 0; double n_zz = 0.0, n_lnfac = 0.0, n_lnu = 0.0;
    const unsigned long long *n_hw = p.hist, *n_hc = p.hist;
    auto decode_next = [&](int it, int tt, int sp) {
        const unsigned long long* rs = rec_s[it & 3];
        const unsigned long long ids = rs[0];
        n_w = (int)(unsigned)(ids & 0xffffffffull);
        const int cw = (int)(unsigned)(ids >> 32);
        n_zz = __longlong_as_double((long long)rs[1]);
        n_lnfac = __longlong_as_double((long long)rs[2]); n_lnu = __longlong_as_double((long long)rs[3]);
        // own row at version t, partner row at version t (+1 when the partner's half went first)
        n_hw = p.hist + ((size_t)tt * WT + n_w) * row + lane;
        n_hc = p.hist + ((size_t)(tt + sp) * WT + cw) * row + lane;
    };
    if (comm) decode_next(0, 0, 0);
    while (t < p.K) {
        int t2 = t, s2 = split, b2 = bb;
        next_item(t2, s2, b2);
        const int par = item & 1;
        int w = 0, all_in = 0;
        double qv = 0.0, sv = 0.0, lnfac = 0.0, lnu = 0.0;   // wave 0; lane k < d: coordinate k, lane d: logp
#ifdef ALABI_STREAM_PROF
        const long long c0 = clock64(); long long c1 = c0;
#endif
        unsigned long long* out_row = nullptr;
        if (comm) {
            w = n_w; lnfac = n_lnfac; lnu = n_lnu;
            const double zz = n_zz;
            const unsigned long long *hw = n_hw, *hc = n_hc;    // this lane's words of the two rows
            const double il_r = consts_s[0][lane], lo_r = consts_s[1][lane], hi_r = consts_s[2][lane], c_r = consts_s[5][lane];
            // The data IS the flag (Guideline 16 form R2): every word of a row is one aligned 8-byte sc1 store
            // over a sentinel NaN that no coordinate or log-probability can equal; lane k polls its own words.
            unsigned long long ws = ALABI_HIST_EMPTY, wc = ALABI_HIST_EMPTY;
            int ok = 1, spins = 0;
            const bool mine = lane <= p.d, needc = lane < p.d;
            while (true) {
                if (mine && ws == ALABI_HIST_EMPTY) ws = ld_sc1(hw);
                if (needc && wc == ALABI_HIST_EMPTY) wc = ld_sc1(hc);
                const int ready = (!mine || ws != ALABI_HIST_EMPTY) && (!needc || wc != ALABI_HIST_EMPTY);
                if (__all(ready)) break;
                if (++spins > p.spin_limit ||
                    ((spins & 63) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    ok = 0;
                    break;
                }
            }
#ifdef ALABI_STREAM_PROF
            c1 = clock64();
#endif
            int inb = 1;
            double qs = 0.0;
            if (ok && mine) {
                sv = __longlong_as_double((long long)ws);
                if (needc) {
                    const double cv = __longlong_as_double((long long)wc);
                    qv = cv - (cv - sv) * zz;
                    inb = (qv > lo_r) && (qv < hi_r);
                    qs = qv * il_r;
                    if (!GENERIC) qs -= c_r;
                }
            }
            all_in = ok ? __all(inb) : 0;
            if (lane < D) qs_s[par][lane] = qs;
            if (lane == 63) qs_s[par][63] = all_in ? 1.0 : 0.0;   // the in-bounds flag travels with the proposal (D <= 16)
            if (lane == 0) {
                if (!ok) {                            // bounded spin ran out: every workgroup leaves, the host falls back
                    ctl_s[0][1] = 1;
                    __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();                              // barrier A: the proposal is in LDS
#ifdef ALABI_STREAM_PROF
        const long long c2 = clock64();
#endif
        double qraw[D];                               // one batch of LDS reads: the proposal and its in-bounds flag
#pragma unroll
        for (int k = 0; k < D; ++k) qraw[k] = qs_s[par][k];
        if (!comm) all_in = __builtin_amdgcn_readfirstlane(__double2hiint(qs_s[par][63])) != 0;
        if (compute && all_in) {
            double q[D];                              // wave-uniform: moved to SGPRs
#pragma unroll
            for (int k = 0; k < D; ++k)
                q[k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(qraw[k])),
                                        __builtin_amdgcn_readfirstlane(__double2loint(qraw[k])));
            double acc = 0.0;
            if (!GENERIC) {
                const double nhq = se_neg_half_norm<D>(q);
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    double fa, fb;
                    se_pair_terms<D>(xa[j], aa[j], q, nhq, fa, fb);
                    acc += fa; acc += fb;
                    if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // four points in flight at a time
                }
            } else
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                double r2a = 0.0, r2b = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double da = xa[j][k].x - q[k], db = xa[j][k].y - q[k];
                    r2a = fma(da, da, r2a);
                    r2b = fma(db, db, r2b);
                }
                // same operation order as ens_half_kernel's lane (first pair by multiply, the rest by fma)
                acc = (j == 0) ? aa[j].x * radial<GENERIC>(r2a, p.kf) : fma(aa[j].x, radial<GENERIC>(r2a, p.kf), acc);
                acc = fma(aa[j].y, radial<GENERIC>(r2b, p.kf), acc);
                if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // four points in flight at a time: enough independent
                                                                       // chains to cover the fp64 latency, bounded temporaries
            }
            const double wsum = wave_sum_dpp(acc);
            if (lane == 63) scratch[par][wv - 1] = wsum;
        }
        double prior_q = 0.0;                         // normal-prior term of this proposal (0.0 adds exactly nothing)
        if (comm) {                                   // idle until barrier B: prepare the store and the next proposal
            if (p.has_prior) prior_q = normal_prior_sum(consts_s[3], consts_s[4], lane, p.d, qv) + p.prior_const;
            out_row = p.hist + ((size_t)(t + 1) * WT + w) * row + lane;
            if (t2 < p.K) decode_next(item + 1, t2, s2);
        }
        if (service) {                                // under the compute waves' kernel sum: ring slot item+2, issue item+3
            if (lane < 4) rec_s[pslot & 3][lane] = pend;
            next_item(pt, psplit, pbb); ++pslot;
            pend = record_load(pt, psplit, pbb);
        }
        __syncthreads();                              // barrier B: the wave partials are in LDS
#ifdef ALABI_STREAM_PROF
        const long long c3 = clock64();
#endif
        if (ctl_s[0][1]) return;
        if (comm) {
            double lp_new = -INFINITY;
            if (all_in) lp_new = fma(p.amp, wave_partials_tree(scratch[par], nwc), p.mean) + prior_q;
            const double lp_old = lane_bcast(sv, p.d);                // lane d loaded logp
            const int acc_flag = (lnfac + lp_new - lp_old > lnu) ? 1 : 0;
            // new row of the walker: lanes k < d coordinates, lane d logp, lane d+1 the acceptance flag
            const double outv = (lane < p.d) ? (acc_flag ? qv : sv) : (acc_flag ? lp_new : lp_old);
            const unsigned long long outw = (lane <= p.d) ? (unsigned long long)__double_as_longlong(outv)
                                                          : (unsigned long long)acc_flag;
            if (lane <= p.d + 1) st_sc1(out_row, outw);
        }
        t = t2; split = s2; bb = b2; ++item;
#ifdef ALABI_STREAM_PROF
        { const long long c4 = clock64();
          prof[0] += c1 - c0; prof[1] += c2 - c1; prof[2] += c3 - c2; prof[3] += c4 - c3; prof[4] += 1; }
#endif
    }
#ifdef ALABI_STREAM_PROF
    if (tid == 0 && blockIdx.x == 3 && blockIdx.y == 0) {
        for (int i = 0; i < 5; ++i) g_stream_prof[i] = prof[i];
        g_stream_prof[5] = clock64() - prof_begin;
    }
#endif
}

#ifdef ALABI_STREAM_PROF
extern "C" int alabi_debug_stream_prof(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stream_prof), sizeof(long long) * 16);
}
#endif

// hist[0] <- (coords, logp); and back: (coords, logp) <- hist[K]
__global__ void __launch_bounds__(256)
ens_hist_fill_kernel(unsigned long long* __restrict__ h, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) h[i] = ALABI_HIST_EMPTY;
}

__global__ void __launch_bounds__(256)
ens_hist_copy_kernel(double* __restrict__ coords, double* __restrict__ logp, unsigned long long* __restrict__ hist_row,
                     int WT, int d, int to_hist) {
    const int i = blockIdx.x * 256 + threadIdx.x, row = d + 2;
    if (i >= WT * row) return;
    const int w = i / row, k = i % row;
    if (k > d) { if (to_hist) hist_row[i] = 0ull; return; }           // acceptance flag of version 0: unused
    double* src = (k < d) ? coords + (size_t)w * d + k : logp + w;
    if (to_hist) hist_row[i] = (unsigned long long)__double_as_longlong(*src);
    else *src = __longlong_as_double((long long)hist_row[i]);
}

// After the persistent kernel: versions 1..K of every walker -> the (thinned) chain, and the acceptance counters.
// One thread per (version, walker, word); a few MB at HBM speed per launch of K steps.
__global__ void __launch_bounds__(256)
ens_hist_chain_kernel(const unsigned long long* __restrict__ hist, int K, int WT, int d, int thin_by,
                      const long long* __restrict__ run_state, const int* __restrict__ err, double* __restrict__ chain,
                      double* __restrict__ chain_logp, unsigned long long* __restrict__ n_accept) {
    if (*err) return;                                  // timed out: the rows are incomplete, the host reruns the chunk
    const int row = d + 2;
    const size_t n = (size_t)K * WT * row;
    const long long done0 = run_state[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % row);
        const size_t vw = i / row;
        const int w = (int)(vw % WT), v = (int)(vw / WT) + 1;
        const unsigned long long word = hist[(size_t)WT * row + i];
        if (k == d + 1) { if (word == 1ull && n_accept) atomicAdd(n_accept + w, 1ull); continue; }
        const long long done = done0 + v;
        if (done % thin_by != 0) continue;
        const size_t slot = (size_t)(done / thin_by - 1);
        if (k < d) { if (chain) chain[(slot * WT + w) * d + k] = __longlong_as_double((long long)word); }
        else if (chain_logp) chain_logp[slot * WT + w] = __longlong_as_double((long long)word);
    }
}

// Persistent-kernel configuration: T compute lanes (+ the hand-off wave and the record wave), PPT point pairs per lane.
// The limits are the largest dimension buckets that compile without VGPR spills
// (hipcc -Rpass-analysis=kernel-resource-usage; 256 VGPRs at 384 threads, 168 at 640).
static int ens_stream_ppt(const alabi_ens* e) {
    const int T = e->threads, half = e->gp->Npad / 2, db = dim_bucket(e->d);
    if ((T != 256 && T != 512) || e->d > 61 || db < 0 || db > 16) return 0;
    if (e->ymap != 0) return 0;   // non-affine y scalers (pow) run on the launch-per-half-step path: this kernel has no VGPR to spare
    const int ppt = (half + T - 1) / T;
    const bool generic = e->gp->kf.type != 0;
    int max_db = 0;
    if (T == 512) max_db = (ppt == 1) ? (generic ? 12 : 16) : (ppt == 2) ? (generic ? 6 : 10) : 0;
    else max_db = (ppt == 1) ? 16 : (ppt == 2) ? (generic ? 12 : 16) : (ppt == 3) ? (generic ? 8 : 12)
                                                                      : (ppt == 4) ? (generic ? 6 : 10) : 0;
    return db <= max_db ? ppt : 0;
}

bool ens_stream_fits(const alabi_ens* e) { return ens_stream_ppt(e) > 0; }

#define ALABI_STREAM_DISPATCH_DIM(DB, ...)                            \
    switch (DB) {                                                     \
        case 1: { constexpr int D = 1; __VA_ARGS__; } break;          \
        case 2: { constexpr int D = 2; __VA_ARGS__; } break;          \
        case 3: { constexpr int D = 3; __VA_ARGS__; } break;          \
        case 4: { constexpr int D = 4; __VA_ARGS__; } break;          \
        case 5: { constexpr int D = 5; __VA_ARGS__; } break;          \
        case 6: { constexpr int D = 6; __VA_ARGS__; } break;          \
        case 8: { constexpr int D = 8; __VA_ARGS__; } break;          \
        case 10: { constexpr int D = 10; __VA_ARGS__; } break;        \
        case 12: { constexpr int D = 12; __VA_ARGS__; } break;        \
        case 16: { constexpr int D = 16; __VA_ARGS__; } break;        \
        default: return ALABI_BAD_ARGUMENT;                           \
    }
#define ALABI_STREAM_LAUNCH(PPT_, TMAX_)                                                                          \
    ALABI_STREAM_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(gp->kf.type,                                              \
        hipLaunchKernelGGL((ens_stream_kernel<D, PPT_, TMAX_, GENERIC>), dim3(e->stream_grid, e->E), dim3(T + 128), 0, s, a)))

// Version history around a persistent launch of K steps: rows 1..K <- sentinel (`fill`: when the rows are polled), row 0 <- (coords, logp) before it;
// (coords, logp) <- row K, chain / counters <- rows 1..K after it.  Shared by ens_stream_kernel and ens_group_kernel.
int ens_se_prepare(alabi_gp* gp, hipStream_t s) {
    if (gp->ens_h_gen == gp->gen && gp->Xc && gp->ens_h) return ALABI_OK;
    int st = ensure_xa(gp, s);                         // the centre of the scaled training inputs (per factor)
    if (st != ALABI_OK) return st;
    const int rows = dim_bucket(gp->d);
    if (!gp->Xc) ALABI_HIP_CHECK(hipMalloc(&gp->Xc, (size_t)ALABI_MAX_DIM * gp->n_cap * sizeof(double)));
    if (!gp->ens_h) ALABI_HIP_CHECK(hipMalloc(&gp->ens_h, (size_t)gp->n_cap * sizeof(double)));
    hipLaunchKernelGGL(ens_se_prepare_kernel, dim3((gp->Npad + 255) / 256), dim3(256), 0, s, gp->Xt, gp->xa_centre, gp->alpha, gp->N,
                       gp->Npad, gp->d, rows, gp->Xc, gp->ens_h);
    ALABI_LAUNCH_CHECK();
    gp->ens_h_gen = gp->gen;
    return ALABI_OK;
}

int launch_ens_hist_prologue(alabi_ens* e, double* coords, double* logp, int K, bool fill, hipStream_t s) {
    const int WT = e->W * e->E, row = e->d + 2;
    if (fill) hipLaunchKernelGGL(ens_hist_fill_kernel, dim3(1024), dim3(256), 0, s, e->hist + (size_t)WT * row, (size_t)K * WT * row);
    hipLaunchKernelGGL(ens_hist_copy_kernel, dim3((WT * row + 255) / 256), dim3(256), 0, s, coords, logp, e->hist, WT, e->d, 1);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_hist_epilogue(alabi_ens* e, double* coords, double* logp, int K, int thin_by, double* chain, double* chain_logp,
                             long long* n_accept, hipStream_t s) {
    const int WT = e->W * e->E, row = e->d + 2;
    hipLaunchKernelGGL(ens_hist_copy_kernel, dim3((WT * row + 255) / 256), dim3(256), 0, s, coords, logp,
                       e->hist + (size_t)K * WT * row, WT, e->d, 0);
    if (chain || chain_logp || n_accept)
        hipLaunchKernelGGL(ens_hist_chain_kernel, dim3(2048), dim3(256), 0, s, e->hist, K, WT, e->d, thin_by, e->run_state,
                           e->err, chain, chain_logp, reinterpret_cast<unsigned long long*>(n_accept));
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_stream(alabi_ens* e, double* coords, double* logp, int K, int thin_by, double* chain, double* chain_logp,
                      long long* n_accept, hipStream_t s) {
    alabi_gp* gp = e->gp;
    const int n0 = (e->W + 1) / 2;
    { const int st0 = launch_ens_hist_prologue(e, coords, logp, K, true, s); if (st0 != ALABI_OK) return st0; }
    StreamArgs a{};
    a.hist = e->hist; a.err = e->err; a.rec = e->draws; a.consts = e->consts;
    const bool se = gp->kf.type == 0;                 // squared exponential: centred inputs and h (se_pair_terms), built by ens_se_prepare
    a.Xt = se ? gp->Xc : gp->Xt; a.alpha = se ? gp->ens_h : gp->alpha; a.centre = gp->xa_centre;
    a.chain = chain; a.chain_logp = chain_logp;
    a.n_accept = reinterpret_cast<unsigned long long*>(n_accept); a.run_state = e->run_state;
    a.K = K; a.W = e->W; a.n0 = n0; a.d = e->d; a.Npad = gp->Npad; a.thin_by = thin_by; a.spin_limit = 1 << 20;
    if (const char* env = getenv("ALABI_ENS_SPIN_LIMIT")) { const int v = atoi(env); if (v > 0) a.spin_limit = v; }   // tests: force a time-out
    a.amp = e->lp_scale * exp(gp->log_amp); a.mean = fma(e->lp_scale, gp->mean, e->lp_shift); a.kf = gp->kf;
    a.has_prior = e->has_prior; a.prior_const = e->prior_const;
    const int db = dim_bucket(e->d);
    // lanes x pairs-per-lane cover Npad/2 point pairs; the launch-per-half-step kernel's lane -> point map (and so its
    // summation order) is reproduced exactly because both run with e->threads compute lanes.
    const int T = e->threads, ppt = ens_stream_ppt(e);
    e->last_path = 1;
    if (T == 256) {
        if (ppt == 1) { ALABI_STREAM_LAUNCH(1, 384); }
        else if (ppt == 2) { ALABI_STREAM_LAUNCH(2, 384); }
        else if (ppt == 3) { ALABI_STREAM_LAUNCH(3, 384); }
        else if (ppt == 4) { ALABI_STREAM_LAUNCH(4, 384); }
        else return ALABI_BAD_ARGUMENT;
    } else if (T == 512) {
        if (ppt == 1) { ALABI_STREAM_LAUNCH(1, 640); }
        else if (ppt == 2) { ALABI_STREAM_LAUNCH(2, 640); }
        else return ALABI_BAD_ARGUMENT;
    } else return ALABI_BAD_ARGUMENT;
    ALABI_LAUNCH_CHECK();
    return launch_ens_hist_epilogue(e, coords, logp, K, thin_by, chain, chain_logp, n_accept, s);
}

int launch_ens_draw(alabi_ens* e, int nsteps, double a, hipStream_t s) {
    int Wp = 1;
    while (Wp < e->W) Wp <<= 1;
    const size_t lds = (size_t)Wp * 12 + 1024;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        ALABI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ens_draw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(ens_draw_kernel, dim3(nsteps, e->E), dim3(256), lds, s, e->seed, e->run_state, e->W, Wp, e->d, a,
                       e->draws);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_prep(alabi_ens* e, const int* order, int n0, const double* u_z, const int* partner,
                    const double* u_acc, double a, hipStream_t s) {
    hipLaunchKernelGGL(ens_prep_kernel, dim3((e->W + 255) / 256), dim3(256), 0, s, order, n0, e->W, u_z, partner, u_acc,
                       a, e->d, e->draws);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_half_args(alabi_ens* e, const HalfArgs& args_in, int nblocks, hipStream_t s) {
    if (nblocks <= 0) return ALABI_OK;
    const int db = dim_bucket(e->d);
    const int threads = e->threads;
    HalfArgs args = args_in;
    args.count = nblocks;
    // more proposals than CUs: NP per workgroup share one pass over the training set (same block size: same bits)
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    const long long total = (long long)nblocks * e->E;
    const char* env = getenv("ALABI_ENS_MULTI");
    int np = 1;
    if (threads <= 512 && db <= 24 && !(env && env[0] == '0') && !args.shist) {
        if (total > 3LL * n_cu && db <= 16) np = 4; else if (total > n_cu) np = 2;   // q[NP][D] lives in registers
        if (env && env[0] == '4' && db <= 16) np = 4;
        if (env && env[0] == '2') np = 2;
        if (env && env[0] == '1') np = 1;
    }
    if (np == 4) {
        ALABI_STREAM_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(e->gp->kf.type, hipLaunchKernelGGL((ens_half_multi_kernel<D, GENERIC, 4>),
            dim3((nblocks + 3) / 4, e->E), dim3(threads), 0, s, args)));
    } else if (np == 2) {
        ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(e->gp->kf.type, hipLaunchKernelGGL((ens_half_multi_kernel<D, GENERIC, 2>),
            dim3((nblocks + 1) / 2, e->E), dim3(threads), 0, s, args)));
    } else {
        ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(e->gp->kf.type, hipLaunchKernelGGL((ens_half_kernel<D, GENERIC>), dim3(nblocks, e->E), dim3(threads), 0, s, args)));
    }
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_lnprob(alabi_ens* e, const double* coords, int nwalkers, double* logp, int gate_box, hipStream_t s) {
    const int db = dim_bucket(e->d);
    alabi_gp* gp = e->gp;
    ALABI_DISPATCH_DIM(db, hipLaunchKernelGGL(ens_lnprob_kernel<D>, dim3(nwalkers), dim3(e->threads), 0, s, coords, e->d,
                                              gp->Xt, gp->alpha, gp->Npad, e->lp_scale * exp(gp->log_amp),
                                              fma(e->lp_scale, gp->mean, e->lp_shift), gp->kf, e->consts, e->has_prior,
                                              e->prior_const, e->ymap, gate_box, logp));
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_propose(alabi_ens* e, const HalfArgs& args, int nblocks, int gate_box, double* q, double* like, hipStream_t s) {
    if (nblocks <= 0) return ALABI_OK;
    const int db = dim_bucket(e->d);
    ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(e->gp->kf.type, hipLaunchKernelGGL((ens_propose_kernel<D, GENERIC>), dim3(nblocks),
                                                                                    dim3(e->threads), 0, s, args, gate_box, q, like)));
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_accept(alabi_ens* e, const HalfArgs& args, int count, const double* q, const double* lp_new, hipStream_t s) {
    if (count <= 0) return ALABI_OK;
    hipLaunchKernelGGL(ens_accept_kernel, dim3((count + 255) / 256), dim3(256), 0, s, args, count, q, lp_new);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_advance(alabi_ens* e, long long n, hipStream_t s) {
    hipLaunchKernelGGL(ens_advance_kernel, dim3(1), dim3(64), 0, s, e->run_state, n);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
