// Persistent ensemble kernel for ensembles whose training set does NOT fit one workgroup's registers
// (N > 2048, or d > 10-16): BASELINE configurations C4 (N = 5000, 1024 walkers) and C5-sized (N = 10000, d = 20).
//
// Same semantics as ens_stream_kernel / ens_half_kernel (emcee's red-blue stretch move driven by alabi/core.py:2319-2325
// with the log-probability of alabi/core.py:2073-2100; CPU statement: oracle/stretch_oracle.py), another blocking:
//
//   * the scaled training set is STATIONARY: the chip's workgroups form NG groups of G members; member m of every group
//     keeps slice m of the (augmented, centred) training rows Xa in LDS for the whole launch (S = Npad / G points, laid out
//     as ready-made B operands of v_mfma_f64_16x16x4);
//   * the proposals MOVE: group g owns list positions [g QP, (g+1) QP) of every half step (QP = 16 Q, Q query tiles).  All
//     its members poll the 2 QP walker rows those proposals read, form the QP proposals (bit-identical arithmetic to the
//     other paths), and evaluate their kernel sums over their own slice on the matrix cores: the exponent -r^2/2 is ONE
//     augmented dot product q'.x' (q' = (q/l - c, 1, -|q/l - c|^2/2), x' = (x/l - c, -|x/l - c|^2/2, 1)), the vector unit
//     only runs the table exp and the alpha FMA (11 instead of 41 fp64 instructions per kernel evaluation);
//   * the G partial sums of a proposal travel through memory exactly like the walker rows do: one aligned 8-byte sc1 store
//     per word over a sentinel NaN, polled with sc1 loads -- the data is the flag (cdna_hip_programming.md Guideline 16,
//     form R2); member m adds the G partials of proposals m, m + G, ... in a FIXED order (j = 0 .. G-1), does their accept
//     tests and publishes the new walker rows in the same version history the other persistent kernel uses.
//
// There is no grid-wide barrier: a workgroup waits only for the rows / partials it reads, every dependency points to an
// earlier half step (or to the same half step's partials, which depend on earlier rows only), so with all workgroups
// resident (at most one per CU) the oldest unfinished half step can always complete.  Every spin is bounded; on a time-out
// the launch sets *err, every workgroup leaves and alabi_ens_run repeats the chunk on the launch-per-half-step path.
//
// Summation order differs from the other paths (tiles of 16 points per wave, waves, members), so chains agree with them
// and with the oracle to rounding (tests: chain <= 1e-7 over hundreds of steps, identical acceptance counts), not bit for
// bit; they are reproducible run to run for a given (W, N, d, #CUs).
//
// Per half step the dependency chain is: rows visible (hop) -> proposals -> kernel sums -> partials visible (hop) ->
// accept -> row stores.  Work per half step at C4: 512 x 5000 kernel evaluations = 40 tile products per workgroup
// (3 MFMA + 44 VALU instructions each) = 1.5 us of the fp64 pipe.
#include <cstdlib>
#include <vector>
#include "gp_device.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define ALABI_GRP_EMPTY 0x7FF8A1AB1D15EA5Eull   // the sentinel of the version history (ensemble.hip: ALABI_HIST_EMPTY)
#define ALABI_GRP_MAXW 8                         // waves per workgroup (512 threads)

struct GroupArgs {
    unsigned long long* hist;            // [(K+1)][E*W][d+2] version history, rows 1..K pre-filled with the sentinel
    unsigned long long* part;            // [2K][E][NG][G][QPAD] partial kernel sums, pre-filled with the sentinel
    int* err;                            // [1] time-out flag
    const unsigned long long* packed;    // proposal records of the chunk: [K][E][W][4] (walker | partner << 32, z, (d-1) ln z, ln u')
    const double* consts;                // [5][ALABI_MAX_DIM]: 1/length scale, lower, upper, prior mean, prior 1/std
    const double* Xa;                    // [4 KS][Npad] augmented centred training rows
    const double* centre;                // [d] centre of the scaled training inputs
    const double* alpha;                 // [Npad]
    int K, W, n0, d, Npad;
    int NG, G, QP, S;                    // groups, members per group, proposals per group and half step, points per member (x16)
    int xcd_map;                         // 1: members of a group share blockIdx % 8 (one XCD under round-robin placement; speed only)
    int spin_limit, has_prior;
    double amp, mean, prior_const;
    KernelFn kf;
};

__device__ inline unsigned long long grp_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void grp_st(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum over the lanes of a 16-lane DPP row, result in EVERY lane of the row, identical bits in all of them
// (row_ror 8, 4, 2, 1: each step adds the same two operands in both lanes of a pair, and a + b == b + a).
__device__ inline double row16_allsum(double v) {
    v += dpp_move<0x128, 0xf>(v);   // row_ror:8
    v += dpp_move<0x124, 0xf>(v);   // row_ror:4
    v += dpp_move<0x122, 0xf>(v);   // row_ror:2
    v += dpp_move<0x121, 0xf>(v);   // row_ror:1
    return v;
}

// LDS layout (units of 8 bytes), the same formula on the host (ens_group_lds_words).
struct GroupLds {
    int etab, xb, al, aop, wsum, rec, lpo, prior, inb, mq, mo, pj, dec, ctl, total;
};
__host__ __device__ inline GroupLds group_lds(int KS, int QPAD, int S, int G, int d) {
    GroupLds L;
    const int nmine = (QPAD + G - 1) / G;
    int o = 0;
    L.etab = o; o += 64;
    L.xb = o; o += S * KS * 4;            // S/16 tiles x KS k-steps x 64 lanes
    L.al = o; o += S;
    L.aop = o; o += QPAD * KS * 4;
    L.wsum = o; o += ALABI_GRP_MAXW * QPAD;
    L.rec = o; o += 4 * QPAD * 4;         // ring of 4 half steps
    L.lpo = o; o += 2 * QPAD;
    L.prior = o; o += 2 * QPAD;
    L.inb = o; o += 2 * QPAD;             // in-box flags (as 8-byte words)
    L.mq = o; o += 2 * nmine * d;         // proposals / old coordinates of the walkers THIS member decides, by half-step parity
    L.mo = o; o += 2 * nmine * d;
    L.pj = o; o += nmine * G;
    L.dec = o; o += 2 * nmine;            // decisions: new logp, accept flag
    L.ctl = o; o += 2;
    L.total = o;
    return L;
}

template <int KS, int Q, bool GENERIC>
__global__ void __launch_bounds__(512)
ens_group_kernel(GroupArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char grp_smem[];
    double* lds = reinterpret_cast<double*>(grp_smem);
    constexpr int QPAD = 16 * Q, KP = 4 * KS;
    const GroupLds L = group_lds(KS, QPAD, p.S, p.G, p.d);
    double* etab = lds + L.etab;
    double* xb = lds + L.xb;
    double* al_s = lds + L.al;
    double* aop = lds + L.aop;
    double* wsum = lds + L.wsum;
    unsigned long long* rec_s = reinterpret_cast<unsigned long long*>(lds + L.rec);
    double* lpo_s = lds + L.lpo;
    double* prior_s = lds + L.prior;
    unsigned long long* inb_s = reinterpret_cast<unsigned long long*>(lds + L.inb);
    double* mq_s = lds + L.mq;
    double* mo_s = lds + L.mo;
    double* pj_s = lds + L.pj;
    double* dec_s = lds + L.dec;
    int* ctl_s = reinterpret_cast<int*>(lds + L.ctl);

    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6, NW = T >> 6;
    const int e = blockIdx.y, E = gridDim.y;
    const int G = p.G, NG = p.NG, d = p.d, row = d + 2, WT = p.W * E;
    int g, m;
    {
        const int b = blockIdx.x, B = NG * G;
        if (p.xcd_map && (B & 7) == 0 && ((B >> 3) % G) == 0) {
            const int xcd = b & 7, slot = b >> 3;
            g = xcd * ((B >> 3) / G) + slot / G;
            m = slot % G;
        } else {
            g = b / G;
            m = b % G;
        }
    }
    // ---- one-time set-up: the member's slice of Xa as B operands, alpha, the exp table, constants ----
    const int tiles_all = p.Npad >> 4;
    const int tile0 = m * (p.S >> 4);
    int ntile = tiles_all - tile0;
    if (ntile > (p.S >> 4)) ntile = p.S >> 4;
    if (ntile < 0) ntile = 0;
    for (int i = tid; i < ntile * KS * 64; i += T) {
        const int ln = i & 63, s = (i >> 6) % KS, tl = (i >> 6) / KS;
        xb[i] = p.Xa[(size_t)(4 * s + (ln >> 4)) * p.Npad + (size_t)(tile0 + tl) * 16 + (ln & 15)];
    }
    for (int i = tid; i < ntile * 16; i += T) al_s[i] = p.alpha[(size_t)tile0 * 16 + i];
    if (tid < 64) etab[tid] = exp2((double)tid * 0.015625);
    if (tid < 2) ctl_s[tid] = 0;
    const int LPR = (d + 2 <= 16) ? 16 : 32;                 // lanes per proposal in the row / proposal phase
    const int lshift = (LPR == 16) ? 4 : 5;
    const int k = tid & (LPR - 1);                           // this thread's word of a row: k < d coordinate, k == d logp
    const int PPP = T >> lshift;                             // proposals per pass over the workgroup
    const int npass = (QPAD + PPP - 1) / PPP;
    const double il_r = (k < d) ? p.consts[k] : 0.0, lo_r = (k < d) ? p.consts[ALABI_MAX_DIM + k] : 0.0;
    const double hi_r = (k < d) ? p.consts[2 * ALABI_MAX_DIM + k] : 0.0;
    const double pm_r = (k < d) ? p.consts[3 * ALABI_MAX_DIM + k] : 0.0, pi_r = (k < d) ? p.consts[4 * ALABI_MAX_DIM + k] : 0.0;
    const double c_r = (k < d) ? p.centre[k] : 0.0;
    const double SC = GENERIC ? 1.0 : ALABI_EXP2S_SCALE;

    // ---- proposal records: ring of 4 half steps in LDS, fetched three half steps ahead (plain loads: written before the launch)
    const int n1 = p.W - p.n0;
    auto half_count = [&](int hh) { const int nh = (hh & 1) ? n1 : p.n0; int c = nh - g * p.QP; c = c < 0 ? 0 : c; return c > p.QP ? p.QP : c; };
    auto rec_load = [&](int hh, int i) -> unsigned long long {       // word i of the group's record block of half step hh
        if (hh >= 2 * p.K || i >= 4 * half_count(hh)) return 0xFFFFFFFFFFFFFFFFull;
        const size_t pos0 = ((size_t)(hh >> 1) * E + e) * p.W + ((hh & 1) ? p.n0 : 0) + (size_t)g * p.QP;
        return p.packed[4 * pos0 + i];
    };
    constexpr int RW = 4 * QPAD;                             // record words per half step
    unsigned long long pend[(RW + 255) / 256];               // T >= 256
    for (int hh = 0; hh < 2; ++hh)
        for (int i = tid; i < RW; i += T) rec_s[(hh & 3) * RW + i] = rec_load(hh, i);
#pragma unroll
    for (int j = 0; j < (RW + 255) / 256; ++j) pend[j] = (tid + j * T < RW) ? rec_load(2, tid + j * T) : 0ull;
    __syncthreads();

    const int tpw = (ntile + NW - 1) / NW;                   // point tiles per wave
    const int tl_begin = wv * tpw, tl_end = (tl_begin + tpw < ntile) ? tl_begin + tpw : ntile;
    const int lr = lane & 15, lk = lane >> 4;
    const int nmine_max = (QPAD + G - 1) / G;

    for (int hh = 0; hh < 2 * p.K; ++hh) {
        const int t = hh >> 1, split = hh & 1, par = hh & 1;
        const int cnt = half_count(hh);
        const unsigned long long* rs = rec_s + (hh & 3) * RW;
        // ---- phase 1: poll the rows, form the proposals, publish the A operands in LDS ----
        int ok = 1;
        for (int ps = 0; ps < npass; ++ps) {
            const int pp = ps * PPP + (tid >> lshift);
            const bool valid = pp < cnt;
            double sv = 0.0, qv = 0.0, zz = 0.0;
            if (valid) {
                const unsigned long long ids = rs[4 * pp];
                const int w = (int)(unsigned)(ids & 0xffffffffull), cw = (int)(unsigned)(ids >> 32);
                zz = __longlong_as_double((long long)rs[4 * pp + 1]);
                const unsigned long long* hw = p.hist + ((size_t)t * WT + w) * row + k;
                const unsigned long long* hc = p.hist + ((size_t)(t + split) * WT + cw) * row + k;
                const bool mine = k <= d, needc = k < d;
                unsigned long long ws = mine ? ALABI_GRP_EMPTY : 0ull, wc = needc ? ALABI_GRP_EMPTY : 0ull;
                int spins = 0;
                while (true) {
                    if (ws == ALABI_GRP_EMPTY) ws = grp_ld(hw);
                    if (wc == ALABI_GRP_EMPTY) wc = grp_ld(hc);
                    if (ws != ALABI_GRP_EMPTY && wc != ALABI_GRP_EMPTY) break;
                    if (++spins > p.spin_limit ||
                        ((spins & 63) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        ok = 0;
                        break;
                    }
                }
                sv = __longlong_as_double((long long)ws);
                if (needc) {
                    const double cv = __longlong_as_double((long long)wc);
                    qv = cv - (cv - sv) * zz;
                }
            }
            // lanes of one proposal: k = 0 .. LPR-1.  In-box test, |q - c|^2 and the normal-prior term by segmented reductions.
            const int out = (valid && k < d) ? !((qv > lo_r) && (qv < hi_r)) : 0;
            const unsigned long long om = __ballot(out);
            const unsigned long long seg = (LPR == 16) ? ((om >> (lane & 48)) & 0xffffull) : ((om >> (lane & 32)) & 0xffffffffull);
            const double qs = (valid && k < d) ? qv * il_r - c_r : 0.0;
            double qq = row16_allsum(qs * qs);
            double pr = 0.0;
            if (p.has_prior) {
                double tt = (valid && k < d) ? (qv - pm_r) * pi_r : 0.0;
                pr = row16_allsum(-0.5 * tt * tt);
            }
            if (LPR == 32) {
                qq += __shfl_xor(qq, 16, 64);
                if (p.has_prior) pr += __shfl_xor(pr, 16, 64);
            }
            if (pp < QPAD && k < KP) {
                double av = 0.0;
                if (valid) av = (k < d) ? qs * SC : (k == d) ? SC : (k == d + 1) ? -0.5 * qq * SC : 0.0;
                aop[pp * KP + k] = av;
            }
            if (valid) {
                if (k == 0) { inb_s[par * QPAD + pp] = (seg == 0ull) ? 1ull : 0ull; prior_s[par * QPAD + pp] = pr + p.prior_const; }
                if (k == d) lpo_s[par * QPAD + pp] = sv;
                if (k < d && (pp % G) == m) {
                    const int i = pp / G;
                    mq_s[(par * nmine_max + i) * d + k] = qv;
                    mo_s[(par * nmine_max + i) * d + k] = sv;
                }
            }
        }
        if (!ok) {
            ctl_s[0] = 1;
            __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();                                     // barrier A: A operands (and the abort word) are in LDS
        if (ctl_s[0]) return;
        // records: slot hh+2 from the registers, issue hh+3 (lands under this half step's kernel sums)
#pragma unroll
        for (int j = 0; j < (RW + 255) / 256; ++j)
            if (tid + j * T < RW) rec_s[((hh + 2) & 3) * RW + tid + j * T] = pend[j];
#pragma unroll
        for (int j = 0; j < (RW + 255) / 256; ++j) pend[j] = (tid + j * T < RW) ? rec_load(hh + 3, tid + j * T) : 0ull;
        // ---- phase 2: kernel sums of the QP proposals over this member's slice, on the matrix cores ----
        double a[Q][KS];
#pragma unroll
        for (int qt = 0; qt < Q; ++qt)
#pragma unroll
            for (int s = 0; s < KS; ++s) a[qt][s] = aop[(qt * 16 + lr) * KP + 4 * s + lk];
        v4f64 sum[Q];
#pragma unroll
        for (int qt = 0; qt < Q; ++qt) sum[qt] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
        for (int tl = tl_begin; tl < tl_end; ++tl) {
            const double* xbt = xb + (size_t)tl * KS * 64 + lane;
            double bop[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) bop[s] = xbt[s * 64];
            const double al = al_s[tl * 16 + lr];
#pragma unroll
            for (int qt = 0; qt < Q; ++qt) {
                v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qt][s], bop[s], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {                // C/D layout: row (proposal) lk + 4 i, column (point) lr
                    const double f = GENERIC ? radial<true>(fmax(-2.0 * acc[i], 0.0), p.kf) : exp2s_tab64(acc[i], etab);
                    sum[qt][i] = fma(al, f, sum[qt][i]);
                }
            }
        }
#pragma unroll
        for (int qt = 0; qt < Q; ++qt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double v = sum[qt][i];
                v += dpp_move<0x111, 0xf>(v);
                v += dpp_move<0x112, 0xf>(v);
                v += dpp_move<0x114, 0xf>(v);
                v += dpp_move<0x118, 0xf>(v);
                if (lr == 15) wsum[wv * QPAD + qt * 16 + lk + 4 * i] = v;
            }
        __syncthreads();                                     // barrier B: the wave partials are in LDS
        if (wv != 0) continue;                               // waves 1.. go straight to the next half step's rows
        // ---- phase 3 (wave 0): this member's partial sums, published like rows ----
        unsigned long long* part_h = p.part + (((size_t)hh * E + e) * NG + g) * (size_t)G * QPAD;
        for (int pp = lane; pp < cnt; pp += 64) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += wsum[w * QPAD + pp];
            grp_st(part_h + (size_t)m * QPAD + pp, (unsigned long long)__double_as_longlong(s));
        }
        // ---- phase 4 (wave 0): proposals m, m + G, ...: gather the G partials, accept test, new rows ----
        const int nm = (cnt > m) ? (cnt - m + G - 1) / G : 0;
        int ok4 = 1;
        for (int base = 0; base < nm * G; base += 64) {
            const int idx = base + lane;
            if (idx < nm * G) {
                const int i = idx / G, j = idx % G;
                const unsigned long long* src = part_h + (size_t)j * QPAD + (m + i * G);
                unsigned long long v = ALABI_GRP_EMPTY;
                int spins = 0;
                while (true) {
                    v = grp_ld(src);
                    if (v != ALABI_GRP_EMPTY) break;
                    if (++spins > p.spin_limit ||
                        ((spins & 63) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        ok4 = 0;
                        break;
                    }
                }
                pj_s[idx] = __longlong_as_double((long long)v);
            }
        }
        if (!__all(ok4)) {                                   // the other waves see the abort word at their next barrier A
            if (lane == 0) { ctl_s[0] = 1; __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nm; i += 64) {
            const int pp = m + i * G;
            double s = 0.0;
            for (int j = 0; j < G; ++j) s += pj_s[i * G + j];
            double lp_new = -INFINITY;
            if (inb_s[par * QPAD + pp]) lp_new = fma(p.amp, s, p.mean) + prior_s[par * QPAD + pp];
            const double lp_old = lpo_s[par * QPAD + pp];
            const double lnfac = __longlong_as_double((long long)rs[4 * pp + 2]);
            const double lnu = __longlong_as_double((long long)rs[4 * pp + 3]);
            const int acc_flag = (lnfac + lp_new - lp_old > lnu) ? 1 : 0;
            dec_s[2 * i] = acc_flag ? lp_new : lp_old;
            dec_s[2 * i + 1] = acc_flag ? 1.0 : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < nm * row; idx += 64) {
            const int i = idx / row, kk = idx % row, pp = m + i * G;
            const int w = (int)(unsigned)(rs[4 * pp] & 0xffffffffull);
            const bool accd = dec_s[2 * i + 1] != 0.0;
            unsigned long long outw;
            if (kk < d) outw = (unsigned long long)__double_as_longlong(accd ? mq_s[(par * nmine_max + i) * d + kk]
                                                                              : mo_s[(par * nmine_max + i) * d + kk]);
            else if (kk == d) outw = (unsigned long long)__double_as_longlong(dec_s[2 * i]);
            else outw = accd ? 1ull : 0ull;
            grp_st(p.hist + ((size_t)(t + 1) * WT + w) * row + kk, outw);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side: the blocking (Q, G, NG) for an ensemble, buffers, launch.
struct GroupPlan {
    int ok, KS, Q, QP, G, NG, S, threads;
    size_t lds_bytes;
};

static int group_n_cu() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

// Cheapest feasible blocking: Q query tiles per group -> NG = ceil(n0 / 16 Q) groups, G = workgroups available per group;
// cost = fp64-pipe cycles of a member's kernel sums per half step (Q tile products per point tile, 64 cycles per MFMA k-step
// + 176 for the four exp / alpha FMAs of a lane) over the 4 SIMDs.
static GroupPlan group_plan(const alabi_ens* e) {
    GroupPlan best{};
    const alabi_gp* gp = e->gp;
    const int d = e->d;
    if (d + 2 > 32 || e->ymap != 0 || e->W < 2) return best;
    const int n_cu = group_n_cu();
    if (e->E > n_cu) return best;
    const int avail = n_cu / e->E;
    const int KS = (d + 2 + 3) / 4;
    const int n0 = (e->W + 1) / 2;
    const int tiles = gp->Npad / 16;
    int force_q = 0, force_g = 0;
    if (const char* env = getenv("ALABI_ENS_GROUP_Q")) force_q = atoi(env);
    if (const char* env = getenv("ALABI_ENS_GROUP_G")) force_g = atoi(env);
    double best_cost = 0.0;
    for (int Q = 1; Q <= 8; Q *= 2) {
        if (force_q && Q != force_q) continue;
        const int QP = 16 * Q;
        const int NG = (n0 + QP - 1) / QP;
        if (NG > avail) continue;
        int G = avail / NG;
        if (G > 64) G = 64;
        if (G > tiles) G = tiles;
        if (force_g && force_g <= G) G = force_g;
        if (G < 1) continue;
        const int S = ((tiles + G - 1) / G) * 16;
        const GroupLds L = group_lds(KS, QP, S, G, d);
        const size_t bytes = (size_t)L.total * 8;
        if (bytes > 160 * 1024 - 1024) continue;
        const double cost = (double)Q * (S / 16) * (KS * 64 + 176) / 4.0 + 40.0 * G;
        if (!best.ok || cost < best_cost) {
            best.ok = 1; best.KS = KS; best.Q = Q; best.QP = QP; best.G = G; best.NG = NG; best.S = S; best.lds_bytes = bytes;
            best_cost = cost;
        }
    }
    if (best.ok) {
        best.threads = 512;
        if (const char* env = getenv("ALABI_ENS_GROUP_THREADS")) { const int v = atoi(env); if (v == 256 || v == 512) best.threads = v; }
    }
    return best;
}

bool ens_group_fits(const alabi_ens* e) { return e->hist && e->err && group_plan(e).ok; }

__global__ void __launch_bounds__(256)
ens_group_fill_kernel(unsigned long long* __restrict__ h, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) h[i] = ALABI_GRP_EMPTY;
}

#define ALABI_GROUP_DISPATCH_Q(Q_, ...)                      \
    switch (Q_) {                                            \
        case 1: { constexpr int Q = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int Q = 2; __VA_ARGS__; } break; \
        case 4: { constexpr int Q = 4; __VA_ARGS__; } break; \
        case 8: { constexpr int Q = 8; __VA_ARGS__; } break; \
        default: return ALABI_BAD_ARGUMENT;                  \
    }
#define ALABI_GROUP_DISPATCH_KS(KS_, ...)                      \
    switch (KS_) {                                             \
        case 1: { constexpr int KS = 1; __VA_ARGS__; } break;  \
        case 2: { constexpr int KS = 2; __VA_ARGS__; } break;  \
        case 3: { constexpr int KS = 3; __VA_ARGS__; } break;  \
        case 4: { constexpr int KS = 4; __VA_ARGS__; } break;  \
        case 5: { constexpr int KS = 5; __VA_ARGS__; } break;  \
        case 6: { constexpr int KS = 6; __VA_ARGS__; } break;  \
        case 7: { constexpr int KS = 7; __VA_ARGS__; } break;  \
        case 8: { constexpr int KS = 8; __VA_ARGS__; } break;  \
        default: return ALABI_BAD_ARGUMENT;                    \
    }

template <int KS, int Q, bool GENERIC>
static int group_launch(const GroupArgs& a, const GroupPlan& pl, int E, hipStream_t s) {
    auto kern = ens_group_kernel<KS, Q, GENERIC>;
    static bool attr_set = false;                            // per instantiation
    if (!attr_set) {
        ALABI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.NG * pl.G, E), dim3(pl.threads), pl.lds_bytes, s, a);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_ens_group(alabi_ens* e, double* coords, double* logp, int K, int thin_by, double* chain, double* chain_logp,
                     long long* n_accept, hipStream_t s) {
    alabi_gp* gp = e->gp;
    const GroupPlan pl = group_plan(e);
    if (!pl.ok) return ALABI_BAD_ARGUMENT;
    int st = ensure_xa(gp, s);
    if (st != ALABI_OK) return st;
    const size_t part_words = (size_t)2 * e->chunk_cap * e->E * pl.NG * pl.G * pl.QP;
    if (e->part_words < part_words) {
        if (e->part) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(e->part); e->part = nullptr; e->part_words = 0; }
        ALABI_HIP_CHECK(hipMalloc(&e->part, part_words * sizeof(unsigned long long)));
        e->part_words = part_words;
    }
    hipLaunchKernelGGL(ens_group_fill_kernel, dim3(2048), dim3(256), 0, s, e->part, (size_t)2 * K * e->E * pl.NG * pl.G * pl.QP);
    if ((st = launch_ens_hist_prologue(e, coords, logp, K, s)) != ALABI_OK) return st;
    GroupArgs a{};
    a.hist = e->hist; a.part = e->part; a.err = e->err; a.packed = e->draws.packed; a.consts = e->consts;
    a.Xa = gp->Xa; a.centre = gp->xa_centre; a.alpha = gp->alpha;
    a.K = K; a.W = e->W; a.n0 = (e->W + 1) / 2; a.d = e->d; a.Npad = gp->Npad;
    a.NG = pl.NG; a.G = pl.G; a.QP = pl.QP; a.S = pl.S;
    a.xcd_map = 1;
    if (const char* env = getenv("ALABI_ENS_GROUP_XCD")) a.xcd_map = env[0] != '0';
    a.spin_limit = 1 << 20;
    if (const char* env = getenv("ALABI_ENS_SPIN_LIMIT")) { const int v = atoi(env); if (v > 0) a.spin_limit = v; }   // tests: force a time-out
    a.has_prior = e->has_prior; a.prior_const = e->prior_const;
    a.amp = e->lp_scale * exp(gp->log_amp); a.mean = fma(e->lp_scale, gp->mean, e->lp_shift); a.kf = gp->kf;
    e->last_path = 3;
    e->group_q = pl.Q; e->group_g = pl.G; e->group_ng = pl.NG;
    ALABI_GROUP_DISPATCH_KS(pl.KS, ALABI_GROUP_DISPATCH_Q(pl.Q, ALABI_DISPATCH_KERNEL(gp->kf.type,
        { st = group_launch<KS, Q, GENERIC>(a, pl, e->E, s); })));
    if (st != ALABI_OK) return st;
    return launch_ens_hist_epilogue(e, coords, logp, K, thin_by, chain, chain_logp, n_accept, s);
}

}  // namespace alabi
