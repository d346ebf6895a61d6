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
//     its members form the QP proposals (bit-identical arithmetic to the other paths) and evaluate their kernel sums over
//     their own slice on the matrix cores: the exponent -r^2/2 is ONE augmented dot product q'.x'
//     (q' = (q/l - c, 1, -|q/l - c|^2/2), x' = (x/l - c, -|x/l - c|^2/2, 1)), the vector unit only runs the table exp and
//     the alpha FMA (11 instead of 41 fp64 instructions per kernel evaluation);
//   * ONE memory hop per half step.  A member publishes two things, each word by one aligned 8-byte sc1 store over a
//     sentinel NaN that the readers poll with sc1 loads (the data is the flag: cdna_hip_programming.md Guideline 16, form
//     R2): before its kernel sums the CANDIDATE of every proposal it is responsible for (proposal, old coordinates, old
//     log-probability, (d-1) ln z, ln u', prior term: everything the accept test needs except the kernel sum), after them
//     its PARTIAL sums.  Nobody waits for an accept decision: whoever needs the row of a walker reads the candidate of the
//     proposal that produced it plus its G partials, adds them in a FIXED order (j = 0 .. G-1) and repeats the accept test
//     -- the same bits in every reader.  Which proposal produced which row follows from the draws alone (ens_link_kernel,
//     before the launch).  The version history `hist` (the chain) is written off the dependency chain with plain stores,
//     by the group that reads a walker's row as its own one step later, and for the last step by a tail pass.
//
// There is no grid-wide barrier: a workgroup waits only for the words it reads, every dependency points to an earlier
// half step, so with all workgroups resident (at most one per CU) the oldest unfinished half step can always complete.
// Every spin is bounded; on a time-out the launch sets *err, every workgroup leaves and alabi_ens_run repeats the chunk on
// the launch-per-half-step path.
//
// Summation order differs from the other paths (tiles of 16 points per wave, waves, members), so chains agree with them
// and with the oracle to rounding (tests: chain <= 1e-7 over hundreds of steps, identical acceptance counts), not bit for
// bit; they are reproducible run to run for a given (W, N, d, #CUs).
//
// Per half step the dependency chain is: partials visible (hop) -> accept tests -> proposals -> kernel sums -> partial
// stores.  Work per half step at C4: 512 x 5000 kernel evaluations = 40 tile products per workgroup (3 MFMA + 44 VALU
// instructions each) = 1.5 us of the fp64 pipe.
#include <cstdlib>
#include <vector>
#include "gp_device.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define ALABI_GRP_EMPTY 0x7FF8A1AB1D15EA5Eull   // the sentinel of the version history (ensemble.hip: ALABI_HIST_EMPTY)
#define ALABI_GRP_MAXW 8                         // waves per workgroup (512 threads)
#define ALABI_GRP_NPJ 2                          // partial words a lane may have to gather per pass (2 PPW G / 64)

struct GroupArgs {
    unsigned long long* hist;            // [(K+1)][E*W][d+2] version history: row 0 = state before the launch, rows 1..K written here
    unsigned long long* part;            // [2K][E][NG][QPAD][G] partial kernel sums, pre-filled with the sentinel
    unsigned long long* cand;            // [2K][E][n0][2d+4] candidates, pre-filled with the sentinel
    int* err;                            // [1] time-out flag
    const unsigned long long* packed;    // proposal records of the chunk: [K][E][W][4] (walker | partner << 32, z, (d-1) ln z, ln u')
    const unsigned long long* link;      // [K][E][W] producers of the two rows a proposal reads (ens_link_kernel)
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
__device__ inline unsigned long long grp_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ inline double grp_dbl(unsigned long long v) { return __longlong_as_double((long long)v); }

// Workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the
// write-through candidate / partial stores and the record prefetch, none of which anybody reads through LDS.
__device__ inline void grp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

// Where a row was produced: the proposal at position `ph` of the active list of chunk-local half step `hp`
// (source = hp << 16 | ph), or -1 for version 0 (the state before the launch, hist row 0).  One thread per list position.
__global__ void __launch_bounds__(256)
ens_link_kernel(DrawBuffers b, int W, int n0) {
    const int t = blockIdx.x, e = blockIdx.y, E = gridDim.y;
    const size_t base = ((size_t)t * E + e) * W;
    const int g0 = e * W;
    for (int pos = threadIdx.x; pos < W; pos += 256) {
        const int wl = b.order[base + pos] - g0, cl = b.cw[base + pos] - g0;
        const int split = pos >= n0;
        auto src = [&](int version, int xl) -> int {
            if (version == 0) return -1;
            const int pp = b.pos_of[((size_t)(version - 1) * E + e) * W + xl];
            const int set = pp >= n0;
            return ((2 * (version - 1) + set) << 16) | (pp - set * n0);
        };
        b.link[base + pos] = (unsigned long long)(unsigned)src(t, wl) | ((unsigned long long)(unsigned)src(t + split, cl) << 32);
    }
}

// LDS layout (units of 8 bytes), the same formula on the host.
struct GroupLds {
    int etab, xb, al, aop, wsum, rec, pw, rd, ctl, total;
};
__host__ __device__ inline GroupLds group_lds(int KS, int QPAD, int S, int G, int d) {
    const int rows = (d + 2 <= 16) ? 8 : 4;   // rows (own + partner) a wave rebuilds per pass
    GroupLds L;
    int o = 0;
    L.etab = o; o += 64;
    L.xb = o; o += S * KS * 4;            // S/16 tiles x KS k-steps x 64 lanes
    L.al = o; o += S;
    L.aop = o; o += QPAD * KS * 4;
    L.wsum = o; o += ALABI_GRP_MAXW * QPAD;
    L.rec = o; o += 4 * QPAD * 5;         // ring of 4 half steps: 4 record words + 1 link word per proposal
    L.pw = o; o += ALABI_GRP_MAXW * rows * G;   // per wave: the G partials of its rows
    L.rd = o; o += ALABI_GRP_MAXW * 16;   // per wave: (logp, accepted) of up to 8 rows
    L.ctl = o; o += 2;
    L.total = o;
    return L;
}

#ifdef ALABI_GROUP_PROF
// phase stamps (s_memrealtime, 100 MHz) of one workgroup, accumulated over the half steps of the last launch:
// [0] rows phase (poll + accept tests) [1] proposals + barrier A [2] kernel sums [3] barrier B + partial stores [4] half steps
// for the wave that forms the first proposals (NW - 1); [8..12] the same for wave 0 (which also publishes the partials)
__device__ long long g_group_prof[16];
extern "C" int alabi_debug_group_prof(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_group_prof), sizeof(long long) * 16);
}
#define GRP_STAMP(x) const long long x = (long long)__builtin_amdgcn_s_memrealtime()
#else
#define GRP_STAMP(x)
#endif

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
    int* ctl_s = reinterpret_cast<int*>(lds + L.ctl);

    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6, NW = T >> 6;
    const int e = blockIdx.y, E = gridDim.y;
    const int G = p.G, NG = p.NG, d = p.d, row = d + 2, WT = p.W * E, CW = 2 * d + 4;
    double* pw_w = lds + L.pw + wv * ((d + 2 <= 16) ? 8 : 4) * G;   // wave-private
    double* rd_w = lds + L.rd + wv * 16;
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
    const int k = lane & (LPR - 1);                          // this lane's word of a row: k < d coordinate, k == d logp
    const int pl = lane >> lshift;                           // proposal slot within the wave
    const int PPW = 64 >> lshift, PPP = NW * PPW;            // proposals per wave / per pass over the workgroup
    const int npass = (QPAD + PPP - 1) / PPP;
    const int pwv = NW - 1 - wv;                             // the LAST wave takes the first proposals: wave 0 publishes the partials
    const double il_r = (k < d) ? p.consts[k] : 0.0, lo_r = (k < d) ? p.consts[ALABI_MAX_DIM + k] : 0.0;
    const double hi_r = (k < d) ? p.consts[2 * ALABI_MAX_DIM + k] : 0.0;
    const double pm_r = (k < d) ? p.consts[3 * ALABI_MAX_DIM + k] : 0.0, pi_r = (k < d) ? p.consts[4 * ALABI_MAX_DIM + k] : 0.0;
    const double c_r = (k < d) ? p.centre[k] : 0.0;
    const double SC = GENERIC ? 1.0 : ALABI_EXP2S_SCALE;
    // partial words this lane gathers per pass: word u is partial j_u of row r_u (rows 2 s, 2 s + 1: own / partner row of slot s)
    const int npj = (2 * PPW * G + 63) >> 6;
    int pj_r[ALABI_GRP_NPJ], pj_j[ALABI_GRP_NPJ];
#pragma unroll
    for (int u = 0; u < ALABI_GRP_NPJ; ++u) {
        const int idx = lane + 64 * u;
        pj_r[u] = idx / G;
        pj_j[u] = idx - pj_r[u] * G;
        if (u >= npj || pj_r[u] >= 2 * PPW) pj_r[u] = -1;
    }

    // ---- proposal records: ring of 4 half steps in LDS, fetched three half steps ahead (plain loads: written before the launch)
    const int n1 = p.W - p.n0;
    auto half_count = [&](int hh) { const int nh = (hh & 1) ? n1 : p.n0; int c = nh - g * p.QP; c = c < 0 ? 0 : c; return c > p.QP ? p.QP : c; };
    constexpr int RW = 5 * QPAD;                             // ring words per half step
    auto rec_load = [&](int hh, int i) -> unsigned long long {       // word i of the group's record block of half step hh
        if (hh >= 2 * p.K) return 0xFFFFFFFFFFFFFFFFull;
        const int c = half_count(hh);
        const size_t pos0 = ((size_t)(hh >> 1) * E + e) * p.W + ((hh & 1) ? p.n0 : 0) + (size_t)g * p.QP;
        if (i < 4 * QPAD) return (i < 4 * c) ? p.packed[4 * pos0 + i] : 0xFFFFFFFFFFFFFFFFull;
        return (i - 4 * QPAD < c) ? p.link[pos0 + (i - 4 * QPAD)] : 0xFFFFFFFFFFFFFFFFull;
    };
    // wave 0 owns the ring: its other memory traffic is the partial stores, so the counted wait in front of the LDS write
    // never sits behind a candidate store or a poll of the row phase
    constexpr int NRL = (RW + 63) / 64;
    unsigned long long pend[NRL];
    if (wv == 0) {
        for (int hh = 0; hh < 2; ++hh)
            for (int i = lane; i < RW; i += 64) rec_s[(hh & 3) * RW + i] = rec_load(hh, i);
#pragma unroll
        for (int j = 0; j < NRL; ++j) pend[j] = (lane + j * 64 < RW) ? rec_load(2, lane + j * 64) : 0ull;
    }
    for (int i = tid; i < ALABI_GRP_MAXW * QPAD; i += T) wsum[i] = 0.0;   // waves beyond NW contribute nothing
    __syncthreads();

    const int tpw = (ntile + NW - 1) / NW;                   // point tiles per wave
    const int tl_begin = wv * tpw, tl_end = (tl_begin + tpw < ntile) ? tl_begin + tpw : ntile;
    const int lr = lane & 15, lk = lane >> 4;

    // Rows of this wave's proposals in ALL passes of a half step, rebuilt from candidates + partials (wave-level: every lane
    // of the wave takes part).  Pass ps, slot pl is proposal ps PPP + pwv PPW + pl of the record block `rs`.  Lane (slot,
    // word k) gets word k of the own row in sv_[ps] (k < d coordinate, k == d logp), coordinate k of the partner row in
    // cv_[ps], and acc_[ps] = was the own row's producing proposal accepted.  A source < 0 is hist row 0.  All first looks of
    // all passes are issued before any is examined: one memory round trip when everything is there.
    // tail_hf >= 0: the rows the proposals of half step tail_hf PRODUCED are wanted (own rows only, source (tail_hf, position)).
    constexpr int NPM = (Q == 1) ? 1 : (Q == 2) ? 2 : 4;     // passes (the host only picks blockings with npass <= NPM)
    int w_[NPM], cw_[NPM], so_[NPM], sp_[NPM];
    double sv_[NPM], cv_[NPM];
    int acc_[NPM];
#ifdef ALABI_GROUP_PROF
    long long prof_poll = 0, prof_setup = 0;
#endif
    auto rows_phase = [&](const unsigned long long* rs, int cnt, int tail_hf) -> int {
        const bool want_partner = tail_hf < 0;
        if (pwv * PPW >= cnt) {                              // wave-uniform: none of this wave's slots holds a proposal
#pragma unroll
            for (int ps = 0; ps < NPM; ++ps) { w_[ps] = 0; cw_[ps] = 0; so_[ps] = -1; sp_[ps] = -1; sv_[ps] = 0.0; cv_[ps] = 0.0; acc_[ps] = 0; }
            return 1;
        }
        unsigned long long v_[NPM][4];
        double pv_[NPM][ALABI_GRP_NPJ];
        const unsigned long long* a_[NPM][4];
        const unsigned long long* pa_[NPM][ALABI_GRP_NPJ];
        unsigned want_[NPM], got_[NPM];
#pragma unroll
        for (int ps = 0; ps < NPM; ++ps) {
            const int ppbase = ps * PPP + pwv * PPW, pp = ppbase + pl;
            const bool valid = ps < npass && pp < cnt;
            unsigned want = 0u;
            w_[ps] = 0; cw_[ps] = 0; so_[ps] = -1; sp_[ps] = -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) { a_[ps][i] = nullptr; v_[ps][i] = 0ull; }
            if (valid) {
                const unsigned long long ids = rs[4 * pp];
                w_[ps] = (int)(unsigned)(ids & 0xffffffffull); cw_[ps] = (int)(unsigned)(ids >> 32);
                if (tail_hf >= 0) {
                    so_[ps] = (tail_hf << 16) | (g * p.QP + pp);
                } else {
                    const unsigned long long lw = rs[4 * QPAD + pp];
                    so_[ps] = (int)(unsigned)(lw & 0xffffffffull); sp_[ps] = (int)(unsigned)(lw >> 32);
                }
                const int src_o = so_[ps], src_p = sp_[ps];
                const unsigned long long* co = (src_o >= 0) ? p.cand + (((size_t)(src_o >> 16) * E + e) * p.n0 + (src_o & 0xffff)) * CW : nullptr;
                const unsigned long long* cp = (want_partner && src_p >= 0)
                                                   ? p.cand + (((size_t)(src_p >> 16) * E + e) * p.n0 + (src_p & 0xffff)) * CW : nullptr;
                if (k < d) {
                    if (co) { a_[ps][0] = co + k; a_[ps][1] = co + d + k; want |= 3u; }
                    else { a_[ps][1] = p.hist + (size_t)w_[ps] * row + k; want |= 2u; }
                    if (want_partner) {
                        if (cp) { a_[ps][2] = cp + k; a_[ps][3] = cp + d + k; want |= 12u; }
                        else { a_[ps][3] = p.hist + (size_t)cw_[ps] * row + k; want |= 8u; }
                    }
                } else if (k == d) {
                    if (co) { a_[ps][0] = co + 2 * d; a_[ps][1] = a_[ps][0] + 1; a_[ps][2] = a_[ps][0] + 2; a_[ps][3] = a_[ps][0] + 3; want |= 15u; }
                    else { a_[ps][0] = p.hist + (size_t)w_[ps] * row + d; want |= 1u; }
                } else if (k == d + 1 && cp) {
                    a_[ps][0] = cp + 2 * d; a_[ps][1] = a_[ps][0] + 1; a_[ps][2] = a_[ps][0] + 2; a_[ps][3] = a_[ps][0] + 3; want |= 15u;
                }
            }
            // partial words: row r_u belongs to slot r_u >> 1 of this wave (own row: even, partner row: odd)
#pragma unroll
            for (int u = 0; u < ALABI_GRP_NPJ; ++u) {
                pa_[ps][u] = nullptr; pv_[ps][u] = 0.0;
                if (ps < npass && pj_r[u] >= 0) {
                    const int sl = pj_r[u] >> 1, which = pj_r[u] & 1;
                    if (ppbase + sl < cnt && (which == 0 || want_partner)) {
                        int src;
                        if (tail_hf >= 0) {
                            src = (tail_hf << 16) | (g * p.QP + ppbase + sl);
                        } else {
                            const unsigned long long lw = rs[4 * QPAD + ppbase + sl];
                            src = which ? (int)(unsigned)(lw >> 32) : (int)(unsigned)(lw & 0xffffffffull);
                        }
                        if (src >= 0) {
                            const int ph = src & 0xffff, gq = ph / QPAD, pq = ph - gq * QPAD;
                            pa_[ps][u] = p.part + ((((size_t)(src >> 16) * E + e) * NG + gq) * QPAD + pq) * G + pj_j[u];
                            want |= 16u << u;
                        }
                    }
                }
            }
            want_[ps] = want; got_[ps] = 0u;
        }
        int spins = 0, ok = 1;
#ifdef ALABI_GROUP_PROF
        prof_setup = (long long)__builtin_amdgcn_s_memrealtime();
#endif
        while (true) {
            unsigned long long tv[NPM][4], tp[NPM][ALABI_GRP_NPJ];
            unsigned miss[NPM];
#pragma unroll
            for (int ps = 0; ps < NPM; ++ps) {               // every look of this round is issued ...
                miss[ps] = want_[ps] & ~got_[ps];
#pragma unroll
                for (int i = 0; i < 4; ++i) { tv[ps][i] = 0ull; if (miss[ps] & (1u << i)) tv[ps][i] = grp_ld(a_[ps][i]); }
#pragma unroll
                for (int u = 0; u < ALABI_GRP_NPJ; ++u) { tp[ps][u] = 0ull; if (miss[ps] & (16u << u)) tp[ps][u] = grp_ld(pa_[ps][u]); }
            }
            bool all = true;
#pragma unroll
            for (int ps = 0; ps < NPM; ++ps) {               // ... before any is examined
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((miss[ps] & (1u << i)) && tv[ps][i] != ALABI_GRP_EMPTY) { v_[ps][i] = tv[ps][i]; got_[ps] |= 1u << i; }
#pragma unroll
                for (int u = 0; u < ALABI_GRP_NPJ; ++u)
                    if ((miss[ps] & (16u << u)) && tp[ps][u] != ALABI_GRP_EMPTY) { pv_[ps][u] = grp_dbl(tp[ps][u]); got_[ps] |= 16u << u; }
                all = all && got_[ps] == want_[ps];
            }
            if (all) break;
            if (++spins > p.spin_limit ||
                ((spins & 63) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                ok = 0;
                break;
            }
        }
#ifdef ALABI_GROUP_PROF
        prof_poll = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int ps = 0; ps < NPM; ++ps) {
            const int ppbase = ps * PPP + pwv * PPW, pp = ppbase + pl;
            sv_[ps] = 0.0; cv_[ps] = 0.0; acc_[ps] = 0;
            if (ps >= npass || ppbase >= cnt) continue;      // wave-uniform
            const bool valid = pp < cnt;
#pragma unroll
            for (int u = 0; u < ALABI_GRP_NPJ; ++u)
                if (pa_[ps][u]) pw_w[pj_r[u] * G + pj_j[u]] = pv_[ps][u];
            __builtin_amdgcn_wave_barrier();
            // accept tests of the producing proposals: lane k == d for the own row, lane k == d + 1 for the partner row
            if (valid && (k == d || (k == d + 1 && want_partner))) {
                const int r = 2 * pl + (k - d);
                const int src = (k == d) ? so_[ps] : sp_[ps];
                double lp = grp_dbl(v_[ps][0]), flag = 0.0;  // source < 0: version 0, logp from hist row 0 (own row only)
                if (src >= 0 && ok) {
                    double sm = 0.0;
                    for (int j = 0; j < G; ++j) sm += pw_w[r * G + j];
                    const double lp_old = grp_dbl(v_[ps][0]), lnfac = grp_dbl(v_[ps][1]), lnu = grp_dbl(v_[ps][2]), prior = grp_dbl(v_[ps][3]);
                    const double lp_new = fma(p.amp, sm, p.mean) + prior;
                    const int acc_flag = (lnfac + lp_new - lp_old > lnu) ? 1 : 0;
                    lp = acc_flag ? lp_new : lp_old;
                    flag = acc_flag ? 1.0 : 0.0;
                }
                rd_w[2 * r] = lp;
                rd_w[2 * r + 1] = flag;
            }
            __builtin_amdgcn_wave_barrier();
            const double lp_o = rd_w[4 * pl], fl_o = rd_w[4 * pl + 1], fl_p = want_partner ? rd_w[4 * pl + 3] : 0.0;
            __builtin_amdgcn_wave_barrier();                 // the next pass overwrites rd_w / pw_w
            acc_[ps] = (so_[ps] >= 0 && fl_o != 0.0) ? 1 : 0;
            if (valid) {
                if (k < d) {
                    sv_[ps] = grp_dbl(acc_[ps] ? v_[ps][0] : v_[ps][1]);
                    cv_[ps] = grp_dbl((sp_[ps] >= 0 && fl_p != 0.0) ? v_[ps][2] : v_[ps][3]);
                } else if (k == d) {
                    sv_[ps] = lp_o;
                }
            }
        }
        return ok;
    };

#ifdef ALABI_GROUP_PROF
    long long prof[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    // per-pass constants of this lane: is this member responsible for the slot's proposal, where its candidate goes
    bool mine_[NPM];
    int cand_off_[NPM];
#pragma unroll
    for (int ps = 0; ps < NPM; ++ps) {
        const int pp = ps * PPP + pwv * PPW + pl;
        mine_[ps] = (pp % G) == m;
        cand_off_[ps] = (g * p.QP + pp) * CW;
    }
    unsigned long long* cand_h = p.cand + (size_t)e * p.n0 * CW;       // candidates of the current half step
    const size_t cand_stride = (size_t)E * p.n0 * CW;
    for (int hh = 0; hh < 2 * p.K; ++hh, cand_h += cand_stride) {
        const int t = hh >> 1;
        const int cnt = half_count(hh);
        const unsigned long long* rs = rec_s + (hh & 3) * RW;
        // ---- phase 1: rebuild the rows, form the proposals, publish candidates and the A operands ----
        GRP_STAMP(c0);
        const int ok = rows_phase(rs, cnt, -1);
        GRP_STAMP(c1);
#pragma unroll
        for (int ps = 0; ps < NPM; ++ps) {
            const int ppbase = ps * PPP + pwv * PPW;
            if (ps >= npass || ppbase >= QPAD) continue;     // wave-uniform
            const int pp = ppbase + pl;
            const bool valid = pp < cnt;
            const int w = w_[ps], acc_o = acc_[ps];
            const double sv = sv_[ps], cv = cv_[ps];
            const double zz = valid ? grp_dbl(rs[4 * pp + 1]) : 0.0;
            const double qv = (valid && k < d) ? cv - (cv - sv) * zz : 0.0;
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
            if (valid && mine_[ps]) {
                // candidate of this proposal (everything the accept test needs but the kernel sum), for whoever reads the row later
                unsigned long long* cn = cand_h + cand_off_[ps];
                if (k < d) { grp_st(cn + k, grp_bits(qv)); grp_st(cn + d + k, grp_bits(sv)); }
                else if (k == d) { grp_st(cn + 2 * d, grp_bits(sv)); grp_st(cn + 2 * d + 3, grp_bits(seg == 0ull ? pr + p.prior_const : -INFINITY)); }
                else if (k == d + 1) { grp_st(cn + 2 * d + 1, rs[4 * pp + 2]); grp_st(cn + 2 * d + 2, rs[4 * pp + 3]); }
                // the chain: version t of the own walker (t = 0 is already there)
                if (t > 0 && k <= d + 1)
                    p.hist[((size_t)t * WT + w) * row + k] = (k <= d) ? grp_bits(sv) : (unsigned long long)acc_o;
            }
        }
        if (!ok) {
            ctl_s[0] = 1;
            __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grp_barrier();                                       // barrier A: A operands (and the abort word) are in LDS
        GRP_STAMP(c2);
        if (ctl_s[0]) return;
        // records: slot hh+2 from the registers, issue hh+3 (lands under this half step's kernel sums)
        if (wv == 0) {
#pragma unroll
            for (int j = 0; j < NRL; ++j)
                if (lane + j * 64 < RW) rec_s[((hh + 2) & 3) * RW + lane + j * 64] = pend[j];
#pragma unroll
            for (int j = 0; j < NRL; ++j) pend[j] = (lane + j * 64 < RW) ? rec_load(hh + 3, lane + j * 64) : 0ull;
        }
        // ---- phase 2: kernel sums of the QP proposals over this member's slice, on the matrix cores ----
        double a[Q][KS];
#pragma unroll
        for (int qt = 0; qt < Q; ++qt)
#pragma unroll
            for (int s = 0; s < KS; ++s) a[qt][s] = aop[(qt * 16 + lr) * KP + 4 * s + lk];
        v4f64 sum[Q];
#pragma unroll
        for (int qt = 0; qt < Q; ++qt) sum[qt] = v4f64{0.0, 0.0, 0.0, 0.0};
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
        GRP_STAMP(c3);
        grp_barrier();                                       // barrier B: the wave partials are in LDS
        // ---- phase 3 (wave 0): this member's partial sums, published like rows; the other waves go on ----
        if (wv == 0) {
            unsigned long long* part_h = p.part + (((size_t)hh * E + e) * NG + g) * (size_t)QPAD * G + m;
            for (int pp = lane; pp < cnt; pp += 64) {
                double x[ALABI_GRP_MAXW];
#pragma unroll
                for (int w = 0; w < ALABI_GRP_MAXW; ++w) x[w] = wsum[w * QPAD + pp];
                const double s = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                grp_st(part_h + (size_t)pp * G, grp_bits(s));
            }
        }
#ifdef ALABI_GROUP_PROF
        { GRP_STAMP(c4); prof[0] += c1 - c0; prof[1] += c2 - c1; prof[2] += c3 - c2; prof[3] += c4 - c3; prof[4] += 1;
          prof[5] += prof_setup - c0; prof[6] += prof_poll - prof_setup; }
#endif
    }
#ifdef ALABI_GROUP_PROF
    if (blockIdx.x == 5 && blockIdx.y == 0 && lane == 0 && (wv == NW - 1 || wv == 0))
        for (int i = 0; i < 7; ++i) g_group_prof[(wv == 0 ? 8 : 0) + i] = prof[i];
#endif
    // ---- tail: version K of every walker (nobody reads it inside the launch): the proposals of the last two half steps ----
    for (int hf = 2 * p.K - 2; hf < 2 * p.K; ++hf) {
        const int cnt = half_count(hf);
        const int ok = rows_phase(rec_s + (hf & 3) * RW, cnt, hf);
        if (!ok) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int ps = 0; ps < NPM; ++ps) {
            const int pp = ps * PPP + pwv * PPW + pl;
            if (ps < npass && pp < cnt && ok && mine_[ps] && k <= d + 1)
                p.hist[((size_t)p.K * WT + w_[ps]) * row + k] = (k <= d) ? grp_bits(sv_[ps]) : (unsigned long long)acc_[ps];
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
    int threads = 512;
    if (const char* env = getenv("ALABI_ENS_GROUP_THREADS")) { const int v = atoi(env); if (v == 256 || v == 512) threads = v; }
    const int ppw = (d + 2 <= 16) ? 4 : 2;                   // proposals per wave and pass of the row phase (16 / 32 lanes each)
    const int g_cap = 64 * ALABI_GRP_NPJ / (2 * ppw);        // partial words a lane gathers per pass: 2 ppw G / 64 <= NPJ
    double best_cost = 0.0;
    for (int Q = 1; Q <= 8; Q *= 2) {
        if (force_q && Q != force_q) continue;
        const int QP = 16 * Q;
        const int npm = (Q == 1) ? 1 : (Q == 2) ? 2 : 4;     // passes the kernel instantiation provides
        if ((QP + (threads / 64) * ppw - 1) / ((threads / 64) * ppw) > npm) continue;
        const int NG = (n0 + QP - 1) / QP;
        if (NG > avail) continue;
        int G = avail / NG;
        if (G > g_cap) G = g_cap;
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
    best.threads = threads;
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
    const int n0 = (e->W + 1) / 2;
    const size_t cand_words = (size_t)2 * e->chunk_cap * e->E * n0 * (2 * e->d + 4);
    if (e->cand_words < cand_words) {
        if (e->cand) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(e->cand); e->cand = nullptr; e->cand_words = 0; }
        ALABI_HIP_CHECK(hipMalloc(&e->cand, cand_words * sizeof(unsigned long long)));
        e->cand_words = cand_words;
    }
    hipLaunchKernelGGL(ens_group_fill_kernel, dim3(2048), dim3(256), 0, s, e->part, (size_t)2 * K * e->E * pl.NG * pl.G * pl.QP);
    hipLaunchKernelGGL(ens_group_fill_kernel, dim3(2048), dim3(256), 0, s, e->cand, (size_t)2 * K * e->E * n0 * (2 * e->d + 4));
    hipLaunchKernelGGL(ens_link_kernel, dim3(K, e->E), dim3(256), 0, s, e->draws, e->W, n0);
    if ((st = launch_ens_hist_prologue(e, coords, logp, K, false, s)) != ALABI_OK) return st;
    GroupArgs a{};
    a.hist = e->hist; a.part = e->part; a.cand = e->cand; a.err = e->err; a.packed = e->draws.packed; a.link = e->draws.link;
    a.consts = e->consts;
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
