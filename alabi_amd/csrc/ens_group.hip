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
//     proposal that produced it plus its G partials, adds them in a FIXED order (seg_allsum) and repeats the accept test
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
#define ALABI_GRP_NW 8                           // waves per workgroup (512 threads, two per SIMD)
// (Round 4, measured and not kept -- tools/experiments/ens_group_exp_table_2048.patch: a 2048-entry exp table (16 KB of LDS) with a
// degree-3 polynomial, one fused multiply-add fewer per kernel evaluation (9 instead of 10 fp64 instructions, 1.93 ulp): C4 4.85 vs
// 4.86 us per half step, C5 size 17.85 vs 17.97 -- the kernel sums are not bound by the count of vector instructions.)

struct GroupArgs {
    unsigned long long* hist;            // [(K+1)][E*W][d+2] version history: row 0 = state before the launch, rows 1..K written here
    unsigned long long* part;            // [2K][E][NG][QPAD][G] partial kernel sums, pre-filled with the sentinel
    unsigned long long* cand;            // [2K][E][n0][2d+4] candidates, pre-filled with the sentinel
    int* err;                            // [1] time-out flag
    const unsigned long long* packed;    // proposal records of the chunk: [K][E][W][4] (walker | partner << 32, z, (d-1) ln z, ln u')
    const unsigned long long* link;      // [K][E][W][2] where the two rows a proposal reads were produced (ens_link_kernel)
    const double* consts;                // [5][ALABI_MAX_DIM]: 1/length scale, lower, upper, prior mean, prior 1/std
    const double* Xa;                    // [4 KS][Npad] augmented centred training rows
    const double* centre;                // [d] centre of the scaled training inputs
    const double* alpha;                 // [Npad]
    int K, W, n0, d, Npad;
    int NG, QP;                          // groups, proposals per group and half step (16 Q)
    int tpm, ltw;                        // point tiles (of 16) per member; tiles per wave kept in LDS (the first RT of a wave's tiles are in registers)
    int xcd_map;                         // 1: members of a group share blockIdx % 8 (one XCD under round-robin placement; speed only)
    int spin_limit, has_prior;
    int poll_delay;                      // s_sleep(1) units (64 cycles) between barrier B and the first look at the fresh partial sums
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

// Value of lane ^ 16 (the other 16-lane row of a 32-lane slot): ds_swizzle in bit mode, and 0x1f / or 0 / xor 0x10
__device__ inline int swz16(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x401F); }
__device__ inline double swz16(double v) { return __hiloint2double(swz16(__double2hiint(v)), swz16(__double2loint(v))); }

// Sum of the G partial sums of a proposal, one per lane of an aligned segment of G = 8 or 16 lanes; result in every lane of the
// segment.  THE summation order of the members' partials: every reader of a row uses this tree, so all of them (and the tail
// pass that writes the chain) repeat the accept test on identical bits.
template <int G, int HALF>
__device__ inline double seg_allsum(double v) {
    v += dpp_move<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xf>(v);   // row_half_mirror
    if (G == 16) v += dpp_move<0x140, 0xf>(v);   // row_mirror
    if (G == 8 && HALF == 16) v += dpp_move<0x128, 0xf>(v);   // 8 partials in the first 8 of the half's 16 lanes (the others hold 0): row_ror:8
    return v;
}

// Where the two rows a proposal reads were produced, as word offsets into the candidate and partial-sum buffers (or -1 for
// version 0: the state before the launch, hist row 0).  Follows from the draws alone: version v of walker x is the outcome
// of the proposal x made in step v - 1, at position pos_of[v-1][x] of that step's lists.  One thread per list position.
__global__ void __launch_bounds__(256)
ens_link_kernel(DrawBuffers b, int W, int n0, int NG, int QPAD, int G, int CW) {
    const int t = blockIdx.x, e = blockIdx.y, E = gridDim.y;
    const size_t base = ((size_t)t * E + e) * W;
    const int g0 = e * W;
    for (int pos = threadIdx.x; pos < W; pos += 256) {
        const int wl = b.order[base + pos] - g0, cl = b.cw[base + pos] - g0;
        const int split = pos >= n0;
        int co[2], po[2];
        for (int which = 0; which < 2; ++which) {
            const int version = which ? t + split : t, xl = which ? cl : wl;
            co[which] = -1; po[which] = -1;
            if (version > 0) {
                const int pp = b.pos_of[((size_t)(version - 1) * E + e) * W + xl];
                const int set = pp >= n0, hp = 2 * (version - 1) + set, ph = pp - set * n0;
                co[which] = ((hp * E + e) * n0 + ph) * CW;
                po[which] = (((hp * E + e) * NG + ph / QPAD) * QPAD + ph % QPAD) * G;
            }
        }
        b.link[2 * (base + pos)] = (unsigned long long)(unsigned)co[0] | ((unsigned long long)(unsigned)co[1] << 32);
        b.link[2 * (base + pos) + 1] = (unsigned long long)(unsigned)po[0] | ((unsigned long long)(unsigned)po[1] << 32);
    }
}

// LDS layout (units of 8 bytes), the same formula on the host.
struct GroupLds {
    int etab, xb, al, aop, wsum, rec, ctl, total;
};
__host__ __device__ inline GroupLds group_lds(int KS, int QPAD, int lds_tiles) {
    const int S = 16 * lds_tiles;
    GroupLds L;
    int o = 0;
    L.etab = o; o += 256;
    L.xb = o; o += S * KS * 4;            // lds_tiles (= 8 waves x tiles per wave in LDS) x KS k-steps x 64 lanes
    L.al = o; o += S;
    L.aop = o; o += QPAD * KS * 4;
    L.wsum = o; o += ALABI_GRP_NW * QPAD;
    L.rec = o; o += 4 * QPAD * 6;         // ring of 4 half steps: 4 record words + 2 link words per proposal
    L.ctl = o; o += 2;
    L.total = o;
    return L;
}

#ifdef ALABI_GROUP_PROF
// phase stamps (s_memrealtime, 100 MHz) of one workgroup, accumulated over the half steps of the last launch:
// [0] rows phase [1] proposals + barrier A [2] kernel sums [3] barrier B + partial stores [4] half steps [5] rows phase: set-up
// [6] rows phase: polling -- for the wave that forms the first proposals (7); [8..] the same for wave 0 (the publisher)
__device__ long long g_group_prof[16];
extern "C" int alabi_debug_group_prof(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_group_prof), sizeof(long long) * 16);
}
#define GRP_STAMP(x) const long long x = (long long)__builtin_amdgcn_s_memrealtime()
#else
#define GRP_STAMP(x)
#endif

// KS k-steps of the augmented dot product (d + 2 <= 4 KS), Q proposal tiles per group.  Rows have d + 2 <= 16 words for
// KS <= 4 and <= 32 beyond: a proposal occupies LPR = 16 / 32 lanes in the row phase, and a group has G <= LPR / 2 members, so
// that the G partials of the own row and the G partials of the partner row of a proposal sit in the two halves of its lanes.
// G = LPR / 2, or (G8: rows of 17..32 words, eight members) G = 8 with the first RT point tiles of every wave held in REGISTERS
// for the whole launch: at d = 20, N = 10000 a member's slice of 79 tiles does not fit LDS (240 KB), and 16 members mean 64
// proposals per group whose row phase takes four passes; with 8 members it is 32 proposals in two passes.
template <int KS, int Q, bool G8, int RT, bool GENERIC>
__global__ void __launch_bounds__(512)
ens_group_kernel(GroupArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char grp_smem[];
    double* lds = reinterpret_cast<double*>(grp_smem);
    constexpr int QPAD = 16 * Q, KP = 4 * KS, NW = ALABI_GRP_NW;
    constexpr int LPR = (KS <= 4) ? 16 : 32, LSH = (KS <= 4) ? 4 : 5, HALF = LPR / 2, G = G8 ? 8 : HALF;
    constexpr int PPW = 64 / LPR, PPP = NW * PPW;            // proposals per wave / per pass over the workgroup
    constexpr int NPM = (QPAD + PPP - 1) / PPP;              // passes of the row phase
    constexpr int RW = 6 * QPAD;                             // ring words per half step
    const GroupLds L = group_lds(KS, QPAD, NW * p.ltw);
    double* etab = lds + L.etab;
    double* xb = lds + L.xb;
    double* al_s = lds + L.al;
    double* aop = lds + L.aop;
    double* wsum = lds + L.wsum;
    unsigned long long* rec_s = reinterpret_cast<unsigned long long*>(lds + L.rec);
    int* ctl_s = reinterpret_cast<int*>(lds + L.ctl);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int e = blockIdx.y, E = gridDim.y;
    const int NG = p.NG, d = p.d, row = d + 2, WT = p.W * E, CW = 2 * d + 4;
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
    // member m owns point tiles [m tpm, (m + 1) tpm) (clipped); wave wv of it the tiles [wv tpw, (wv + 1) tpw) of those: the first
    // RT of a wave's tiles live in its registers, the others in LDS (ltw per wave)
    const int tiles_all = p.Npad >> 4;
    const int tile0 = m * p.tpm;
    int ntile = tiles_all - tile0;
    if (ntile > p.tpm) ntile = p.tpm;
    if (ntile < 0) ntile = 0;
    const int tpw = (p.tpm + NW - 1) / NW;                   // (= RT + ltw when RT > 0)
    int wcount = ntile - wv * tpw;                           // tiles of this wave
    wcount = wcount < 0 ? 0 : (wcount > tpw ? tpw : wcount);
    const int nreg = wcount < RT ? wcount : RT, nlds = wcount - nreg;
    const int lr = lane & 15, lk = lane >> 4;
    double xr[RT > 0 ? RT : 1][KS], ar[RT > 0 ? RT : 1];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const size_t pt = ((size_t)tile0 + wv * tpw + r) * 16 + lr;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) xr[r][s2] = (r < nreg) ? p.Xa[(size_t)(4 * s2 + lk) * p.Npad + pt] : 0.0;
        ar[r] = (r < nreg) ? p.alpha[pt] : 0.0;
    }
    for (int lt = 0; lt < p.ltw; ++lt) {                     // this wave's LDS tiles (zeros beyond its count: alpha = 0 adds nothing)
        const bool have = lt < nlds;
        const size_t pt = ((size_t)tile0 + wv * tpw + RT + lt) * 16 + lr;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2)
            xb[((size_t)(wv * p.ltw + lt) * KS + s2) * 64 + lane] = have ? p.Xa[(size_t)(4 * s2 + lk) * p.Npad + pt] : 0.0;
        if (lk == 0) al_s[(wv * p.ltw + lt) * 16 + lr] = have ? p.alpha[pt] : 0.0;
    }
    if (tid < 256) etab[tid] = exp2((double)tid * 0.00390625);
    if (tid < 2) ctl_s[tid] = 0;
    for (int i = tid; i < NW * QPAD; i += 512) wsum[i] = 0.0;
    const int k = lane & (LPR - 1);                          // this lane's word of a row: k < d coordinate, k == d logp
    const int pl = lane >> LSH;                              // proposal slot within the wave
    const bool seg = k >= HALF;                              // which half of the slot's lanes: own row's / partner row's partials
    const int pwv = NW - 1 - wv;                             // the LAST wave takes the first proposals: wave 0 publishes the partials
    const double il_r = (k < d) ? p.consts[k] : 0.0, lo_r = (k < d) ? p.consts[ALABI_MAX_DIM + k] : 0.0;
    const double hi_r = (k < d) ? p.consts[2 * ALABI_MAX_DIM + k] : 0.0;
    const double pm_r = (k < d) ? p.consts[3 * ALABI_MAX_DIM + k] : 0.0, pi_r = (k < d) ? p.consts[4 * ALABI_MAX_DIM + k] : 0.0;
    const double c_r = (k < d) ? p.centre[k] : 0.0;
    const double SC = GENERIC ? 1.0 : ALABI_EXP2S256_SCALE;

    // ---- proposal records: ring of 4 half steps in LDS, fetched three half steps ahead (plain loads: written before the launch).
    // Wave 0 owns the ring: its other memory traffic is the partial stores, so the counted wait in front of the LDS write
    // never sits behind a candidate store or a poll of the row phase.
    const int n1 = p.W - p.n0;
    auto half_count = [&](int hh) { const int nh = (hh & 1) ? n1 : p.n0; int c = nh - g * p.QP; c = c < 0 ? 0 : c; return c > p.QP ? p.QP : c; };
    auto rec_load = [&](int hh, int i) -> unsigned long long {       // word i of the group's record block of half step hh
        if (hh >= 2 * p.K) return 0xFFFFFFFFFFFFFFFFull;
        const int c = half_count(hh);
        const size_t pos0 = ((size_t)(hh >> 1) * E + e) * p.W + ((hh & 1) ? p.n0 : 0) + (size_t)g * p.QP;
        if (i < 4 * QPAD) return (i < 4 * c) ? p.packed[4 * pos0 + i] : 0xFFFFFFFFFFFFFFFFull;
        return (i - 4 * QPAD < 2 * c) ? p.link[2 * pos0 + (i - 4 * QPAD)] : 0xFFFFFFFFFFFFFFFFull;
    };
    constexpr int NRL = (RW + 63) / 64;
    unsigned long long pend[NRL];
    if (wv == 0) {
        for (int hh = 0; hh < 2; ++hh)
            for (int i = lane; i < RW; i += 64) rec_s[(hh & 3) * RW + i] = rec_load(hh, i);
#pragma unroll
        for (int jj = 0; jj < NRL; ++jj) pend[jj] = (lane + jj * 64 < RW) ? rec_load(2, lane + jj * 64) : 0ull;
    }
    __syncthreads();


    // per-pass constants of this lane: its proposal, is this member responsible for it, where its candidate goes
    bool mine_[NPM];
    int cand_off_[NPM];
#pragma unroll
    for (int ps = 0; ps < NPM; ++ps) {
        const int pp = ps * PPP + pwv * PPW + pl;
        // member (pp + PPW (pp / G)) mod G publishes proposal pp: the proposals of one member sit in slots of DIFFERENT waves
        // (with pp mod G they were all in one wave's slot, and that wave's candidate / chain stores set the pace of the row
        // phase: 4.8 -> 2.9 us of accept tests + proposals at C5 size)
        mine_[ps] = ((pp + PPW * (pp / G)) % G) == m;
        cand_off_[ps] = (g * p.QP + pp) * CW;
    }

    // Rows of this wave's proposals in ALL passes of a half step, rebuilt from candidates + partials (wave-level: every lane
    // of the wave takes part).  Pass ps, slot pl is proposal ps PPP + pwv PPW + pl of the record block `rs`.  Lane (slot,
    // word k) gets word k of the own row in sv_[ps] (k < d coordinate, k == d logp), coordinate k of the partner row in
    // cv_[ps], and acc_[ps] = was the own row's producing proposal accepted.  An offset < 0 means hist row 0.  The lanes of
    // a slot also hold, one each, the G partials of the own row (lower half) and of the partner row (upper half) and every
    // lane of a half loads that row's four accept-test scalars, so the test is repeated in registers (seg_allsum), no LDS.
    // All first looks of all passes are issued before any is examined: one memory round trip when everything is there.
    // tail_hf >= 0: the rows the proposals of half step tail_hf PRODUCED are wanted (own rows only).
#ifdef ALABI_GROUP_PROF
    long long prof_poll = 0, prof_setup = 0;
#endif
    // body(ps, pp, valid, w, sv, cv, acc_o) is called once per pass that holds proposals of this wave, in pass order
    auto rows_phase = [&](const unsigned long long* rs, int cnt, int tail_hf, auto&& body) -> int {
        if (pwv * PPW >= cnt) return 1;                      // wave-uniform: none of this wave's slots holds a proposal
        const bool want_partner = tail_hf < 0;
        // words of a pass: bits 0..3 own q_k, own old_k, partner q_k, partner old_k (k < d); bit 4 this lane's partial;
        // bits 5..8 the accept-test scalars (old logp, (d-1) ln z, ln u', prior term) of this half's row
        unsigned long long va_[NPM][4], vc_[NPM][4], vb_[NPM];
        const unsigned long long* pa_[NPM][4];
        const unsigned long long *pb_[NPM], *pc_[NPM];
        unsigned want_[NPM];
        int w_[NPM];
#pragma unroll
        for (int ps = 0; ps < NPM; ++ps) {
            const int pp = ps * PPP + pwv * PPW + pl;
            const bool valid = pp < cnt;
            unsigned want = 0u;
            w_[ps] = 0;
            vb_[ps] = 0ull; pb_[ps] = nullptr; pc_[ps] = nullptr;
#pragma unroll
            for (int i = 0; i < 4; ++i) { pa_[ps][i] = nullptr; va_[ps][i] = 0ull; vc_[ps][i] = 0ull; }
            if (valid) {
                const unsigned long long ids = rs[4 * pp];
                const int w = (int)(unsigned)(ids & 0xffffffffull), cw = (int)(unsigned)(ids >> 32);
                w_[ps] = w;
                int co, cp, po, pq;
                if (tail_hf >= 0) {
                    co = ((tail_hf * E + e) * p.n0 + g * p.QP + pp) * CW; cp = -1;
                    po = (((tail_hf * E + e) * NG + g) * QPAD + pp) * G; pq = -1;
                } else {
                    const unsigned long long l0 = rs[4 * QPAD + 2 * pp], l1 = rs[4 * QPAD + 2 * pp + 1];
                    co = (int)(unsigned)(l0 & 0xffffffffull); cp = (int)(unsigned)(l0 >> 32);
                    po = (int)(unsigned)(l1 & 0xffffffffull); pq = (int)(unsigned)(l1 >> 32);
                }
                if (k < d) {
                    if (co >= 0) { pa_[ps][0] = p.cand + co + k; pa_[ps][1] = p.cand + co + d + k; want |= 3u; }
                    else { pa_[ps][1] = p.hist + (size_t)w * row + k; want |= 2u; }
                    if (want_partner) {
                        if (cp >= 0) { pa_[ps][2] = p.cand + cp + k; pa_[ps][3] = p.cand + cp + d + k; want |= 12u; }
                        else { pa_[ps][3] = p.hist + (size_t)cw * row + k; want |= 8u; }
                    }
                }
                const int my_c = seg ? cp : co, my_p = seg ? pq : po;
                if (my_c >= 0 && (!seg || want_partner)) {
                    pc_[ps] = p.cand + my_c + 2 * d;
                    want |= 0x1E0u;
                    if ((k & (HALF - 1)) < G) { pb_[ps] = p.part + my_p + (k & (HALF - 1)); want |= 0x10u; }
                } else if (!seg) {
                    pc_[ps] = p.hist + (size_t)w * row + d;  // version 0: the logp of hist row 0
                    want |= 0x20u;
                }
            }
            want_[ps] = want;
        }
        int spins = 0, ok = 1;
#ifdef ALABI_GROUP_PROF
        prof_setup = (long long)__builtin_amdgcn_s_memrealtime();
#endif
        // (a) one look at the words that were published at least a half step ago (candidates, hist row 0) ...
#pragma unroll
        for (int ps = 0; ps < NPM; ++ps) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (want_[ps] & (32u << i)) vc_[ps][i] = grp_ld(pc_[ps] + i);
                if (want_[ps] & (1u << i)) va_[ps][i] = grp_ld(pa_[ps][i]);
            }
        }
        // (b) ... and looks at the FRESH words, the partial sums of the half step just before.  A look costs a memory round
        // trip (0.6-0.7 us), so the first one is timed for when the partials are expected to be visible (`poll_delay`): a
        // look that leaves too early comes back empty and the next one a whole round trip later.
        {
            for (int z = 0; z < p.poll_delay; ++z) __builtin_amdgcn_s_sleep(1);
            unsigned bmiss = 0u;
#pragma unroll
            for (int ps = 0; ps < NPM; ++ps) if (want_[ps] & 16u) bmiss |= 1u << ps;
            while (bmiss) {
                unsigned long long tb[NPM];
#pragma unroll
                for (int ps = 0; ps < NPM; ++ps) { tb[ps] = ALABI_GRP_EMPTY; if (bmiss & (1u << ps)) tb[ps] = grp_ld(pb_[ps]); }
#pragma unroll
                for (int ps = 0; ps < NPM; ++ps)
                    if ((bmiss & (1u << ps)) && tb[ps] != ALABI_GRP_EMPTY) { vb_[ps] = tb[ps]; bmiss &= ~(1u << ps); }
                if (bmiss && (++spins > p.spin_limit ||
                              ((spins & 63) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))) {
                    ok = 0;
                    break;
                }
            }
        }
        // (c) the old words again, should one of them not have been there at the first look
        while (ok) {
            bool all = true;
#pragma unroll
            for (int ps = 0; ps < NPM; ++ps)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if ((want_[ps] & (32u << i)) && vc_[ps][i] == ALABI_GRP_EMPTY) all = false;
                    if ((want_[ps] & (1u << i)) && va_[ps][i] == ALABI_GRP_EMPTY) all = false;
                }
            if (all) break;
#pragma unroll
            for (int ps = 0; ps < NPM; ++ps)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if ((want_[ps] & (32u << i)) && vc_[ps][i] == ALABI_GRP_EMPTY) vc_[ps][i] = grp_ld(pc_[ps] + i);
                    if ((want_[ps] & (1u << i)) && va_[ps][i] == ALABI_GRP_EMPTY) va_[ps][i] = grp_ld(pa_[ps][i]);
                }
            if (++spins > p.spin_limit ||
                ((spins & 15) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                ok = 0;
                break;
            }
        }
#ifdef ALABI_GROUP_PROF
        prof_poll = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int ps = 0; ps < NPM; ++ps) {
            if (ps * PPP + pwv * PPW >= cnt) continue;       // wave-uniform
            const int pp = ps * PPP + pwv * PPW + pl;
            // accept test of the proposal that produced this half's row, in every lane of the half
            const double sm = seg_allsum<G, HALF>(grp_dbl(vb_[ps]));
            const double lp_old = grp_dbl(vc_[ps][0]);
            int accf = 0;
            double lp_sel = lp_old;
            if (want_[ps] & 0x40u) {                          // this half's row has a producing proposal inside the launch
                const double lp_new = fma(p.amp, sm, p.mean) + grp_dbl(vc_[ps][3]);
                accf = (grp_dbl(vc_[ps][1]) + lp_new - lp_old > grp_dbl(vc_[ps][2])) ? 1 : 0;
                lp_sel = accf ? lp_new : lp_old;
            }
            // the other half's outcome
            int o_acc;
            double o_lp;
            if (LPR == 16) {
                o_acc = __builtin_amdgcn_update_dpp(0, accf, 0x128, 0xf, 0xf, true);   // row_ror:8
                o_lp = dpp_move<0x128, 0xf>(lp_sel);
            } else {
                o_acc = swz16(accf);
                o_lp = swz16(lp_sel);
            }
            const int flag_o = seg ? o_acc : accf, flag_p = seg ? accf : o_acc;
            const double lp_o = seg ? o_lp : lp_sel;
            double sv = 0.0, cv = 0.0;
            if (k < d) {
                sv = grp_dbl(flag_o ? va_[ps][0] : va_[ps][1]);
                cv = grp_dbl(flag_p ? va_[ps][2] : va_[ps][3]);
            } else if (k == d) {
                sv = lp_o;
            }
            body(ps, pp, pp < cnt, w_[ps], sv, cv, flag_o);
        }
        return ok;
    };

#ifdef ALABI_GROUP_PROF
    long long prof[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    unsigned long long* cand_h = p.cand + (size_t)e * p.n0 * CW;       // candidates of the current half step
    const size_t cand_stride = (size_t)E * p.n0 * CW;
    for (int hh = 0; hh < 2 * p.K; ++hh, cand_h += cand_stride) {
        const int t = hh >> 1;
        const int cnt = half_count(hh);
        const unsigned long long* rs = rec_s + (hh & 3) * RW;
        unsigned long long* hist_t = p.hist + (size_t)t * WT * row;
        // ---- phase 1: rebuild the rows, form the proposals, publish candidates and the A operands ----
        GRP_STAMP(c0);
        // slots of this wave beyond the half step's proposals but inside the tile: zero A operands
#pragma unroll
        for (int ps = 0; ps < NPM; ++ps) {
            const int ppbase = ps * PPP + pwv * PPW;
            if (ppbase < QPAD && ppbase + pl >= cnt && k < KP) aop[(ppbase + pl) * KP + k] = 0.0;
        }
        const int ok = rows_phase(rs, cnt, -1, [&](int ps, int pp, bool valid, int w, double sv, double cv, int acc_o) {
            const double zz = valid ? grp_dbl(rs[4 * pp + 1]) : 0.0;
            const double qv = (valid && k < d) ? cv - (cv - sv) * zz : 0.0;
            // lanes of one proposal: k = 0 .. LPR-1.  In-box test, |q - c|^2 and the normal-prior term by segmented reductions.
            const int out = (valid && k < d) ? !((qv > lo_r) && (qv < hi_r)) : 0;
            const unsigned long long om = __ballot(out);
            const unsigned long long seg_out = (LPR == 16) ? ((om >> (lane & 48)) & 0xffffull) : ((om >> (lane & 32)) & 0xffffffffull);
            const double qs = (valid && k < d) ? qv * il_r - c_r : 0.0;
            double qq = row16_allsum(qs * qs);
            double pr = 0.0;
            if (p.has_prior) {
                double tt = (valid && k < d) ? (qv - pm_r) * pi_r : 0.0;
                pr = row16_allsum(-0.5 * tt * tt);
            }
            if (LPR == 32) {
                qq += swz16(qq);
                if (p.has_prior) pr += swz16(pr);
            }
            if (valid && k < KP)
                aop[pp * KP + k] = (k < d) ? qs * SC : (k == d) ? SC : (k == d + 1) ? -0.5 * qq * SC : 0.0;
            if (valid && mine_[ps]) {
                // candidate of this proposal (everything the accept test needs but the kernel sum), for whoever reads the row later
                unsigned long long* cn = cand_h + cand_off_[ps];
                if (k < d) { grp_st(cn + k, grp_bits(qv)); grp_st(cn + d + k, grp_bits(sv)); }
                else if (k == d) { grp_st(cn + 2 * d, grp_bits(sv)); grp_st(cn + 2 * d + 3, grp_bits(seg_out == 0ull ? pr + p.prior_const : -INFINITY)); }
                else if (k == d + 1) { grp_st(cn + 2 * d + 1, rs[4 * pp + 2]); grp_st(cn + 2 * d + 2, rs[4 * pp + 3]); }
                // the chain: version t of the own walker (t = 0 is already there)
                if (t > 0 && k <= d + 1)
                    hist_t[(size_t)w * row + k] = (k <= d) ? grp_bits(sv) : (unsigned long long)acc_o;
            }
        });
        GRP_STAMP(c1);
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
            for (int jj = 0; jj < NRL; ++jj)
                if (lane + jj * 64 < RW) rec_s[((hh + 2) & 3) * RW + lane + jj * 64] = pend[jj];
#pragma unroll
            for (int jj = 0; jj < NRL; ++jj) pend[jj] = (lane + jj * 64 < RW) ? rec_load(hh + 3, lane + jj * 64) : 0ull;
        }
        // ---- phase 2: kernel sums of the QP proposals over this member's slice, on the matrix cores ----
        // (waves w and w + 4 share a SIMD; w finishes its sums 15-20 % before w + 4.  Raising the younger wave's s_setprio for
        // the phase was measured and changes nothing: 5.06 / 24.5 us per half step at C4 / C5-sized either way; neither does
        // giving the older wave 55 % of the pair's tile products: the sums then end together, 15.0 / 14.0 instead of 15.6 / 12.8 us
        // at C5 size, but the half step gains 0.1 us -- the SIMD's fp64 pipe is simply busy for that long.)
        double a[Q][KS];
#pragma unroll
        for (int qt = 0; qt < Q; ++qt)
#pragma unroll
            for (int s = 0; s < KS; ++s) a[qt][s] = aop[(qt * 16 + lr) * KP + 4 * s + lk];
        v4f64 sum[Q];
#pragma unroll
        for (int qt = 0; qt < Q; ++qt) sum[qt] = v4f64{0.0, 0.0, 0.0, 0.0};
        auto tile_products = [&](const double* bop, double al) {
#pragma unroll
            for (int qt = 0; qt < Q; ++qt) {
                v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qt][s2], bop[s2], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {                // C/D layout: row (proposal) lk + 4 i, column (point) lr
                    const double f = GENERIC ? radial<true>(fmax(-2.0 * acc[i], 0.0), p.kf) : exp2s_tab256(acc[i], etab);
                    sum[qt][i] = fma(al, f, sum[qt][i]);
                }
            }
        };
#pragma unroll
        for (int r = 0; r < RT; ++r)
            tile_products(xr[r], ar[r]);                     // (tiles beyond nreg hold x = 0, alpha = 0: fma(0, 1, sum) = sum exactly --
                                                             // unguarded, the RT tiles are straight-line code the compiler can interleave)
        for (int lt = 0; lt < nlds; ++lt) {
            const double* xbt = xb + (size_t)(wv * p.ltw + lt) * KS * 64 + lane;
            double bop[KS];
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) bop[s2] = xbt[s2 * 64];
            tile_products(bop, al_s[(wv * p.ltw + lt) * 16 + lr]);
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
        // (Round 4, measured and not kept -- tools/experiments/ens_group_publisher_last_arriver.patch: the wave that finishes its sums
        // LAST, found with an arrival counter in LDS, publishing in front of barrier B instead of wave 0 behind it: C4 4.85 -> 5.04 us
        // per half step, C5 size 17.87 -> 18.05, with the poll delay re-tuned 4.96.  The last arriver is one of the waves w + 4, which at
        // C4 are the waves that run the rows phase: the publish then sits in front of their own set-up and first look, while wave 0
        // has nothing else to do there.)
        if (wv == 0) {
            unsigned long long* part_h = p.part + (((size_t)hh * E + e) * NG + g) * (size_t)QPAD * G + m;
            for (int pp = lane; pp < cnt; pp += 64) {
                double x[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) x[w] = wsum[w * QPAD + pp];
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
        unsigned long long* hist_K = p.hist + (size_t)p.K * WT * row;
        const int ok = rows_phase(rec_s + (hf & 3) * RW, cnt, hf, [&](int ps, int pp, bool valid, int w, double sv, double cv, int acc_o) {
            if (valid && mine_[ps] && k <= d + 1) hist_K[(size_t)w * row + k] = (k <= d) ? grp_bits(sv) : (unsigned long long)acc_o;
        });
        if (!ok) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side: the blocking for an ensemble, buffers, launch.
struct GroupPlan {
    int ok, KS, Q, QP, G, NG, RT, tpm, ltw;
    size_t lds_bytes;
};
#define ALABI_GRP_RT 5                           // point tiles per wave held in registers by the G8 instantiations

#define ALABI_GRP_MAX_DEV 64                     // per-device caches (a process may touch several devices)
static int group_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 0 && dev < ALABI_GRP_MAX_DEV ? dev : 0;
}
static int group_n_cu() {
    static int n_cu[ALABI_GRP_MAX_DEV] = {0};
    const int dev = group_device();
    if (n_cu[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu[dev] = v;
    }
    return n_cu[dev];
}

// G members per group: 8 for d <= 14 (rows of <= 16 words); for wider rows 8 with four point tiles per wave in registers when the
// rest of the slice then fits LDS, else 16; as many groups as the CUs allow, each with the smallest power-of-two number Q of
// 16-proposal tiles that covers its share of a half step.
static GroupPlan group_plan(const alabi_ens* e) {
    const alabi_gp* gp = e->gp;
    const int d = e->d;
    GroupPlan none{};
    if (d + 2 > 32 || e->ymap != 0 || e->W < 2) return none;
    const int n_cu = group_n_cu();
    if (e->E > n_cu) return none;
    const int KS = (d + 2 + 3) / 4;
    const int n0 = (e->W + 1) / 2;
    const int tiles = gp->Npad / 16;
    const char* g16 = getenv("ALABI_ENS_GROUP_G16");          // tests: the 16-member blocking where 8 members would be chosen
    for (int attempt = 0; attempt < 2; ++attempt) {
        GroupPlan pl{};
        const bool wide = KS > 4;
        if (attempt == 0 && g16 && g16[0] == '1') continue;  // attempt 0: the instantiation with register-resident point tiles
        const int G = (wide && attempt == 1) ? 16 : 8;
        const int RT = (attempt == 0) ? ALABI_GRP_RT : 0;
        const int ng_max = n_cu / e->E / G;
        if (ng_max < 1) continue;
        int Q = 1;
        while (Q < 8 && (n0 + 16 * Q - 1) / (16 * Q) > ng_max) Q *= 2;
        if ((n0 + 16 * Q - 1) / (16 * Q) > ng_max) continue;
        if (wide && Q > 4) continue;                         // passes of the row phase: 16 Q proposals / (8 waves x 2) <= 4
        if (!wide && RT > 0 && Q > 2) continue;              // (register budget of the narrow-row instantiations with many proposal tiles)
        pl.KS = KS; pl.Q = Q; pl.QP = 16 * Q; pl.G = G; pl.NG = (n0 + pl.QP - 1) / pl.QP; pl.RT = RT;
        pl.tpm = (tiles + G - 1) / G;
        const int tpw = (pl.tpm + ALABI_GRP_NW - 1) / ALABI_GRP_NW;
        pl.ltw = tpw > RT ? tpw - RT : 0;
        pl.lds_bytes = (size_t)group_lds(KS, pl.QP, ALABI_GRP_NW * pl.ltw).total * 8;
        pl.ok = pl.lds_bytes <= 160 * 1024 - 512;
        if (pl.ok) return pl;
    }
    return none;
}

bool ens_group_fits(const alabi_ens* e) { return e->hist && e->err && group_plan(e).ok; }

// The hand-off buffers of the group kernel (partial sums, candidate rows) for a full chunk: allocated on first use, ~0.3 GB at C4
// and ~0.9 GB at the C5 size.  false = the blocking does not fit the 32-bit word offsets of the link records or the memory is not
// there -- alabi_ens_run then takes another path (ens_stream_kernel or one launch per half step) instead of failing.
bool ens_group_buffers(alabi_ens* e, hipStream_t s) {
    const GroupPlan pl = group_plan(e);
    if (!pl.ok) return false;
    const int n0 = (e->W + 1) / 2, CW = 2 * e->d + 4;
    const size_t part_per_half = (size_t)e->E * pl.NG * pl.QP * pl.G, cand_per_half = (size_t)e->E * n0 * CW;
    if (2 * (size_t)e->chunk_cap * part_per_half > 0x7fffffffull || 2 * (size_t)e->chunk_cap * cand_per_half > 0x7fffffffull) return false;
    auto ensure = [&](unsigned long long** p, size_t* have, size_t need) {
        if (*have >= need) return true;
        if (*p) { (void)hipStreamSynchronize(s); (void)hipFree(*p); *p = nullptr; *have = 0; }
        if (hipMalloc(p, need * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
        *have = need;
        return true;
    };
    return ensure(&e->part, &e->part_words, 2 * e->chunk_cap * part_per_half) && ensure(&e->cand, &e->cand_words, 2 * e->chunk_cap * cand_per_half);
}

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
    if constexpr (KS > 4 && Q > 4) {
        return ALABI_BAD_ARGUMENT;
    } else {
        constexpr bool HAS_RT = (KS > 4) || (Q <= 2);        // instantiations with register-resident tiles
        const void* kern = (pl.RT > 0 && HAS_RT) ? reinterpret_cast<const void*>(ens_group_kernel<KS, Q, (KS > 4), (HAS_RT ? ALABI_GRP_RT : 0), GENERIC>)
                                                 : reinterpret_cast<const void*>(ens_group_kernel<KS, Q, false, 0, GENERIC>);
        static bool attr_set[ALABI_GRP_MAX_DEV][2] = {};    // per instantiation of this function: per device, per variant
        const int var = (pl.RT > 0 && HAS_RT) ? 1 : 0, dev = group_device();
        if (!attr_set[dev][var]) {
            ALABI_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev][var] = true;
        }
        GroupArgs args = a;
        void* params[] = {&args};
        ALABI_HIP_CHECK(hipLaunchKernel(kern, dim3(pl.NG * pl.G, E), dim3(512), params, pl.lds_bytes, s));
        return ALABI_OK;
    }
}

int launch_ens_group(alabi_ens* e, double* coords, double* logp, int K, int thin_by, double* chain, double* chain_logp,
                     long long* n_accept, hipStream_t s) {
    alabi_gp* gp = e->gp;
    const GroupPlan pl = group_plan(e);
    if (!pl.ok) return ALABI_BAD_ARGUMENT;
    int st = ensure_xa(gp, s);
    if (st != ALABI_OK) return st;
    const int n0 = (e->W + 1) / 2, CW = 2 * e->d + 4;
    const size_t part_per_half = (size_t)e->E * pl.NG * pl.QP * pl.G, cand_per_half = (size_t)e->E * n0 * CW;
    if (!ens_group_buffers(e, s)) return ALABI_BAD_ARGUMENT;   // (alabi_ens_run asks before it chooses this kernel)
    hipLaunchKernelGGL(ens_group_fill_kernel, dim3(2048), dim3(256), 0, s, e->part, 2 * K * part_per_half);
    hipLaunchKernelGGL(ens_group_fill_kernel, dim3(2048), dim3(256), 0, s, e->cand, 2 * K * cand_per_half);
    hipLaunchKernelGGL(ens_link_kernel, dim3(K, e->E), dim3(256), 0, s, e->draws, e->W, n0, pl.NG, pl.QP, pl.G, CW);
    if ((st = launch_ens_hist_prologue(e, coords, logp, K, false, s)) != ALABI_OK) return st;
    GroupArgs a{};
    a.hist = e->hist; a.part = e->part; a.cand = e->cand; a.err = e->err; a.packed = e->draws.packed; a.link = e->draws.link;
    a.consts = e->consts;
    a.Xa = gp->Xa; a.centre = gp->xa_centre; a.alpha = gp->alpha;
    a.K = K; a.W = e->W; a.n0 = n0; a.d = e->d; a.Npad = gp->Npad;
    a.NG = pl.NG; a.QP = pl.QP; a.tpm = pl.tpm; a.ltw = pl.ltw;
    a.xcd_map = 1;                                           // measured at C4: 6.23 us per half step with it, 6.58 without (first version)
    a.spin_limit = 1 << 20;
    if (const char* env = getenv("ALABI_ENS_SPIN_LIMIT")) { const int v = atoi(env); if (v > 0) a.spin_limit = v - 1; }   // tests: force a time-out (1: the first miss)
    a.has_prior = e->has_prior; a.prior_const = e->prior_const;
    a.poll_delay = 6;                                        // ~0.16 us: measured optimum at C4 (5.13 vs 5.21 us per half step at 0, 5.5 at 20)
    a.amp = e->lp_scale * exp(gp->log_amp); a.mean = fma(e->lp_scale, gp->mean, e->lp_shift); a.kf = gp->kf;
    e->last_path = 3;
    { const int rec[8] = {pl.Q, pl.G, pl.NG, pl.RT, pl.tpm, pl.ltw, pl.KS, (int)pl.lds_bytes};
      for (int i = 0; i < 8; ++i) e->group_plan[i] = rec[i]; }
    ALABI_GROUP_DISPATCH_KS(pl.KS, ALABI_GROUP_DISPATCH_Q(pl.Q, ALABI_DISPATCH_KERNEL(gp->kf.type,
        { st = group_launch<KS, Q, GENERIC>(a, pl, e->E, s); })));
    if (st != ALABI_OK) return st;
    return launch_ens_hist_epilogue(e, coords, logp, K, thin_by, chain, chain_logp, n_accept, s);
}

}  // namespace alabi
