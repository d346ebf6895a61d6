// Batched GP prediction for gfx950: mu* = k(X*,X) alpha + m and var* = k** - |L^-1 k*|^2.
//
// Replaces george's gp.predict(y, X*, return_var=) reached from the reference at
// alabi/core.py:85, :95 (cached likelihood), :1441, :1486 (surrogate_log_likelihood),
// :1601 (acquisition), :1812/:1828 (bookkeeping).  K* is never materialised in HBM.
//
// mean, small M  : one workgroup per query, lanes along the training points (the same device
//                  function the ensemble sampler uses for a walker's log-probability).
// mean, large M  : 64 queries per workgroup, lanes along queries (no cross-lane reduction);
//                  256 training points at a time staged in LDS and broadcast to the lanes.
//                  Bound by the fp64 vector/transcendental rate: N (2d+2) flops + N exp per
//                  query, ~8(d+1) bytes of HBM per query.
// variance       : per tile of 64 queries a blocked forward substitution V = L^-1 K*^T.  For
//                  each 64-row block kb: the K* block is evaluated on the fly, the
//                  off-diagonal part  C = K*_kb - sum_{j<kb} L[kb,j] V_j  runs on the fp64
//                  matrix cores (v_mfma_f64_16x16x4_f64, L block and V block staged in LDS),
//                  the 64x64 diagonal solve runs as 16 columns per wave with 4 lanes per
//                  column exchanging the solved entry by wave shuffle.  V_j tiles live in a
//                  per-workgroup HBM workspace (they do not fit in 160 KB of LDS) and are
//                  re-read through L2.  N^2 flops per query (MFMA bound), L streamed once per
//                  64-query tile.
#include <cstdlib>
#include "gp_device.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int D>
__global__ void __launch_bounds__(256)
predict_mean_rowwise_kernel(const double* __restrict__ Xt, const double* __restrict__ alpha, int Npad,
                            const double* __restrict__ Xs, int d, DimVec inv_len, double amp, double mean,
                            KernelFn kf, double* __restrict__ mu) {
    __shared__ double q[ALABI_MAX_DIM];
    __shared__ double scratch[16];
    const long long m = blockIdx.x;
    if (threadIdx.x < D) q[threadIdx.x] = (threadIdx.x < d) ? Xs[m * d + threadIdx.x] * inv_len.v[threadIdx.x] : 0.0;
    __syncthreads();
    double s = gp_kernel_dot_block<D>(Xt, alpha, Npad, q, scratch, kf);
    if (threadIdx.x == 0) mu[m] = fma(amp, s, mean);
}

template <int D, bool GENERIC>
__global__ void __launch_bounds__(256)
predict_mean_tile_kernel(const double* __restrict__ Xt, const double* __restrict__ alpha, int Npad,
                         const double* __restrict__ Xs, int d, long long M, DimVec inv_len, double amp,
                         double mean, KernelFn kf, double* __restrict__ mu, double* __restrict__ mu_part, long long mu_stride) {
    __shared__ double xt[D][256];
    __shared__ double al[256];
    __shared__ double part[4][64];
    __shared__ double etab[32];                                  // 2^(j/32) for exp_tab32 (first barrier of the loop orders it)
    const int tid = threadIdx.x, c = tid & 63, w = tid >> 6;
    if (tid < 32) etab[tid] = exp2((double)tid * 0.03125);
    const long long m = (long long)blockIdx.x * 64 + c;
    double q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) q[k] = (m < M && k < d) ? Xs[m * d + k] * inv_len.v[k] : 0.0;
    double acc = 0.0;
    // gridDim.y > 1: whole 256-point stages of the training set are dealt to gridDim.y workgroups per query tile (too few
    // tiles to fill the chip otherwise); mean_combine_kernel adds the parts in order
    const int stages = (Npad + 255) / 256;
    const int st_lo = (int)((long long)stages * blockIdx.y / gridDim.y), st_hi = (int)((long long)stages * (blockIdx.y + 1) / gridDim.y);
    for (int n0 = st_lo * 256; n0 < st_hi * 256; n0 += 256) {
        __syncthreads();
        const int n = n0 + tid;
#pragma unroll
        for (int k = 0; k < D; ++k) xt[k][tid] = (n < Npad) ? Xt[(size_t)k * Npad + n] : 0.0;
        al[tid] = (n < Npad) ? alpha[n] : 0.0;
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < 64; ++j) {
            const int nn = w * 64 + j;
            double r2 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double df = xt[k][nn] - q[k];
                r2 = fma(df, df, r2);
            }
            acc = fma(al[nn], GENERIC ? radial<true>(r2, kf) : exp_tab32(-0.5 * r2, etab), acc);
        }
    }
    part[w][c] = acc;
    __syncthreads();
    if (tid < 64 && m < M) {
        const double t = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        if (gridDim.y == 1) mu[m] = fma(amp, t, mean);
        else mu_part[(size_t)blockIdx.y * mu_stride + m] = t;
    }
}

__global__ void __launch_bounds__(256)
mean_combine_kernel(const double* __restrict__ mu_part, int nsplit, long long mu_stride, long long M, double amp, double mean,
                    double* __restrict__ mu) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    double t = 0.0;
    for (int k = 0; k < nsplit; ++k) t += mu_part[(size_t)k * mu_stride + m];
    mu[m] = fma(amp, t, mean);
}

// IDENT = true turns the same blocked forward substitution into a triangular inversion: the right-hand side of
// tile t is columns 64t..64t+63 of the identity, the V tiles are the OUTPUT (ws[t] = L^-1[:, 64t:64t+64] as
// [Npad][64], rows above block t are never written nor read), and nothing else is produced.
template <int D, bool IDENT = false>
__global__ void __launch_bounds__(256, 2)
predict_var_kernel(const double* __restrict__ L, const double* __restrict__ dinv, const double* __restrict__ Xt,
                   const double* __restrict__ alpha, int N, int Npad, const double* __restrict__ Xs, int d, long long M, DimVec inv_len,
                   double amp, double mean, double* __restrict__ ws, double* __restrict__ mu,
                   double* __restrict__ var, int split, KernelFn kf) {
    // `split` is always 0.  The `split != k` tests below are opaque to the compiler and put the GEMM step and the
    // unrolled diagonal solve into basic blocks of their own: as ONE block the register allocator hoists the
    // solve's LDS reads into the GEMM phase and spills 100 VGPRs to scratch inside the hot loop
    // (measured on MI355X: 11.3 ms -> 6.9 ms per 65536 queries at N=2000).
    __shared__ double As[64][66];   // L[kb,j] block, then L[kb,kb]
    __shared__ double Vs[64][80];   // V_j tile (MFMA B operand), then the C tile
    __shared__ double xtr[D][64];   // scaled coordinates of training block kb
    __shared__ double alb[64];
    __shared__ double dis[64];      // 1 / L_rr of block kb
    const int tid = threadIdx.x, c = tid & 63, w = tid >> 6;
    const int lr = c & 15, lk = c >> 4;  // MFMA lane decomposition within the wave
    const int nb = Npad / 64, ld = Npad;
    double* V = ws + (size_t)blockIdx.x * Npad * 64;
    const long long ntiles = (M + 63) / 64;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m = tile * 64 + c;
        if (IDENT) V = ws + (size_t)tile * Npad * 64;
        const int kb0 = IDENT ? (int)tile : 0;       // first block row with a non-zero right-hand side
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = (!IDENT && m < M && k < d) ? Xs[m * d + k] * inv_len.v[k] : 0.0;
        double mu_acc = 0.0, ss = 0.0;
        for (int kb = kb0; kb < nb; ++kb) {
            __syncthreads();   // the previous block's solve has finished with As / Vs / xtr / alb / dis
            // Software pipeline: the 64x64 L block and V tile of step j+1 travel HBM/L2 -> registers (16 B per
            // lane, 8 + 8 loads) while step j runs on the matrix cores; the first stage is issued here so
            // that the K* evaluation below hides it.
            f64x2 pa[8], pv[8];
            {
                const double* Lb = L + (size_t)(kb * 64) * ld + (kb > kb0 ? kb0 * 64 : kb * 64);
                const double* V0 = V + (size_t)(kb0 * 64) * 64;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
                    pa[i] = *reinterpret_cast<const f64x2*>(Lb + (size_t)r * ld + 2 * c2);
                    if (kb > kb0 && split != 1) pv[i] = reinterpret_cast<const f64x2*>(V0)[e];
                }
            }
            for (int e = tid; e < D * 64; e += 256) xtr[e >> 6][e & 63] = Xt[(size_t)(e >> 6) * Npad + kb * 64 + (e & 63)];
            if (tid < 64) { alb[tid] = alpha[kb * 64 + tid]; dis[tid] = dinv[kb * 64 + tid]; }
            __syncthreads();
            // K* block: thread (column c, wave w) evaluates rows 16w .. 16w+15
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int r = w * 16 + i;
                double r2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double df = xtr[k][r] - q[k];
                    r2 = fma(df, df, r2);
                }
                double kv = (kb * 64 + r < N) ? amp * radial(r2, kf) : 0.0;
                if (IDENT) kv = (kb == kb0 && r == c) ? 1.0 : 0.0;
                mu_acc = fma(kv, alb[r], mu_acc);
                Vs[r][c] = kv;
            }
            __syncthreads();
            v4f64 acc[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[n][i] = Vs[16 * w + lk + 4 * i][16 * n + lr];
            for (int j = kb0; j < kb; ++j) {
                __syncthreads();       // every wave is done reading As / Vs
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
                    *reinterpret_cast<f64x2*>(&As[r][2 * c2]) = pa[i];
                    *reinterpret_cast<f64x2*>(&Vs[r][2 * c2]) = pv[i];
                }
                __syncthreads();
                {   // next stage: L[kb, j+1] and V_{j+1}, or the diagonal block L[kb, kb] after the last step
                    const int jn = j + 1;
                    const double* Lb = L + (size_t)(kb * 64) * ld + jn * 64;
                    const double* Vj = V + (size_t)(jn * 64) * 64;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
                        if (split != 2) pa[i] = *reinterpret_cast<const f64x2*>(Lb + (size_t)r * ld + 2 * c2);
                        if (jn < kb && split != 1) pv[i] = reinterpret_cast<const f64x2*>(Vj)[e];
                    }
                }
                if (split != 3)
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    double a = -As[16 * w + lr][4 * ks + lk];
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        double b = Vs[4 * ks + lk][16 * n + lr];
                        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) Vs[16 * w + lk + 4 * i][16 * n + lr] = acc[n][i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // pa holds L[kb,kb]; only its lower triangle is read below
                const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
                *reinterpret_cast<f64x2*>(&As[r][2 * c2]) = pa[i];
            }
            __syncthreads();
            // diagonal solve: wave w owns columns 16w..16w+15; lane (col lr, group lk) holds rows == lk (mod 4)
            const int col = 16 * w + lr;
            double v[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = Vs[4 * t + lk][col];
            if (split != 4)
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                const int owner = r & 3, t = r >> 2;
                double x = v[t] * dis[r];
                x = __shfl(x, lr + 16 * owner, 64);
                if (lk == owner) v[t] = x;
#pragma unroll
                for (int t2 = 0; t2 < 16; ++t2) {
                    if (4 * t2 + 3 > r) {
                        const int r2 = 4 * t2 + lk;
                        if (r2 > r) v[t2] = fma(-As[r2][r], x, v[t2]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                V[(size_t)(kb * 64 + 4 * t + lk) * 64 + col] = v[t];
                ss = fma(v[t], v[t], ss);
            }
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (IDENT) continue;
        {
            const long long mc = tile * 64 + 16 * w + lr;
            if (lk == 0 && mc < M) var[mc] = amp - ss;
        }
        __syncthreads();
        Vs[w][c] = mu_acc;
        __syncthreads();
        if (tid < 64 && m < M) mu[m] = ((Vs[0][c] + Vs[1][c]) + (Vs[2][c] + Vs[3][c])) + mean;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised variance kernel (the one launch_predict_var uses for d <= 16).  Same algorithm and data layout as
// predict_var_kernel above -- per 64-query tile a blocked forward substitution V = L^-1 K*^T, off-diagonal updates on the
// fp64 matrix cores -- but the workgroup is split into producers and consumers so that memory traffic and MFMA work overlap
// (ablation of the kernel above at N=2000, M=65536: MFMA alone 3.3 ms, loads + LDS staging + solves alone 4.15 ms,
// together 7.0 ms: they were serialised by the two barriers of every step).
//
//   768 threads, ONE workgroup per CU (LDS: two full stages, 155 KB).
//   waves 0-3 (consumers)  run the 64 MFMAs of a stage out of LDS buffer b&1, and the 64x64 diagonal solve at the end of a
//                          block row; nothing else.
//   waves 4-11 (producers) stream the flattened stage sequence (kb, s), s = kb0..kb (s == kb: the diagonal block):
//                          global -> registers two stages ahead, registers -> LDS buffer (b+1)&1 while the consumers work on
//                          stage b; during a diagonal solve they evaluate the K* block of block row kb+2 and park it in
//                          the workspace rows where V_{kb+2} will later be written (the consumers fetch it as their next
//                          accumulator while they solve).
//   One barrier per stage (two in a diagonal stage).  V_s must exist before a producer loads it: a stage whose V block is
//   not finished yet gets its V part issued late; the one stage that needs V the moment it is produced (block row kb0+1)
//   receives it from the consumers through LDS directly.
// Pointer parameters of a (non-inlined) device function are generic: loads through them are FLAT instructions, which
// count on the LDS counter as well and would serialise the prefetch pipeline with every LDS access.  The role functions
// therefore cast their buffer pointers to the global address space first.
typedef const double __attribute__((address_space(1)))* ws_gcptr;
typedef double __attribute__((address_space(1)))* ws_gptr;
typedef const f64x2 __attribute__((address_space(1)))* ws_gcptr2;
// Module-scope LDS (two full stages): named directly by the role functions, so every access stays an LDS instruction
// (passed as pointer arguments they degrade to generic pointers: 64-bit address arithmetic and a null check per access).
__shared__ double ws_As[2][64][66];
__shared__ double ws_Vs[2][64][80];
__shared__ double ws_dis[2][64];

// Producer waves of one tile.  Separate noinline functions per role: compiled as one body, the register allocator
// merges the live ranges of both roles and spills hundreds of VGPRs.  No store and no spill may sit in the stage loop: on
// gfx950 loads and stores share one counter, and with both kinds pending the compiler can only wait for ALL of them
// (vmcnt(0)), which would drain the stage that is meant to stay in flight.
template <int TM>
__device__ __attribute__((noinline)) void
ws_produce_tile(const double* L_, const double* dinv_, int Npad, const double* V_, int kb0) {
    const ws_gcptr L = (ws_gcptr)L_, dinv = (ws_gcptr)dinv_, V = (ws_gcptr)V_;
    const int t8 = threadIdx.x - 256;          // 256 producer threads: 8 + 8 loads per stage each
    // kb0: first block row with a non-zero right-hand side (0 for predictions; the tile's own block row when the columns of
    // the identity are pushed through to build L^-1): the stage sequence is (kb, s), kb0 <= s <= kb < nb
    const int nb = Npad / 64, ld = Npad, nr = nb - kb0, nstages = nr * (nr + 1) / 2;
    // Two register sets; every stage issues exactly 8 + TM/8 + 1 loads (a diagonal stage loads a V block it does not use), so
    // the wait for the OLDER set is a constant vmcnt and the younger set stays in flight.
    f64x2 pa[2][8], pv[2][TM / 8];
    double pd[2];
    // Thread t8 owns row t8 >> 2 of a 64 x 64 block and the 16-byte chunks (t8 & 3) + 4 i, i = 0..7, of that row: one base
    // address per block, the eight loads and LDS writes differ by immediates (64 B apart; a wave covers 16 rows x 64 B).
    const int prow = t8 >> 2, pch = t8 & 3;
#define ALABI_WS_ISSUE(SET, KB, S)                                                                            \
    {                                                                                                         \
        const ws_gcptr2 Lb_ = (ws_gcptr2)(L + ((size_t)((KB) * 64 + prow)) * ld + (S) * 64) + pch;            \
        const ws_gcptr2 Vj_ = (ws_gcptr2)(V + ((size_t)(((S) < (KB) ? (S) : kb0) * 64 + prow)) * TM) + pch;   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) pa[SET][i] = Lb_[4 * i];                                \
        _Pragma("unroll") for (int i = 0; i < TM / 8; ++i) pv[SET][i] = Vj_[4 * i];                           \
        pd[SET] = dinv[(KB) * 64 + (t8 & 63)];                                                                \
    }
#define ALABI_WS_TO_LDS(SET, BUF, WITH_V)                                                                     \
    {                                                                                                         \
        f64x2* as_ = reinterpret_cast<f64x2*>(&ws_As[BUF][prow][0]) + pch;                                    \
        f64x2* vs_ = reinterpret_cast<f64x2*>(&ws_Vs[BUF][prow][0]) + pch;                                    \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) as_[4 * i] = pa[SET][i];                                \
        if (WITH_V) { _Pragma("unroll") for (int i = 0; i < TM / 8; ++i) vs_[4 * i] = pv[SET][i]; }           \
        if (t8 < 64) ws_dis[BUF][t8] = pd[SET];                                                               \
    }
#define ALABI_WS_ADVANCE(KB, S) { if (++(S) > (KB)) { ++(KB); (S) = kb0; } }
    // ---- phase A: block rows kb0..kb0+3, one stage at a time (a V block may be needed a stage or two after it is produced) ----
    int b = 0, kb = kb0, sj = kb0;             // the stage the consumers are working on
    ALABI_WS_ISSUE(0, kb0, kb0)
    ALABI_WS_TO_LDS(0, 0, false)
    __syncthreads();                           // stage 0 is in LDS
    int k1 = kb0, s1 = kb0;                    // stage b + 1
    ALABI_WS_ADVANCE(k1, s1)
    while (b < nstages && kb < kb0 + 4) {
        const int nbuf = (b & 1) ^ 1;
        const bool diag = sj == kb;
        if (b > 0) __syncthreads();            // stage b is in LDS buffer b & 1; buffer nbuf is free
        if (diag) __syncthreads();             // the consumers' mid-stage barrier
        if (b + 1 < nstages) {
            ALABI_WS_ISSUE(0, k1, s1)
            // stage (kb0 + 1, kb0) needs V_kb0 the moment it is produced: the consumers hand it over in LDS themselves
            ALABI_WS_TO_LDS(0, nbuf, (s1 < k1 && b > 0))
        }
        ++b;
        ALABI_WS_ADVANCE(kb, sj)
        ALABI_WS_ADVANCE(k1, s1)
    }
    // ---- phase B: block rows >= kb0 + 4, two stages ahead.  Set 1 holds stage b + 1 for even b - bA, set 0 for odd. ----
    if (b < nstages) {
        // here (kb, sj) = (kb0 + 4, kb0) = stage b (already in LDS), (k1, s1) = stage b + 1
        int k2 = k1, s2 = s1;
        ALABI_WS_ADVANCE(k2, s2)               // stage b + 2
        ALABI_WS_ISSUE(1, k1, s1)
        ALABI_WS_ISSUE(0, k2, s2)
        int k3 = k2, s3 = s2;                  // stage b + 3, clamped to the last stage (harmless reloads at the very end)
#define ALABI_WS_STAGE(NSET)                                                                                  \
        {                                                                                                     \
            const int nbuf = (b & 1) ^ 1;                                                                     \
            const bool diag = sj == kb;                                                                       \
            __syncthreads();               /* stage b is in LDS buffer b & 1; buffer nbuf is free */          \
            if (diag) __syncthreads();     /* the consumers' mid-stage barrier */                             \
            ALABI_WS_TO_LDS(NSET, nbuf, true)                                                                 \
            if (k3 < nb - 1 || s3 < k3) ALABI_WS_ADVANCE(k3, s3)                                              \
            ALABI_WS_ISSUE(NSET, k3, s3)                                                                      \
            ++b;                                                                                              \
            ALABI_WS_ADVANCE(kb, sj)                                                                          \
        }
        while (b < nstages) {
            ALABI_WS_STAGE(1)
            if (b >= nstages) break;
            ALABI_WS_STAGE(0)
        }
#undef ALABI_WS_STAGE
    }
#undef ALABI_WS_ISSUE
#undef ALABI_WS_TO_LDS
#undef ALABI_WS_ADVANCE
}

// Consumer waves of one tile: returns this lane's share of |L^-1 k*|^2 (before the cross-lane fold).
template <int TM>
__device__ __attribute__((noinline)) double
ws_consume_tile(int Npad, double* V_, int kb0) {
    constexpr int NT = TM / 16;            // MFMA column tiles per wave
    constexpr int CPW = TM / 4;            // solve: columns per wave ...
    constexpr int LPC = 64 / CPW;          // ... lanes per column (lane group g holds rows g, g + LPC, ...)
    constexpr int RPL = 64 / LPC;          // ... rows per lane
    const ws_gptr V = (ws_gptr)V_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const int lr = lane & 15, lk = lane >> 4;  // MFMA lane decomposition
    const int cl = lane % CPW, g = lane / CPW;
    const int nb = Npad / 64;
    double ss = 0.0;
    __syncthreads();                   // stage 0 is in LDS
    v4f64 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[n][i] = V[(size_t)(kb0 * 64 + 16 * w + lk + 4 * i) * TM + 16 * n + lr];
    int b = 0;
    for (int kb = kb0; kb < nb; ++kb)
        for (int sj = kb0; sj <= kb; ++sj, ++b) {
            const int buf = b & 1, nbuf = buf ^ 1;
            if (b > 0) __syncthreads();    // stage b is in LDS buffer buf
            if (sj < kb) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const double a = -ws_As[buf][16 * w + lr][4 * ks + lk];
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const double bb = ws_Vs[buf][4 * ks + lk][16 * n + lr];
                        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[n], 0, 0, 0);
                    }
                }
                continue;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) ws_Vs[buf][16 * w + lk + 4 * i][16 * n + lr] = acc[n][i];
            if (kb + 1 < nb) {             // next block row's accumulator seed (the K* pre-pass left it in the workspace)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[n][i] = V[(size_t)((kb + 1) * 64 + 16 * w + lk + 4 * i) * TM + 16 * n + lr];
            }
            __syncthreads();
            // diagonal solve: wave w owns columns CPW w .. CPW w + CPW - 1; lane (col cl, group g) holds rows == g (mod LPC)
            const int col = CPW * w + cl;
            double v[RPL];
#pragma unroll
            for (int t = 0; t < RPL; ++t) v[t] = ws_Vs[buf][LPC * t + g][col];
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                const int owner = r % LPC, t = r / LPC;
                double x = v[t] * ws_dis[buf][r];
                x = __shfl(x, cl + CPW * owner, 64);
                // rows of the pivot's own group: lanes below the pivot are done (select, no branch: straight-line code);
                // every later group takes the update unconditionally
                const double a_own = ws_As[buf][LPC * t + g][r];
                v[t] = (g == owner) ? x : ((g > owner) ? fma(-a_own, x, v[t]) : v[t]);
#pragma unroll
                for (int t2 = t + 1; t2 < RPL; ++t2) v[t2] = fma(-ws_As[buf][LPC * t2 + g][r], x, v[t2]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < RPL; ++t) {
                V[(size_t)(kb * 64 + LPC * t + g) * TM + col] = v[t];
                if (kb == kb0) ws_Vs[nbuf][LPC * t + g][col] = v[t];   // the next stage needs this block at once: LDS hand-over
                ss = fma(v[t], v[t], ss);
            }
        }
    return ss;
}

// K* pre-pass of the wave-specialised variance path: one workgroup per 64-query tile writes the tile's K* block
// (Npad x 64, the accumulator seeds of every block row) into the tile's workspace and the predictive mean; the training
// points are staged 256 at a time in LDS exactly as in predict_mean_tile_kernel.
template <int D, bool GENERIC>
__global__ void __launch_bounds__(256)
predict_kstar_tile_kernel(const double* __restrict__ Xt, const double* __restrict__ alpha, int N, int Npad,
                          const double* __restrict__ Xs, int d, long long M, DimVec inv_len, double amp, double mean,
                          KernelFn kf, double* __restrict__ ws, double* __restrict__ mu, int TM,
                          double* __restrict__ mu_part, long long mu_stride) {
    __shared__ double xt[D][256];
    __shared__ double al[256];
    __shared__ double part[4][64];
    const int tid = threadIdx.x, c = tid & 63, w = tid >> 6;
    const long long m = (long long)blockIdx.x * 64 + c;
    // gridDim.y > 1: the training points are split into gridDim.y ranges of whole 256-point stages (few query tiles cannot fill
    // the chip otherwise); every range writes its K* rows and its part of the mean sum, predict_var_w_final_kernel adds the parts
    const int stages = (Npad + 255) / 256;
    const int st_lo = (int)((long long)stages * blockIdx.y / gridDim.y), st_hi = (int)((long long)stages * (blockIdx.y + 1) / gridDim.y);
    // 64 queries per workgroup = 64 / TM variance tiles of TM queries, each [Npad][TM] in the workspace
    double* V = ws + ((size_t)blockIdx.x * (64 / TM) + c / TM) * Npad * TM + (c % TM);
    double q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) q[k] = (m < M && k < d) ? Xs[m * d + k] * inv_len.v[k] : 0.0;
    double acc = 0.0;
    for (int n0 = st_lo * 256; n0 < st_hi * 256; n0 += 256) {
        __syncthreads();
        const int n = n0 + tid;
#pragma unroll
        for (int k = 0; k < D; ++k) xt[k][tid] = (n < Npad) ? Xt[(size_t)k * Npad + n] : 0.0;
        al[tid] = (n < Npad) ? alpha[n] : 0.0;
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < 64; ++j) {
            const int nn = w * 64 + j;
            if (n0 + nn >= Npad) break;                    // workgroup-uniform per wave
            double r2 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double df = xt[k][nn] - q[k];
                r2 = fma(df, df, r2);
            }
            const double kv = (n0 + nn < N) ? amp * radial<GENERIC>(r2, kf) : 0.0;
            acc = fma(kv, al[nn], acc);
            V[(size_t)(n0 + nn) * TM] = kv;
        }
    }
    part[w][c] = acc;
    __syncthreads();
    if (tid < 64 && m < M) {
        const double t = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        if (gridDim.y == 1) mu[m] = t + mean;
        else mu_part[(size_t)blockIdx.y * mu_stride + m] = t;
    }
}

// One workgroup per CU walks over the tiles of TM queries; tile t owns workspace rows ws[t] (K* seeds in, V out).
// TM = 64 for throughput; TM = 16 when there are too few queries to give every CU a 64-wide tile (a tile's latency is
// its 528 dependent stages at N = 2000, and a 16-wide stage is four times less MFMA work).
template <int TM>
__global__ void __launch_bounds__(512)
predict_var_ws_kernel(const double* __restrict__ L, const double* __restrict__ dinv, int Npad, long long M, double amp,
                      double* __restrict__ ws, double* __restrict__ var, int ident) {
    constexpr int CPW = TM / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ntiles = (M + TM - 1) / TM;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        double* V = ws + (size_t)tile * Npad * TM;
        __syncthreads();                       // the previous tile is completely finished with the LDS stages
        // identity right-hand sides (building L^-1): query m is column m, zero above its own block row
        const int kb0 = ident ? (int)((tile * TM) / 64) : 0;
        if (wv >= 4) {
            ws_produce_tile<TM>(L, dinv, Npad, V, kb0);
        } else {
            double ss = ws_consume_tile<TM>(Npad, V, kb0);   // lane (column lane % CPW, row group lane / CPW) of wave wv's columns
#pragma unroll
            for (int off = CPW; off < 64; off <<= 1) ss += __shfl_xor(ss, off, 64);
            const long long mc = tile * TM + CPW * wv + (lane % CPW);
            if (lane < CPW && mc < M) var[mc] = amp - ss;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Variance as a plain product with the cached W = L^-1:  var = amp - colnorm^2(W K*^T).  Same producer / consumer
// pipeline and the same stage sequence (kb, s), s <= kb, as predict_var_ws_kernel, but a stage is now
// acc += W[kb,s] K*_s with NO dependency between stages: nothing waits for a V block, there is no diagonal solve and no
// mid-stage barrier, and the block rows of one query tile can be split over several workgroups (`parts`) when there are
// fewer tiles than CUs.  K* comes from the pre-pass (never overwritten here).  Agrees with the substitution kernels to
// ~1e-14 amp at nugget e^-12 (tools/prof_small_batch.py).
__device__ inline void wg_row_range(int nb, int part, int parts, int& r0, int& r1) {
    // contiguous block rows with about the same number of stages: rows [r0, r1), stages of row r = r + 1
    const long long S = (long long)nb * (nb + 1) / 2;
    auto bound = [&](int p) {
        if (p <= 0) return 0;
        if (p >= parts) return nb;
        const long long target = S * p / parts;
        int r = (int)((sqrt(8.0 * (double)target + 1.0) - 1.0) * 0.5);
        while ((long long)r * (r + 1) / 2 < target) ++r;
        while (r > 0 && (long long)(r - 1) * r / 2 >= target) --r;
        return r < nb ? r : nb;
    };
    r0 = bound(part); r1 = bound(part + 1);
}

__device__ __attribute__((noinline)) void
wg_produce_tile(const double* W_, const double* K_, int Npad, int r0, int r1) {
    const ws_gcptr W = (ws_gcptr)W_, Kst = (ws_gcptr)K_;
    const int t8 = threadIdx.x - 256;          // 256 producer threads: 8 + 8 loads per stage each
    const int prow = t8 >> 2, pch = t8 & 3;
    const long long nst = (long long)r1 * (r1 + 1) / 2 - (long long)r0 * (r0 + 1) / 2;
    if (nst <= 0) return;
    f64x2 pa[2][8], pv[2][8];
#define ALABI_WG_ISSUE(SET, KB, S)                                                                            \
    {                                                                                                         \
        const ws_gcptr2 Wb_ = (ws_gcptr2)(W + (size_t)(S) * Npad * 64 + ((size_t)((KB) * 64 + prow)) * 64) + pch; \
        const ws_gcptr2 Kb_ = (ws_gcptr2)(Kst + ((size_t)((S) * 64 + prow)) * 64) + pch;                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) { pa[SET][i] = Wb_[4 * i]; pv[SET][i] = Kb_[4 * i]; }   \
    }
#define ALABI_WG_TO_LDS(SET, BUF)                                                                             \
    {                                                                                                         \
        f64x2* as_ = reinterpret_cast<f64x2*>(&ws_As[BUF][prow][0]) + pch;                                    \
        f64x2* vs_ = reinterpret_cast<f64x2*>(&ws_Vs[BUF][prow][0]) + pch;                                    \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) { as_[4 * i] = pa[SET][i]; vs_[4 * i] = pv[SET][i]; }   \
    }
#define ALABI_WG_ADVANCE(KB, S) { if (++(S) > (KB)) { ++(KB); (S) = 0; } }
    // stage b + 1 sits in set (b + 1) & 1; (k3, s3) is the next stage to issue, clamped to the last one
    int k3 = r0, s3 = 0;
    const int klast = r1 - 1;
    ALABI_WG_ISSUE(0, k3, s3)                  // stage 0
    if (k3 < klast || s3 < k3) ALABI_WG_ADVANCE(k3, s3)
    ALABI_WG_ISSUE(1, k3, s3)                  // stage 1
    ALABI_WG_TO_LDS(0, 0)
    if (k3 < klast || s3 < k3) ALABI_WG_ADVANCE(k3, s3)
    ALABI_WG_ISSUE(0, k3, s3)                  // stage 2
    __syncthreads();                           // stage 0 is in LDS
#define ALABI_WG_STAGE(NSET)                                                                                  \
    {                                                                                                         \
        __syncthreads();               /* stage b is in LDS buffer b & 1; the other buffer is free */         \
        ALABI_WG_TO_LDS(NSET, (int)((b & 1) ^ 1))                                                             \
        if (k3 < klast || s3 < k3) ALABI_WG_ADVANCE(k3, s3)                                                   \
        ALABI_WG_ISSUE(NSET, k3, s3)                                                                          \
        ++b;                                                                                                  \
    }
    long long b = 0;
    // the consumers run one barrier per stage after the first; stage 0 needs none here beyond the one above
    if (nst > 1) {
        // iteration for stage b writes stage b + 1: executed for b = 0 .. nst - 2, preceded by that stage's barrier
        // (stage 0's barrier is the __syncthreads above, so the first iteration skips it)
        ALABI_WG_TO_LDS(1, 1)
        if (k3 < klast || s3 < k3) ALABI_WG_ADVANCE(k3, s3)
        ALABI_WG_ISSUE(1, k3, s3)
        b = 1;
        while (b < nst - 1) {
            ALABI_WG_STAGE(0)
            if (b >= nst - 1) break;
            ALABI_WG_STAGE(1)
        }
        __syncthreads();                       // the barrier of the last stage
    }
#undef ALABI_WG_STAGE
#undef ALABI_WG_ISSUE
#undef ALABI_WG_TO_LDS
#undef ALABI_WG_ADVANCE
}

// Consumer waves: returns this lane's partial sums of squares for its four columns 16 n + lr (rows 16w + lk + 4i).
__device__ __attribute__((noinline)) v4f64
wg_consume_tile(int r0, int r1) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const int lr = lane & 15, lk = lane >> 4;
    v4f64 ss = v4f64{0.0, 0.0, 0.0, 0.0};
    if (r1 <= r0) return ss;
    __syncthreads();                           // stage 0 is in LDS
    v4f64 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = v4f64{0.0, 0.0, 0.0, 0.0};
    long long b = 0;
    for (int kb = r0; kb < r1; ++kb)
        for (int sj = 0; sj <= kb; ++sj, ++b) {
            const int buf = (int)(b & 1);
            if (b > 0) __syncthreads();        // stage b is in LDS buffer buf
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const double a = ws_As[buf][16 * w + lr][4 * ks + lk];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const double bb = ws_Vs[buf][4 * ks + lk][16 * n + lr];
                    acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[n], 0, 0, 0);
                }
            }
            if (sj == kb) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ss[n] = fma(acc[n][i], acc[n][i], ss[n]);
                    acc[n] = v4f64{0.0, 0.0, 0.0, 0.0};
                }
            }
        }
    return ss;
}

// grid = (tiles in flight, parts); partial[(tile * parts + part) * 64 + column]
__global__ void __launch_bounds__(512)
predict_var_w_kernel(const double* __restrict__ W, const double* __restrict__ Kst, int Npad, long long ntiles, int parts,
                     double* __restrict__ partial) {
    __shared__ double red[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int nb = Npad / 64, part = blockIdx.y;
    int r0, r1;
    wg_row_range(nb, part, parts, r0, r1);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const double* K = Kst + (size_t)tile * Npad * 64;
        __syncthreads();                       // the previous tile is completely finished with the LDS stages
        if (wv >= 4) {
            wg_produce_tile(W, K, Npad, r0, r1);
        } else {
            v4f64 ss = wg_consume_tile(r0, r1);
#pragma unroll
            for (int n = 0; n < 4; ++n) {      // over the row groups lk, then over the four waves (rows 16w ..)
                ss[n] += __shfl_xor(ss[n], 16, 64);
                ss[n] += __shfl_xor(ss[n], 32, 64);
                if (lk == 0) red[wv][16 * n + lr] = ss[n];
            }
        }
        __syncthreads();
        if (tid < 64) partial[((size_t)tile * parts + part) * 64 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same product with TWO matrix-core waves per SIMD: 128 queries per workgroup (two neighbouring 64-query K* tiles),
// eight consumer waves (row group c & 3, query half c >> 2: waves c and c + 4 share a SIMD and cover each other's LDS
// latency and barrier waits) and four producer waves.  A stage is half a 64 x 64 block of W (32 columns) against 32 rows
// of both K* tiles, so the LDS holds 2 x (64 x 33 + 32 x 144) doubles = 108 KB and every byte of W read from L2 / HBM feeds
// twice as many queries as in predict_var_w_kernel.  The accumulation order per output is unchanged (k ascending inside
// a block, blocks s ascending), so both kernels return the same bits.  tools/micro/mfma_f64_rate: 66-70 TFLOP/s with two
// MFMA waves per SIMD against 56-59 with one.
__shared__ double w2_As[2][64][33];
__shared__ double w2_Vs[2][32][144];

// Uniform values arrive in VGPRs through the call ABI of a noinline function: readfirstlane moves them back to SGPRs so
// that the stage counters, the address arithmetic and the loop branches are scalar.
__device__ inline const double* w2_uniform_ptr(const double* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const double*)(((unsigned long long)hi << 32) | lo);
}

__device__ __attribute__((noinline)) void
w2_produce_tile(const double* W_, const double* K0_, const double* K1_, int Npad, int r0, int r1) {
    const ws_gcptr W = (ws_gcptr)w2_uniform_ptr(W_), K0 = (ws_gcptr)w2_uniform_ptr(K0_), K1 = (ws_gcptr)w2_uniform_ptr(K1_);
    Npad = __builtin_amdgcn_readfirstlane(Npad); r0 = __builtin_amdgcn_readfirstlane(r0); r1 = __builtin_amdgcn_readfirstlane(r1);
    const int t8 = threadIdx.x - 512;          // 256 producer threads
    const int wrow = t8 >> 2, wch = t8 & 3;    // W: row of the block, 8 of its 32 columns (4 x 16-byte loads)
    const int krow = t8 >> 3, kch = t8 & 7;    // K*: row of the 32, 8 of the 64 queries of each tile (4 + 4 loads)
    const long long nst = 2 * ((long long)r1 * (r1 + 1) / 2 - (long long)r0 * (r0 + 1) / 2);
    if (nst <= 0) return;
    f64x2 pw[2][4], pk[2][8];
    int kb = r0, sj = 0, hf = 0;               // the next stage to issue; stays on the last stage once it is reached, so
    const int klast = r1 - 1;                  // that the issue is unconditional (straight-line code keeps vmcnt exact:
                                               // the wait before a set is written leaves the other set's loads in flight)
    const size_t wlane = (size_t)wrow * 64 + 8 * wch, klane = (size_t)krow * 64 + 8 * kch;
#define ALABI_W2_ISSUE(SET)                                                                                          \
    {                                                                                                                \
        const ws_gcptr2 Wb_ = (ws_gcptr2)(W + ((size_t)sj * Npad + (size_t)kb * 64) * 64 + 32 * hf + wlane);         \
        const size_t ko_ = ((size_t)(sj * 64 + 32 * hf)) * 64 + klane;                                               \
        const ws_gcptr2 Ka_ = (ws_gcptr2)(K0 + ko_), Kb_ = (ws_gcptr2)(K1 + ko_);                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { pw[SET][i] = Wb_[i]; pk[SET][i] = Ka_[i]; pk[SET][4 + i] = Kb_[i]; } \
        const int adv_ = (kb < klast || sj < kb || hf == 0) ? 1 : 0;                                                 \
        const int nh_ = hf ^ adv_, carry_ = adv_ & hf;                                                               \
        const int wrap_ = carry_ & (sj >= kb ? 1 : 0);                                                               \
        hf = nh_; sj = wrap_ ? 0 : sj + carry_; kb += wrap_;                                                         \
    }
#define ALABI_W2_TO_LDS(SET)                                                                                         \
    {                                                                                                                \
        double* as_ = &w2_As[SET][wrow][8 * wch];                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { as_[2 * i] = pw[SET][i][0]; as_[2 * i + 1] = pw[SET][i][1]; } \
        f64x2* va_ = reinterpret_cast<f64x2*>(&w2_Vs[SET][krow][8 * kch]);                                           \
        f64x2* vb_ = reinterpret_cast<f64x2*>(&w2_Vs[SET][krow][64 + 8 * kch]);                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { va_[i] = pk[SET][i]; vb_[i] = pk[SET][4 + i]; }              \
    }
    // Step j (after barrier #j, while the consumers work on stage j): stage j + 1 goes from register set (j + 1) & 1 to LDS
    // buffer (j + 1) & 1 -- free since barrier #j -- and the set is refilled with stage j + 3; then barrier #(j + 1).
    // Two steps per loop iteration with no branch between them: the wait before a set is written is vmcnt(12), the other
    // set's twelve loads stay in flight across the barrier.
#define ALABI_W2_STEP(SET) { ALABI_W2_TO_LDS(SET) ALABI_W2_ISSUE(SET) __syncthreads(); }
    ALABI_W2_ISSUE(0)                          // stage 0
    ALABI_W2_ISSUE(1)                          // stage 1
    ALABI_W2_TO_LDS(0)
    ALABI_W2_ISSUE(0)                          // stage 2
    __syncthreads();                           // barrier #0: stage 0 is in LDS
    long long j = 0;
    for (; j + 2 <= nst - 1; j += 2) {
        ALABI_W2_STEP(1)
        ALABI_W2_STEP(0)
    }
    if (j < nst - 1) ALABI_W2_STEP(1)
#undef ALABI_W2_TO_LDS
#undef ALABI_W2_STEP
#undef ALABI_W2_ISSUE
}

// Consumer wave c: rows 32 (c & 1) .. +31 of the block row (two 16-row groups), queries 32 (c >> 1) .. +31 (two 16-query
// groups): two A and two B operands feed four MFMAs per k-step (1.0 LDS read per MFMA instead of 1.25 for a 16 x 64
// wave tile -- the LDS bandwidth, not the matrix cores, is the tighter budget with eight consumer waves).  Returns the
// partial sums of squares ss[2 ri + n] of row group ri, query group n (rows lk + 4 i of the row group), accumulated in
// the same order as predict_var_w_kernel does for its row group 2 (c & 1) + ri.
__device__ __attribute__((noinline)) v4f64
w2_consume_tile(int r0, int r1) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    r0 = __builtin_amdgcn_readfirstlane(r0); r1 = __builtin_amdgcn_readfirstlane(r1);
    const int row0 = 32 * (c & 1), q0 = 32 * (c >> 1);
    const int lr = lane & 15, lk = lane >> 4;
    v4f64 ss = v4f64{0.0, 0.0, 0.0, 0.0};
    v4f64 acc[2][2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[ri][n] = v4f64{0.0, 0.0, 0.0, 0.0};
    int buf = 0;
    for (int kb = r0; kb < r1; ++kb)
        for (int sj = 0; sj <= kb; ++sj)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                __syncthreads();               // this stage is in LDS buffer buf
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const double a0 = w2_As[buf][row0 + lr][4 * ks + lk], a1 = w2_As[buf][row0 + 16 + lr][4 * ks + lk];
                    const double b0 = w2_Vs[buf][4 * ks + lk][q0 + lr], b1 = w2_Vs[buf][4 * ks + lk][q0 + 16 + lr];
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                }
                buf ^= 1;
                if (sj == kb && hf == 1) {
#pragma unroll
                    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) ss[2 * ri + n] = fma(acc[ri][n][i], acc[ri][n][i], ss[2 * ri + n]);
                            acc[ri][n] = v4f64{0.0, 0.0, 0.0, 0.0};
                        }
                }
            }
    return ss;
}

// grid = (groups of two K* tiles in flight, parts); partial[(tile * parts + part) * 64 + column] as predict_var_w_kernel
__global__ void __launch_bounds__(768)
predict_var_w2_kernel(const double* __restrict__ W, const double* __restrict__ Kst, int Npad, long long ntiles, int parts,
                      double* __restrict__ partial) {
    __shared__ double red[4][128];             // [row group of the block row][query of the 128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int nb = Npad / 64, part = blockIdx.y;
    int r0, r1;
    wg_row_range(nb, part, parts, r0, r1);
    const long long ngroups = (ntiles + 1) / 2;
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long long t0 = 2 * g, t1 = (2 * g + 1 < ntiles) ? 2 * g + 1 : t0;    // an odd last tile is paired with itself
        __syncthreads();                       // the previous group is completely finished with the LDS stages
        if (wv >= 8) {
            w2_produce_tile(W, Kst + (size_t)t0 * Npad * 64, Kst + (size_t)t1 * Npad * 64, Npad, r0, r1);
        } else {
            v4f64 ss = w2_consume_tile(r0, r1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {      // q = 2 ri + n
                ss[q] += __shfl_xor(ss[q], 16, 64);
                ss[q] += __shfl_xor(ss[q], 32, 64);
                if (lk == 0) red[2 * (wv & 1) + (q >> 1)][32 * (wv >> 1) + 16 * (q & 1) + lr] = ss[q];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const long long tile = 2 * g + (tid >> 6);
            if (tile < ntiles)
                partial[((size_t)tile * parts + part) * 64 + (tid & 63)] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        }
    }
}

__global__ void __launch_bounds__(256)
predict_var_w_final_kernel(const double* __restrict__ partial, int parts, long long M, double amp, double* __restrict__ var,
                           const double* __restrict__ mu_part, int nsplit, long long mu_stride, double mean, double* __restrict__ mu) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const double* p = partial + (size_t)(m / 64) * parts * 64 + (m % 64);
    double s = 0.0;
    for (int k = 0; k < parts; ++k) s += p[(size_t)k * 64];
    var[m] = amp - s;
    if (nsplit > 1) {                                      // the mean sum arrived in nsplit ranges of training points
        double t = 0.0;
        for (int k = 0; k < nsplit; ++k) t += mu_part[(size_t)k * mu_stride + m];
        mu[m] = t + mean;
    }
}

static int ensure_mupart(alabi_gp* gp, size_t bytes, hipStream_t s);

// ---- predict-mean with the dot products on the matrix cores (large batches) -------------------------------------
// mu*(q) = amp sum_n alpha_n f(r2(q, x_n)) + m (reference call sites alabi/core.py:85, :1812).  With the rows
// x' = (x, -|x|^2/2, 1) and q' = (q, 1, -|q|^2/2) the product q'.x' IS -r2/2, the argument of the squared-exponential:
// v_mfma_f64_16x16x4 forms it for 16 queries x 16 training points per instruction (ceil((d+2)/4) of them per tile) while the
// vector unit only evaluates exp and the alpha FMA -- 21 instead of 41 fp64 VALU instructions per kernel evaluation, and
// the two pipes run side by side.  One wavefront owns 64 queries (four 16-row A operands kept in registers) and walks over
// the training points 16 at a time (B operand and alpha: one double per lane per k-step, L2-resident); a lane ends with
// the partial sums of 16 queries over its point column, folded across the 16 lanes of its row with DPP adds.
// Centre of the scaled training inputs, one workgroup per dimension: r2 is translation invariant, and the augmented dot
// product q'.x' = q.x - |x|^2/2 - |q|^2/2 loses eps * (|x|^2 + |q|^2) / 2 absolutely, so both sides are taken relative to the
// training mean (inputs that sit thousands of length scales from the origin would otherwise cost digits of the exponent).
__global__ void __launch_bounds__(256)
xa_centre_kernel(const double* __restrict__ Xt, int Npad, int N, double* __restrict__ centre) {
    __shared__ double scratch[16];
    const int k = blockIdx.x;
    double t = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) t += Xt[(size_t)k * Npad + n];
    t = block_sum(t, scratch);
    if (threadIdx.x == 0) centre[k] = t / (double)N;
}

__global__ void __launch_bounds__(256)
build_xa_kernel(const double* __restrict__ Xt, int Npad, int d, int rows, const double* __restrict__ centre,
                double* __restrict__ Xa) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= Npad) return;
    double xx = 0.0;
    for (int k = 0; k < d; ++k) {
        const double v = Xt[(size_t)k * Npad + n] - centre[k];
        Xa[(size_t)k * Npad + n] = v;
        xx = fma(v, v, xx);
    }
    Xa[(size_t)d * Npad + n] = -0.5 * xx;
    Xa[(size_t)(d + 1) * Npad + n] = 1.0;
    for (int k = d + 2; k < rows; ++k) Xa[(size_t)k * Npad + n] = 0.0;
}

template <int KS, bool GENERIC>
__global__ void __launch_bounds__(256)
predict_mean_mfma_kernel(const double* __restrict__ Xa, const double* __restrict__ alpha, int Npad,
                         const double* __restrict__ Xs, int d, long long M, DimVec inv_len, const double* __restrict__ centre,
                         double amp, double mean, KernelFn kf, double* __restrict__ mu, double* __restrict__ mu_part, long long mu_stride,
                         int pts_per_part) {
    __shared__ double etab[256];                                 // 2^(j/256) for exp2s_tab256
    etab[threadIdx.x] = exp2((double)threadIdx.x * 0.00390625);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // squared exponential: the query operand carries the factor 256 / ln2, so the product arrives as the argument of 2^(./256)
    const double qs = GENERIC ? 1.0 : ALABI_EXP2S256_SCALE;
    const int lr = lane & 15, lk = lane >> 4;
    const long long q0 = ((long long)blockIdx.x * 4 + wv) * 64;
    if (q0 >= M) return;
    // A operands: query row lr of tile qt, coordinate 4 s + lk of k-step s (the augmented row q' = (q / l - c, 1, -|q / l - c|^2 / 2),
    // c = centre of the scaled training inputs, the same shift build_xa_kernel applied to the rows of Xa)
    double a[4][KS];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const long long m = q0 + 16 * qt + lr;
        const bool valid = m < M;
        double qq = 0.0;
        for (int k = 0; k < d; ++k) {
            const double v = valid ? Xs[m * d + k] * inv_len.v[k] - centre[k] : 0.0;
            qq = fma(v, v, qq);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 4 * s + lk;
            double v = 0.0;
            if (valid) v = (c < d) ? Xs[m * d + c] * inv_len.v[c] - centre[c] : (c == d) ? 1.0 : (c == d + 1) ? -0.5 * qq : 0.0;
            a[qt][s] = v * qs;
        }
    }
    double sum[4][4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[qt][i] = 0.0;
    const double* xb = Xa + (size_t)lk * Npad + lr;       // B operand: point column lr of the tile, coordinate 4 s + lk
    // gridDim.y > 1 (medium batches: too few 256-query workgroups to fill the chip): the training points are dealt to gridDim.y
    // workgroups per query block in runs of pts_per_part (a multiple of 16); mean_combine_kernel adds the parts in order
    const int n_lo = blockIdx.y * pts_per_part, n_hi = (n_lo + pts_per_part < Npad) ? n_lo + pts_per_part : Npad;
    for (int n0 = n_lo; n0 < n_hi; n0 += 16) {
        double b[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = xb[(size_t)(4 * s) * Npad + n0];
        const double al = alpha[n0 + lr];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qt][s], b[s], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {                 // C/D layout: row (query) lk + 4 i, column (point) lr
                const double f = GENERIC ? radial<true>(fmax(-2.0 * acc[i], 0.0), kf) : exp2s_tab256(acc[i], etab);
                sum[qt][i] = fma(al, f, sum[qt][i]);
            }
        }
    }
    // fold the 16 point columns of every query: the 16 lanes of a DPP row share lk
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = sum[qt][i];
            v += dpp_move<0x111, 0xf>(v);
            v += dpp_move<0x112, 0xf>(v);
            v += dpp_move<0x114, 0xf>(v);
            v += dpp_move<0x118, 0xf>(v);
            const long long m = q0 + 16 * qt + lk + 4 * i;
            if (lr == 15 && m < M) {
                if (mu_part) mu_part[(size_t)blockIdx.y * mu_stride + m] = v;
                else mu[m] = fma(amp, v, mean);
            }
        }
}

int ensure_xa(alabi_gp* gp, hipStream_t s) {
    const int rows = round_up(gp->d + 2, 4);
    if (!gp->Xa) ALABI_HIP_CHECK(hipMalloc(&gp->Xa, (size_t)rows * gp->n_cap * sizeof(double)));
    if (!gp->xa_centre) ALABI_HIP_CHECK(hipMalloc(&gp->xa_centre, ALABI_MAX_DIM * sizeof(double)));
    if (gp->xa_gen != gp->factor_gen) {
        hipLaunchKernelGGL(xa_centre_kernel, dim3(gp->d), dim3(256), 0, s, gp->Xt, gp->Npad, gp->N, gp->xa_centre);
        hipLaunchKernelGGL(build_xa_kernel, dim3((gp->Npad + 255) / 256), dim3(256), 0, s, gp->Xt, gp->Npad, gp->d, rows,
                           gp->xa_centre, gp->Xa);
        ALABI_LAUNCH_CHECK();
        gp->xa_gen = gp->factor_gen;
    }
    return ALABI_OK;
}

#define ALABI_DISPATCH_KS(KS_, ...)                                                             \
    switch (KS_) {                                                                              \
        case 1: { constexpr int KS = 1; __VA_ARGS__; } break;                                   \
        case 2: { constexpr int KS = 2; __VA_ARGS__; } break;                                   \
        case 3: { constexpr int KS = 3; __VA_ARGS__; } break;                                   \
        case 4: { constexpr int KS = 4; __VA_ARGS__; } break;                                   \
        case 5: { constexpr int KS = 5; __VA_ARGS__; } break;                                   \
        case 6: { constexpr int KS = 6; __VA_ARGS__; } break;                                   \
        case 7: { constexpr int KS = 7; __VA_ARGS__; } break;                                   \
        case 8: { constexpr int KS = 8; __VA_ARGS__; } break;                                   \
        default: return ALABI_BAD_ARGUMENT;                                                     \
    }

int launch_predict_mean(alabi_gp* gp, const double* Xs, long long M, double* mu, hipStream_t s) {
    if (M <= 0) return ALABI_OK;
    const int db = dim_bucket(gp->d);
    const double amp = exp(gp->log_amp);
    const char* mf = getenv("ALABI_PM_MFMA");
    // d + 2 <= 32 and at least 2048 queries: the matrix-core kernel.  256-query workgroups alone fill the chip from ~200 000 queries
    // on; below that (round 3) the training points are split over `parts` workgroups per query block as well, three workgroups
    // per CU in all, and mean_combine_kernel adds the parts.  Measured at N = 2000, d = 10 (vector kernels before -> now):
    // 2048 queries 28 -> 21 us, 4096 53 -> 25, 10^4 48 -> 38, 16384 63 -> 47, 32768 113 (unsplit matrix-core kernel) -> 79,
    // 65536 192 -> 139; N = 5000: 10^4 queries 104 -> 77 us; N = 10000, d = 20: 310 -> 203 us (gpurun_out/s2_pm_split*.txt)
    if (M >= 2048 && gp->d + 2 <= 32 && !(mf && mf[0] == '0')) {
        int st = ensure_xa(gp, s);
        if (st != ALABI_OK) return st;
        const int ks = (gp->d + 2 + 3) / 4;
        const long long wgs = (M + 255) / 256;
        if (wgs > 0x7fffffffLL) return ALABI_BAD_ARGUMENT;
        int dev = 0, n_cu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        int parts = 1, pts = gp->Npad;
        const long long per_cu = 3;
        if (wgs < per_cu * n_cu) {
            parts = (int)((per_cu * n_cu + wgs - 1) / wgs);
            const int tiles16 = gp->Npad / 16;
            if (parts > tiles16 / 4) parts = tiles16 / 4;            // at least four 16-point tiles per part
            if (parts < 1) parts = 1;
            pts = ((tiles16 + parts - 1) / parts) * 16;
            parts = (gp->Npad + pts - 1) / pts;
        }
        const long long mu_stride = wgs * 256;
        if (parts > 1 && (st = ensure_mupart(gp, (size_t)parts * mu_stride * sizeof(double), s)) != ALABI_OK) return st;
        ALABI_DISPATCH_KS(ks, ALABI_DISPATCH_KERNEL(gp->kf.type, hipLaunchKernelGGL((predict_mean_mfma_kernel<KS, GENERIC>), dim3((unsigned)wgs, parts),
            dim3(256), 0, s, gp->Xa, gp->alpha, gp->Npad, Xs, gp->d, M, gp->inv_len, gp->xa_centre, amp, gp->mean, gp->kf, mu,
            parts > 1 ? gp->mupart : (double*)nullptr, mu_stride, pts)));
        if (parts > 1)
            hipLaunchKernelGGL(mean_combine_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, gp->mupart, parts, mu_stride, M, amp,
                               gp->mean, mu);
        ALABI_LAUNCH_CHECK();
        return ALABI_OK;
    }
    if (M <= 4096) {
        ALABI_DISPATCH_DIM(db, hipLaunchKernelGGL(predict_mean_rowwise_kernel<D>, dim3((unsigned)M), dim3(256), 0, s,
                                                  gp->Xt, gp->alpha, gp->Npad, Xs, gp->d, gp->inv_len, amp,
                                                  gp->mean, gp->kf, mu));
    } else {
        const long long tiles = (M + 63) / 64;
        if (tiles > 0x7fffffffLL) return ALABI_BAD_ARGUMENT;
        int dev = 0, n_cu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        int nsplit = 1;
        if (tiles < 2LL * n_cu) {                              // fewer tiles than two per CU: split the training points
            const int stages = (gp->Npad + 255) / 256;
            nsplit = (int)(2LL * n_cu / tiles);
            if (nsplit > stages) nsplit = stages;
            if (nsplit > 8) nsplit = 8;
            if (nsplit < 1) nsplit = 1;
        }
        const long long mu_stride = tiles * 64;
        if (nsplit > 1) {
            int st = ensure_mupart(gp, (size_t)nsplit * mu_stride * sizeof(double), s);
            if (st != ALABI_OK) return st;
        }
        ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(gp->kf.type, hipLaunchKernelGGL((predict_mean_tile_kernel<D, GENERIC>), dim3((unsigned)tiles, nsplit), dim3(256), 0, s,
                                                  gp->Xt, gp->alpha, gp->Npad, Xs, gp->d, M, gp->inv_len, amp,
                                                  gp->mean, gp->kf, mu, gp->mupart, mu_stride)));
        if (nsplit > 1)
            hipLaunchKernelGGL(mean_combine_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, gp->mupart, nsplit, mu_stride, M, amp,
                               gp->mean, mu);
    }
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int ensure_winv(alabi_gp* gp, hipStream_t s);
static int ensure_small(alabi_gp* gp, size_t bytes, hipStream_t s);

static int ensure_mupart(alabi_gp* gp, size_t bytes, hipStream_t s) {
    if (gp->mupart_bytes >= bytes) return ALABI_OK;
    if (gp->mupart) {
        ALABI_HIP_CHECK(hipStreamSynchronize(s));
        ALABI_HIP_CHECK(hipFree(gp->mupart));
        gp->mupart = nullptr; gp->mupart_bytes = 0;
    }
    ALABI_HIP_CHECK(hipMalloc(&gp->mupart, bytes));
    gp->mupart_bytes = bytes;
    return ALABI_OK;
}

// Should this variance request go through the cached L^-1?  (see the comment in launch_predict_var)
int want_winv(alabi_gp* gp, long long M) {
    const char* envw = getenv("ALABI_PV_W");
    if (gp->req_gen != gp->factor_gen) { gp->req_gen = gp->factor_gen; gp->var_requests = 0; }
    gp->var_requests++;
    if (envw && envw[0] == '0') return 0;
    if (envw && envw[0] == '1') return 1;
    // With the recursive block inversion the cache costs less than the dependency chain of ONE substitution launch
    // (0.24 vs 0.6 ms at N = 2000, 2.1 vs 3.5 ms at N = 5000), so it is built on the first request; the substitution
    // kernels remain the fallback when the two Npad^2 buffers cannot be had.
    (void)M;
    return 1;
}

int launch_predict_var(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var, hipStream_t s) {
    if (M <= 0) return ALABI_OK;
    const int db = dim_bucket(gp->d);
    const long long tiles = (M + 63) / 64;
    const char* legacy = getenv("ALABI_PV_LEGACY");
    const bool ws_kernel = db <= 32 && !(legacy && legacy[0] == '1');
    const double amp = exp(gp->log_amp);
    if (ws_kernel) {
        // wave-specialised path: every tile owns Npad x 64 doubles of workspace (K* seeds in, V out); queries are processed
        // in chunks so that the workspace stays around 2 GiB (131072 queries at N = 2048)
        int dev = 0, n_cu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        // 16-wide tiles when 64-wide ones cannot give every CU a tile
        const int TM = (M <= 16LL * n_cu) ? 16 : 64;
        // tiles per chunk: a whole number of rounds over the CUs within ~2 GiB of workspace (at least one round)
        long long chunk_tiles = (2LL << 30) / ((long long)gp->Npad * 64 * 8);     // in 64-query units
        chunk_tiles = chunk_tiles / (2 * n_cu) * (2 * n_cu);      // predict_var_w2_kernel takes two tiles per workgroup
        if (chunk_tiles < 2 * n_cu) chunk_tiles = 2 * n_cu;
        if (const char* env = getenv("ALABI_PV_CHUNK_TILES")) { const long long v = atoll(env); if (v > 0) chunk_tiles = v; }   // tests
        long long chunk = chunk_tiles * 64;                                       // queries per chunk
        if (chunk > M) chunk = (M + 63) / 64 * 64;
        size_t need = (size_t)(chunk / 64) * gp->Npad * 64 * sizeof(double);
        if (need > gp->ws_bytes) {
            if (gp->ws) {
                ALABI_HIP_CHECK(hipStreamSynchronize(s));
                dev_cache_give(gp->ws, gp->ws_bytes);
                gp->ws = nullptr; gp->ws_bytes = 0;
            }
            // short of memory: halve the chunk down to one round over the CUs before giving up
            size_t got = 0;
            while (dev_alloc_cached((void**)&gp->ws, need, &got) != (int)hipSuccess) {
                if (chunk <= 64LL * n_cu) return ALABI_HIP_ERROR;
                chunk = (chunk / 2 + 63) / 64 * 64;
                if (chunk < 64LL * n_cu) chunk = 64LL * n_cu;
                need = (size_t)(chunk / 64) * gp->Npad * 64 * sizeof(double);
            }
            gp->ws_bytes = got;
        }
        // Product with the cached L^-1 (no dependency between stages, block rows of a tile split over `parts` workgroups
        // when the tiles alone cannot fill the chip); the substitution kernel when there is no room for the cache.
        // Building the cache (recursive block inversion, 0.24 ms at N = 2000, 2.1 ms at N = 5000) is cheaper than the chain of
        // one substitution launch, so it is built on the first request.  ALABI_PV_W=0 forbids it.
        int use_w = want_winv(gp, M);
        if (use_w) {
            const int stw = ensure_winv(gp, s);
            if (stw == ALABI_NOT_COMPUTED) use_w = 0;
            else if (stw != ALABI_OK) return stw;
        }
        for (long long m0 = 0; m0 < M; m0 += chunk) {
            const long long mc = (M - m0 < chunk) ? M - m0 : chunk;
            const long long groups = (mc + 63) / 64;                               // K* workgroups (64 queries each)
            const int TMc = use_w ? 64 : TM;
            // few query tiles: the K* pre-pass splits the training points over gridDim.y workgroups per tile (W path only)
            int nsplit = 1;
            if (use_w && groups < 2LL * n_cu) {                               // (two workgroups of the pre-pass fit on a CU)
                const int stages = (gp->Npad + 255) / 256;
                nsplit = (int)(2LL * n_cu / groups);
                if (nsplit > stages) nsplit = stages;
                if (nsplit > 8) nsplit = 8;
                if (nsplit < 1) nsplit = 1;
            }
            const long long mu_stride = (mc + 63) / 64 * 64;
            double* mu_part = nullptr;
            if (nsplit > 1) {
                int stp = ensure_mupart(gp, (size_t)nsplit * mu_stride * sizeof(double), s);
                if (stp != ALABI_OK) return stp;
                mu_part = gp->mupart;
            }
            ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(gp->kf.type, hipLaunchKernelGGL((predict_kstar_tile_kernel<D, GENERIC>),
                dim3((unsigned)groups, nsplit), dim3(256), 0, s, gp->Xt, gp->alpha, gp->N, gp->Npad, Xs + m0 * gp->d, gp->d, mc,
                gp->inv_len, amp, gp->mean, gp->kf, gp->ws, mu + m0, TMc, mu_part, mu_stride)));
            if (use_w) {
                const int nb = gp->Npad / 64;
                // two MFMA waves per SIMD (128 queries per workgroup) from 2048 queries on; below that the 64-query
                // workgroups spread over more CUs (0.24 vs 0.28 ms at 256 queries, N = 2000)
                const char* e2 = getenv("ALABI_PV_W2");
                const bool w2 = groups >= 32 && !(e2 && e2[0] == '0');
                const long long wgs = w2 ? (groups + 1) / 2 : groups;            // workgroups' worth of queries
                int parts = 1;
                if (wgs < n_cu) { parts = (int)(n_cu / wgs); if (parts > nb) parts = nb; if (parts < 1) parts = 1; }
                const int gx = (int)(wgs < n_cu ? wgs : n_cu);
                int st2 = ensure_small(gp, (size_t)groups * parts * 64 * sizeof(double), s);
                if (st2 != ALABI_OK) return st2;
                if (w2)
                    hipLaunchKernelGGL(predict_var_w2_kernel, dim3(gx, parts), dim3(768), 0, s, gp->winv, gp->ws, gp->Npad, groups,
                                       parts, gp->small);
                else
                    hipLaunchKernelGGL(predict_var_w_kernel, dim3(gx, parts), dim3(512), 0, s, gp->winv, gp->ws, gp->Npad, groups,
                                       parts, gp->small);
                hipLaunchKernelGGL(predict_var_w_final_kernel, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, s, gp->small, parts, mc,
                                   amp, var + m0, mu_part, nsplit, mu_stride, gp->mean, mu + m0);
                continue;
            }
            const long long tiles_c = (mc + TM - 1) / TM;
            const int grid_c = (int)(tiles_c < n_cu ? tiles_c : n_cu);
            if (TM == 16)
                hipLaunchKernelGGL(predict_var_ws_kernel<16>, dim3(grid_c), dim3(512), 0, s, gp->L, gp->dinv, gp->Npad, mc, amp,
                                   gp->ws, var + m0, 0);
            else
                hipLaunchKernelGGL(predict_var_ws_kernel<64>, dim3(grid_c), dim3(512), 0, s, gp->L, gp->dinv, gp->Npad, mc, amp,
                                   gp->ws, var + m0, 0);
        }
        ALABI_LAUNCH_CHECK();
        return ALABI_OK;
    }
    const int grid = (int)(tiles < 512 ? tiles : 512);
    const size_t need = (size_t)grid * gp->Npad * 64 * sizeof(double);
    if (need > gp->ws_bytes) {
        if (gp->ws) {
            ALABI_HIP_CHECK(hipStreamSynchronize(s));
            dev_cache_give(gp->ws, gp->ws_bytes);
            gp->ws = nullptr; gp->ws_bytes = 0;
        }
        size_t got = 0;
        if (dev_alloc_cached((void**)&gp->ws, need, &got) != (int)hipSuccess) return ALABI_HIP_ERROR;
        gp->ws_bytes = got;
    }
    const int split = 0;   // a run-time zero the kernel tests against (see its first comment): keeps two code regions apart
    ALABI_DISPATCH_DIM(db, hipLaunchKernelGGL(predict_var_kernel<D>, dim3(grid), dim3(256), 0, s, gp->L, gp->dinv, gp->Xt,
                                              gp->alpha, gp->N, gp->Npad, Xs, gp->d, M, gp->inv_len, amp,
                                              gp->mean, gp->ws, mu, var, split, gp->kf));
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

__global__ void __launch_bounds__(256)
ident_seed_kernel(double* __restrict__ seeds, int Npad) {
    // 16-wide tiles: seeds[t][n][c] = (n == 16 t + c)
    const size_t n_el = (size_t)Npad * Npad;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n_el; e += (size_t)gridDim.x * 256) {
        const size_t t = e / ((size_t)Npad * 16), rem = e % ((size_t)Npad * 16);
        seeds[e] = ((rem >> 4) == 16 * t + (rem & 15)) ? 1.0 : 0.0;
    }
}

__global__ void __launch_bounds__(256)
retile_16_to_64_kernel(const double* __restrict__ src, double* __restrict__ dst, int Npad) {
    // src[t16][n][16] -> dst[t64][n][64]
    const size_t n_el = (size_t)Npad * Npad;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n_el; e += (size_t)gridDim.x * 256) {
        const size_t t64 = e / ((size_t)Npad * 64), rem = e % ((size_t)Npad * 64);
        const size_t n = rem >> 6, c = rem & 63;
        dst[e] = src[((t64 * 4 + (c >> 4)) * Npad + n) * 16 + (c & 15)];
    }
}

// L^-1 tile-major into dst: dst[t] = L^-1[:, 64t:64t+64] as [Npad][64].  The columns of the identity are pushed through the
// wave-specialised substitution kernel as 16-wide tiles (Npad / 16 independent workgroups, 1.3 us per stage) into the
// variance workspace and re-tiled; the first-generation kernel (Npad / 64 workgroups, 3.3 us per stage) when the workspace
// cannot be had.
int launch_factor_inverse_into(alabi_gp* gp, double* dst, hipStream_t s) {
    const int nb = gp->Npad / 64;
    const size_t need = (size_t)gp->Npad * gp->Npad * sizeof(double);
    bool fast = dst != gp->ws && nb >= 2;
    if (fast && need > gp->ws_bytes) {
        if (gp->ws) {
            ALABI_HIP_CHECK(hipStreamSynchronize(s));
            dev_cache_give(gp->ws, gp->ws_bytes);
            gp->ws = nullptr; gp->ws_bytes = 0;
        }
        size_t got = 0;
        if (dev_alloc_cached((void**)&gp->ws, need, &got) != (int)hipSuccess) fast = false;
        else gp->ws_bytes = got;
    }
    if (fast) {
        // recursive block inversion on the matrix cores (gp_inverse.hip); ALABI_WINV_DNC=0 keeps the substitution chains
        const char* ednc = getenv("ALABI_WINV_DNC");
        if (!(ednc && ednc[0] == '0')) return launch_factor_inverse_dnc(gp, gp->ws, dst, s);
        int dev = 0, n_cu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        const int tiles = gp->Npad / 16;
        hipLaunchKernelGGL(ident_seed_kernel, dim3(1024), dim3(256), 0, s, gp->ws, gp->Npad);
        hipLaunchKernelGGL(predict_var_ws_kernel<16>, dim3(tiles < n_cu ? tiles : n_cu), dim3(512), 0, s, gp->L, gp->dinv, gp->Npad,
                           (long long)gp->Npad, 0.0, gp->ws, gp->work, 1);
        hipLaunchKernelGGL(retile_16_to_64_kernel, dim3(1024), dim3(256), 0, s, gp->ws, dst, gp->Npad);
        ALABI_LAUNCH_CHECK();
        return ALABI_OK;
    }
    hipLaunchKernelGGL((predict_var_kernel<1, true>), dim3(nb), dim3(256), 0, s, gp->L, gp->dinv, gp->Xt, gp->alpha, gp->N,
                       gp->Npad, (const double*)nullptr, gp->d, (long long)gp->Npad, gp->inv_len, 1.0, 0.0, dst,
                       (double*)nullptr, (double*)nullptr, 0, gp->kf);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

// ... into the variance workspace (the gradient's use)
int launch_factor_inverse(alabi_gp* gp, hipStream_t s) {
    const size_t need = (size_t)gp->Npad * gp->Npad * sizeof(double);
    if (need > gp->ws_bytes) {
        if (gp->ws) {
            ALABI_HIP_CHECK(hipStreamSynchronize(s));
            dev_cache_give(gp->ws, gp->ws_bytes);
            gp->ws = nullptr; gp->ws_bytes = 0;
        }
        size_t got = 0;
        if (dev_alloc_cached((void**)&gp->ws, need, &got) != (int)hipSuccess) return ALABI_HIP_ERROR;
        gp->ws_bytes = got;
    }
    return launch_factor_inverse_into(gp, gp->ws, s);
}

// Variance of at most 16 queries from the cached W = L^-1: v = W k*, var = amp - |v|^2.  One workgroup per block row kb
// forms v_kb = sum_{j <= kb} W[kb,j] K*_j on the matrix cores (the K* blocks are re-evaluated per workgroup: 64 x 16
// kernel values per block) and leaves its 16 partial sums of squares; a second tiny kernel adds them in block order.
template <int D, bool GENERIC>
__global__ void __launch_bounds__(256)
predict_var_small_kernel(const double* __restrict__ W, const double* __restrict__ Xt, int N, int Npad,
                         const double* __restrict__ Xs, int d, int M, DimVec inv_len, double amp, KernelFn kf,
                         double* __restrict__ partial) {
    __shared__ double Wt[64][66];
    __shared__ double Ks[64][18];
    __shared__ double qs[16][D];
    __shared__ double red[4][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int kb = blockIdx.x;
    // blockIdx.y selects a group of 16 queries (up to 512 queries go through this kernel: one round of nb x groups
    // workgroups costs less than the K* pre-pass of the tile kernels, which only fills M / 64 CUs)
    Xs += (size_t)blockIdx.y * 16 * d;
    M -= (int)blockIdx.y * 16;
    if (M > 16) M = 16;
    for (int e = tid; e < 16 * D; e += 256) {
        const int m = e / D, k = e % D;
        qs[m][k] = (m < M && k < d) ? Xs[(size_t)m * d + k] * inv_len.v[k] : 0.0;
    }
    v4f64 acc = v4f64{0.0, 0.0, 0.0, 0.0};
    // Block j + 1 (its W tile and the training coordinates of its 64 points) is requested while block j is multiplied: the
    // loop is a chain of kb + 1 dependent stages and a global round trip per stage would dominate it.  The request is
    // unconditional (the last block is simply asked for again) so that the waits stay counted.
    f64x2 wreg[8];
    double x[D];
#define ALABI_SMALL_REQUEST(J)                                                                               \
    {                                                                                                        \
        const f64x2* Wb_ = reinterpret_cast<const f64x2*>(W + (size_t)(J) * Npad * 64 + (size_t)(kb * 64) * 64); \
        _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) wreg[e_] = Wb_[tid + 256 * e_];                     \
        _Pragma("unroll") for (int k = 0; k < D; ++k) x[k] = Xt[(size_t)k * Npad + (J) * 64 + lane];         \
    }
    ALABI_SMALL_REQUEST(0)
    for (int j = 0; j <= kb; ++j) {
        __syncthreads();                                   // the previous block's MFMAs are done with Wt / Ks (and qs is set)
#pragma unroll
        for (int e_ = 0; e_ < 8; ++e_) {
            const int e = tid + 256 * e_;
            *reinterpret_cast<f64x2*>(&Wt[e >> 5][2 * (e & 31)]) = wreg[e_];
        }
        {   // K*_j: thread (point p = lane, queries 4w .. 4w+3)
            const int n = j * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 4 * w + i;
                double r2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) { const double df = x[k] - qs[m][k]; r2 = fma(df, df, r2); }
                Ks[lane][m] = (n < N && m < M) ? amp * radial<GENERIC>(r2, kf) : 0.0;
            }
        }
        const int jn = (j < kb) ? j + 1 : kb;
        ALABI_SMALL_REQUEST(jn)
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[16 * w + lr][4 * ks + lk], Ks[4 * ks + lk][lr], acc, 0, 0, 0);
    }
#undef ALABI_SMALL_REQUEST
    // acc[i] = v[row 16w + lk + 4i][query lr]: squares summed over the rows of this wave, then over the waves
    double ss = fma(acc[0], acc[0], fma(acc[1], acc[1], fma(acc[2], acc[2], acc[3] * acc[3])));
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) red[w][lane] = ss;
    __syncthreads();
    if (tid < 16) partial[((size_t)blockIdx.y * gridDim.x + kb) * 16 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ void __launch_bounds__(64)
predict_var_small_final_kernel(const double* __restrict__ partial, int nb, int M, double amp, double* __restrict__ var) {
    const int m = blockIdx.x * 64 + threadIdx.x;           // query; group m / 16, slot m % 16
    if (m >= M) return;
    const double* pg = partial + (size_t)(m >> 4) * nb * 16 + (m & 15);
    double s = 0.0;
    for (int kb = 0; kb < nb; ++kb) s += pg[(size_t)kb * 16];
    var[m] = amp - s;
}

// The cached W = L^-1 of the current factor (tile-major, Npad^2 doubles); ALABI_NOT_COMPUTED when there is no room for it.
int ensure_winv(alabi_gp* gp, hipStream_t s) {
    const size_t need = (size_t)gp->Npad * gp->Npad * sizeof(double);
    if (need > gp->winv_bytes) {
        if (gp->winv) {
            ALABI_HIP_CHECK(hipStreamSynchronize(s));
            dev_cache_give(gp->winv, gp->winv_bytes);
            gp->winv = nullptr; gp->winv_bytes = 0;
        }
        size_t got = 0;
        if (dev_alloc_cached((void**)&gp->winv, need, &got) != (int)hipSuccess) return ALABI_NOT_COMPUTED;
        gp->winv_bytes = got;
        gp->winv_gen = -1;
    }
    if (gp->winv_gen != gp->factor_gen) {
        int st = launch_factor_inverse_into(gp, gp->winv, s);
        if (st != ALABI_OK) return st;
        gp->winv_gen = gp->factor_gen;
    }
    return ALABI_OK;
}

static int ensure_small(alabi_gp* gp, size_t bytes, hipStream_t s) {
    if (gp->small_bytes >= bytes) return ALABI_OK;
    if (gp->small) {
        ALABI_HIP_CHECK(hipStreamSynchronize(s));
        ALABI_HIP_CHECK(hipFree(gp->small));
        gp->small = nullptr; gp->small_bytes = 0;
    }
    ALABI_HIP_CHECK(hipMalloc(&gp->small, bytes));
    gp->small_bytes = bytes;
    return ALABI_OK;
}

int launch_predict_var_small(alabi_gp* gp, const double* Xs, int M, double* mu, double* var, hipStream_t s) {
    const int nb = gp->Npad / 64, db = dim_bucket(gp->d);
    int st = ensure_winv(gp, s);
    if (st == ALABI_NOT_COMPUTED) return launch_predict_var(gp, Xs, M, mu, var, s);   // no room: substitution kernel
    if (st != ALABI_OK) return st;
    const int groups = (M + 15) / 16;
    if ((st = ensure_small(gp, (size_t)groups * nb * 16 * sizeof(double), s)) != ALABI_OK) return st;
    if ((st = launch_predict_mean(gp, Xs, M, mu, s)) != ALABI_OK) return st;
    const double amp = exp(gp->log_amp);
    ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(gp->kf.type, hipLaunchKernelGGL((predict_var_small_kernel<D, GENERIC>), dim3(nb, groups),
        dim3(256), 0, s, gp->winv, gp->Xt, gp->N, gp->Npad, Xs, gp->d, M, gp->inv_len, amp, gp->kf, gp->small)));
    hipLaunchKernelGGL(predict_var_small_final_kernel, dim3((M + 63) / 64), dim3(64), 0, s, gp->small, nb, M, amp, var);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
