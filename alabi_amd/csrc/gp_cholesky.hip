// Blocked right-looking Cholesky (lower, in place, fp64) for gfx950.
//
// Replaces scipy.linalg.cholesky inside george's BasicSolver.compute, reached from the
// reference at alabi/core.py:1158 (every active-learning iteration), :1430, :1577 and
// alabi/gp_utils.py:243.  Work: N^3/3 flops; the dense trailing update runs on the fp64
// matrix cores (v_mfma_f64_16x16x4_f64) with both 64x64 panels staged in LDS.
//
// Per 64-column block step kb (two launches; the first diagonal block has its own one-wave launch):
//   1. trsm_panel   : A[i,kb] <- A[i,kb] * L_kk^-T.  One lane owns one row: the substitution along the
//                     row is right-looking in registers (two dependent operations per column), L_kk is
//                     read from LDS as broadcasts, divisions are multiplications by dinv = 1/L_jj.
//   2. syrk_update  : A[i,j] -= A[i,kb] * A[j,kb]^T for kb < j <= i, one 64x64 tile per workgroup,
//                     4 waves x (16 rows x 64 cols) x K=64 on MFMA.  The workgroup of tile (kb+1,kb+1)
//                     then FACTORISES that tile while it is still in LDS (potrf_tile_lds: one wave,
//                     16-column slabs in registers, v_readlane broadcasts, rank-16 MFMA updates between
//                     slabs), so the diagonal factorisation costs no launch of its own.  A non-positive
//                     pivot is reported as LAPACK's potrf `info` (1-based).
// The matrix is [Npad, Npad] row-major with identity padding, so every block is full.
#include <cstdlib>
#include <array>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>
#include "gp_device.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// One fp64 MFMA rank-16 update of a 16x16 tile held in LDS:  C -= P Q^T, with P = rows pr.. and Q = rows qr.. of the
// same 16-column slab (columns c0..c0+15) of `M`.  One wavefront; lane l: A[m=l&15][k=l>>4], B[k=l>>4][n=l&15],
// C/D row (l>>4)+4i, column l&15.
template <int LD>
__device__ inline void tile_update_16(double (*C)[LD], int cr, int cc, double (*Pm)[LD], int pr, double (*Qm)[LD], int qr,
                                      int c0, int lane) {
    const int lr = lane & 15, lk = lane >> 4;
    v4f64 acc;
    double a[4], b[4];                                      // all twelve LDS reads in flight at once (one round trip, not five)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { a[kk] = Pm[pr + lr][c0 + 4 * kk + lk]; b[kk] = Qm[qr + lr][c0 + 4 * kk + lk]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = C[cr + lk + 4 * i][cc + lr];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(a[kk]), "+v"(b[kk]));
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[kk], b[kk], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) C[cr + lk + 4 * i][cc + lr] = acc[i];
}

// 1/sqrt(piv) from the hardware estimate r0 (relative error <= 5.2e-8, tools/micro/rsq_accuracy) and ONE third-order step:
// with e = 1 - piv r0^2, 1/sqrt(piv) = r0 (1 + e/2 + 3 e^2/8 + O(e^3)), the dropped term is ~1e-22; measured 1.4e-16.
// Four dependent operations after the estimate (two Newton steps are six); the slab recurrence scales its column by this
// value, so it sits on the critical chain of every pivot.
__device__ inline double pivot_rsqrt(double piv) {
    const double r0 = __builtin_amdgcn_rsq(piv);
    const double e = fma(-(piv * r0), r0, 1.0);
    return fma(r0 * e, fma(0.375, e, 0.5), r0);
}

// One 16-column slab of the diagonal block's factorisation; lane = row, a[j] = the row's entry in slab column j.  Right-
// looking: pivot j is broadcast from its lane, the column is scaled by 1/sqrt(pivot) in EVERY lane -- the diagonal lane
// thereby gets L_jj = piv / sqrt(piv) (1.1 ulp) without a select -- and the row's remaining slab columns take their rank-1
// update at once, L[c0+k][c0+j] arriving by v_readlane from the lane that owns row c0+k.  The wave runs one instruction per
// ~5 cycles whatever its kind, so the slab costs what it issues: nothing per pivot but the chain itself -- no branch, no
// diagonal select, no bookkeeping of the reciprocals (potrf_dinv forms them from the finished diagonal) and no test of the
// pivot: a non-positive or non-finite pivot turns its own and every later column into NaN (rsq of it is NaN or inf, 0 * inf
// = NaN), the earlier columns stay finite, so the FIRST diagonal entry that is not > 0 afterwards is LAPACK's `info`
// (potrf_first_bad).
// (Round 3, measured and not kept: (i) the multipliers L[c0+k][c0+j], k >= j + 2, as uniform-address LDS reads of the just-scaled
// column instead of v_readlane pairs, with the reciprocal square root of pivot j + 1 interleaved by hand with the updates of
// pivot j -- bit-identical, no faster; (ii) the trailing columns updated from the UNSCALED column and 1 / pivot, which shortens
// the dependent chain from pivot to pivot from two cross-lane hops + eight operations to one hop + six but adds four
// instructions per pivot -- 7.6 -> 8.0 us for the 64-pivot factorisation.  Data-dependent s_memrealtime stamps (ALABI_CHOL_PROF)
// then put wave 0's slab recurrence at 1.1-1.2 us per 16 pivots = 172 cycles per pivot for ~31 instructions: a lone wave issues
// one instruction per ~5.5 cycles and the recurrence is bound by that COUNT, as the round-2 text says; the four recurrences are
// 4.6 of the factorisation's 7.5 us, the rank-16 updates between them, their barriers and the slab's LDS traffic the rest.
// (iii) EIGHT-column slabs in the task queue's diagonal factorisation (38 % fewer recurrence instructions: 8 x 156 instead of
// 4 x 504; rank-8 tile updates on all waves, the tile column that holds the slab written back in its second half only):
// the recurrences fell from 5.2 to 3.6 us and the seven instead of three slab boundaries (two barriers + an LDS round trip +
// two dependent matrix-core instructions each, ~0.45 us) took it back -- N = 2000 0.527 vs 0.527 ms on one box.)
__device__ inline void potrf_slab(double (&a)[16], int c0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double rinv = pivot_rsqrt(lane_bcast(a[j], c0 + j));
        a[j] *= rinv;
        double bc[16];                                      // all broadcasts of the column first: distinct SGPR pairs, so no
#pragma unroll                                              // readlane -> use wait states between them and the FMAs
        for (int k = j + 1; k < 16; ++k) bc[k] = lane_bcast(a[j], c0 + k);
#pragma unroll
        for (int k = j + 1; k < 16; ++k) a[k] = fma(-a[j], bc[k], a[k]);
    }
}

// 1-based index of the first diagonal entry of the finished tile that is not > 0 (NaN included), 0 if there is none; `lll` =
// L[lane][lane], one full wave.
__device__ inline int potrf_first_bad(double lll) {
    const unsigned long long m = __ballot(!(lll > 0.0));
    return m ? __ffsll((long long)m) : 0;
}

// 1 / L_ll for the row of `lane` from the finished diagonal: hardware reciprocal (4.5e-8) + two Newton steps.
__device__ inline double potrf_dinv(double lll) {
    double r = __builtin_amdgcn_rcp(lll);
    r = fma(fma(-lll, r, 1.0), r, r);
    r = fma(fma(-lll, r, 1.0), r, r);
    return r;
}

// Diagonal block held in LDS (row stride LD doubles), factorised in place by ONE wavefront in 16-column slabs.  Inside a
// slab every lane (= row) keeps its 16 entries in registers and the recurrence is right-looking: after pivot j the row's
// remaining slab columns take their rank-1 update at once, L[c0+k][c0+j] arriving by v_readlane from the lane that owns
// row c0+k, so the dependent chain per column is readlane -> rsqrt/Newton -> scale -> readlane -> one FMA and the other
// updates fill its shadow.  After a slab the trailing tiles get its rank-16 update on the matrix cores.  A non-positive
// pivot is reported as LAPACK's potrf `info` (1-based).  Returns 1/L_ii of row `lane`.  `__syncthreads` here is executed
// by one wave only when the caller's other waves wait at a later barrier, so plain wave-level ordering is used instead.
template <int LD>
__device__ inline double potrf_tile_lds(double (*Ls)[LD], int lane, int kb, int* __restrict__ info) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c0 = 16 * s;
        double a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = Ls[lane][c0 + j];      // row `lane`, this slab (rows < c0 carry unused values)
        potrf_slab(a, c0);
#pragma unroll
        for (int j = 0; j < 16; ++j) Ls[lane][c0 + j] = a[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // one wave: LDS writes above are ordered before the reads below
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // rank-16 update of the tiles right of / below the slab (lower triangle of the 16x16 tile grid)
#pragma unroll
        for (int ti = s + 1; ti < 4; ++ti)
#pragma unroll
            for (int tk = s + 1; tk <= ti; ++tk) tile_update_16<LD>(Ls, 16 * ti, 16 * tk, Ls, 16 * ti, Ls, 16 * tk, c0, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const double lll = Ls[lane][lane];
    const int bad = potrf_first_bad(lll);
    if (bad != 0 && lane == 0) atomicCAS(info, 0, kb * 64 + bad);
    return potrf_dinv(lll);
}

// The same factorisation by a whole 256-thread workgroup: wave 0 runs the slab recurrences, the rank-16 tile updates
// between slabs (6, 3, 1 tiles) are dealt to the four waves.  Every thread must call it; returns 1/L_ii in wave 0.
template <int LD>
__device__ inline double potrf_tile_lds_wg(double (*Ls)[LD], int tid, int kb, int* __restrict__ info) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c0 = 16 * s;
        if (w == 0) {
            double a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = Ls[lane][c0 + j];
            potrf_slab(a, c0);
#pragma unroll
            for (int j = 0; j < 16; ++j) Ls[lane][c0 + j] = a[j];
        }
        if (s == 3) break;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int ti = s + 1; ti < 4; ++ti)
#pragma unroll
            for (int tk = s + 1; tk <= ti; ++tk, ++t)
                if ((t & 3) == w) tile_update_16<LD>(Ls, 16 * ti, 16 * tk, Ls, 16 * ti, Ls, 16 * tk, c0, lane);
        __syncthreads();
    }
    __syncthreads();
    if (w != 0) return 1.0;
    const double lll = Ls[lane][lane];
    const int bad = potrf_first_bad(lll);
    if (bad != 0 && lane == 0) atomicCAS(info, 0, kb * 64 + bad);
    return potrf_dinv(lll);
}

// First diagonal block (the later ones are factorised inside syrk_update_kernel by the workgroup that finishes them).
__global__ void __launch_bounds__(64)
potrf_diag_kernel(double* __restrict__ A, int ld, int kb, int* __restrict__ info, double* __restrict__ dinv) {
    __shared__ double Ls[64][65];
    const int lane = threadIdx.x;
    double* Ab = A + (size_t)(kb * 64) * ld + kb * 64;
#pragma unroll
    for (int r = 0; r < 64; ++r) Ls[r][lane] = Ab[(size_t)r * ld + lane];   // coalesced rows, all 64 loads in flight
    __syncthreads();
    dinv[kb * 64 + lane] = potrf_tile_lds<65>(Ls, lane, kb, info);
#pragma unroll
    for (int r = 0; r < 64; ++r)
        if (lane <= r) Ab[(size_t)r * ld + lane] = Ls[r][lane];
}

// One row's recurrence over a 16-column slab of X L_kk^T = B (right-looking along the row: two dependent operations per
// column -- scale, first update -- and the other updates fill their shadow).  The 120 strictly-lower entries of the slab's
// triangle and the 16 reciprocals are wave-uniform LDS broadcasts; they are fetched in four column groups (3, 3, 4, 6
// columns: 42, 33, 30, 15 entries), each while the group before it is being applied, so the chain never waits for LDS and at
// most two groups are live: ~215 registers instead of 364 for fetching all 120 up front.  That matters beyond this kernel:
// a workgroup of the panel chain has to fit into the hole one retired workgroup of the bulk trailing update leaves on a SIMD
// (512 - 232 registers), or the chain cannot overlap that update at all.  sched_barrier keeps the compiler from sinking a
// group's reads next to their uses.
template <int J0, int J1, int LD>
__device__ inline void trsm_group_fetch(double (*lkk)[LD], const double* di, int c0, double* lg, double* dg) {
    int q = 0;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        dg[j - J0] = di[c0 + j];
#pragma unroll
        for (int k = j + 1; k < 16; ++k) lg[q++] = lkk[c0 + k][c0 + j];
    }
}
template <int J0, int J1>
__device__ inline void trsm_group_apply(double* b, const double* lg, const double* dg) {
    int q = 0;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        b[j] *= dg[j - J0];
#pragma unroll
        for (int k = j + 1; k < 16; ++k) b[k] = fma(-b[j], lg[q++], b[k]);
    }
}
template <int LD>
__device__ inline void trsm_slab_row(double (*lkk)[LD], double (*bs)[LD], const double* di, int row, int c0) {
    double b[16], l0[42], l1[33], l2[30], l3[15], d0[3], d1[3], d2[4], d3[6];
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = bs[row][c0 + j];
    trsm_group_fetch<0, 3, LD>(lkk, di, c0, l0, d0);
    __builtin_amdgcn_sched_barrier(0);
    trsm_group_fetch<3, 6, LD>(lkk, di, c0, l1, d1);
    trsm_group_apply<0, 3>(b, l0, d0);
    __builtin_amdgcn_sched_barrier(0);
    trsm_group_fetch<6, 10, LD>(lkk, di, c0, l2, d2);
    trsm_group_apply<3, 6>(b, l1, d1);
    __builtin_amdgcn_sched_barrier(0);
    trsm_group_fetch<10, 16, LD>(lkk, di, c0, l3, d3);
    trsm_group_apply<6, 10>(b, l2, d2);
    __builtin_amdgcn_sched_barrier(0);
    trsm_group_apply<10, 16>(b, l3, d3);
#pragma unroll
    for (int j = 0; j < 16; ++j) bs[row][c0 + j] = b[j];
}

// X * L_kk^T = B for 64 rows; four wavefronts, wave w owns rows 16w..16w+15.  16-column slabs: inside a slab one lane
// runs one row's recurrence (trsm_slab_row: L_kk read from LDS as broadcasts, divisions are multiplications by dinv, no
// cross-lane traffic); the slab's effect on the remaining columns is a rank-16 update on the matrix cores, one row tile per
// wave.
__global__ void __launch_bounds__(256)
trsm_panel_kernel(double* __restrict__ A, int ld, int kb, const double* __restrict__ dinv) {
    __shared__ double lkk[64][65];
    __shared__ double bs[64][65];
    __shared__ double di[64];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const double* Lb = A + (size_t)(kb * 64) * ld + kb * 64;
    double* Bb = A + (size_t)((kb + 1 + blockIdx.x) * 64) * ld + kb * 64;
    #pragma unroll
    for (int e_ = 0; e_ < 16; ++e_) {
        const int e = tid + 256 * e_;
        const int r = e >> 6, c = e & 63;
        lkk[r][c] = Lb[(size_t)r * ld + c];
        bs[r][c] = Bb[(size_t)r * ld + c];
    }
    if (tid < 64) di[tid] = dinv[kb * 64 + tid];
    __syncthreads();
    const int row = 16 * w + (lane & 15);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c0 = 16 * s;
        if (lane < 16) trsm_slab_row<65>(lkk, bs, di, row, c0);
        __syncthreads();
        // B[rows of this wave, later slabs] -= X_s * L_kk[later rows, slab]^T
#pragma unroll
        for (int t = s + 1; t < 4; ++t) tile_update_16<65>(bs, 16 * w, 16 * t, bs, 16 * w, lkk, 16 * t, c0, lane);
        __syncthreads();
    }
    #pragma unroll
    for (int e_ = 0; e_ < 16; ++e_) {
        const int e = tid + 256 * e_;
        const int r = e >> 6, c = e & 63;
        Bb[(size_t)r * ld + c] = bs[r][c];
    }
}

// C[bi,bj] -= P[bi] * P[bj]^T with P[b] = A[b-block rows, kb-block cols].  Workgroup 0 owns the tile (kb+1, kb+1), which is
// complete after this update: it factorises it on the spot (one wave, potrf_tile_lds), so the next block step starts with
// its panel solve and the diagonal factorisation costs no launch, no reload and overlaps the other tiles' updates.
// jc > 0 restricts the update to the first jc block columns of the trailing matrix (the rest of a 256-column panel; the
// columns beyond it receive the whole panel at once from syrk_panel_kernel): tiles are then numbered column by column.
__global__ void __launch_bounds__(256)
syrk_update_kernel(double* __restrict__ A, int ld, int kb, int* __restrict__ info, double* __restrict__ dinv, int jc, int T) {
    __shared__ double Pi[64][66];
    __shared__ double Pj[64][66];
    int t = blockIdx.x;
    int ti, tj;
    if (jc > 0) {
        tj = 0;
        int off = 0;
        while (tj + 1 < jc && t >= off + (T - tj)) { off += T - tj; ++tj; }
        ti = tj + (t - off);
    } else {
        ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        tj = t - ti * (ti + 1) / 2;
    }
    const int bi = kb + 1 + ti, bj = kb + 1 + tj;
    const int tid = threadIdx.x;
    const double* Ai = A + (size_t)(bi * 64) * ld + kb * 64;
    const double* Aj = A + (size_t)(bj * 64) * ld + kb * 64;
    #pragma unroll
    for (int e_ = 0; e_ < 16; ++e_) {
        const int e = tid + 256 * e_;
        int r = e >> 6, c = e & 63;
        Pi[r][c] = Ai[(size_t)r * ld + c];
        Pj[r][c] = Aj[(size_t)r * ld + c];
    }
    const int w = tid >> 6, l = tid & 63;
    const int lr = l & 15, lk = l >> 4;
    double* C = A + (size_t)(bi * 64 + 16 * w) * ld + bj * 64;
    v4f64 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[n][i] = C[(size_t)(lk + 4 * i) * ld + 16 * n + lr];
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        double a = -Pi[16 * w + lr][4 * ks + lk];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            double b = Pj[16 * n + lr][4 * ks + lk];
            acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
        }
    }
    if (t == 0) {
        __syncthreads();                                   // every wave is done reading Pi / Pj
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) Pi[16 * w + lk + 4 * i][16 * n + lr] = acc[n][i];
        __syncthreads();
        const double rinv = potrf_tile_lds_wg<66>(Pi, tid, bi, info);
        if (w == 0) dinv[bi * 64 + l] = rinv;
        double* D = A + (size_t)(bi * 64) * ld + bi * 64;
#pragma unroll
        for (int r0_ = 0; r0_ < 64; r0_ += 4) {
            const int r = r0_ + w;
            if (l <= r) D[(size_t)r * ld + l] = Pi[r][l];
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) C[(size_t)(lk + 4 * i) * ld + 16 * n + lr] = acc[n][i];
}

// Trailing update with a whole 256-column panel: C[I,J] -= P_I P_J^T for 128 x 128 tiles (I >= J) of the matrix behind the
// panel, P_X = A[rows of tile X, panel columns] (128 x KP, KP = 64 * width of the panel in blocks <= 256).  Rank-64 updates
// re-read and re-write every trailing tile once per block column (5-7x the algorithmic traffic); here a
// tile is read and written ONCE per four block columns and each staged panel byte feeds a 128-wide output: 11 flop per
// byte of L2 traffic instead of 4.  Four wavefronts, each owns a 64 x 64 quadrant = 4 x 4 accumulator tiles of
// v_mfma_f64_16x16x4 (128 VGPRs); the panel rows are staged 16 columns at a time through two LDS buffers (row stride 18
// doubles: conflict-free ds_read_b64 for the fragment layout), the next slice is fetched into registers while the current
// one is multiplied; one barrier per slice.  Tile columns [tc0, tc1) of the trailing matrix are processed (look-ahead: the
// columns of the next panel on the main stream, the rest on a second stream).
// FUSE_POTRF = false is the bulk instantiation (the look-ahead remainder): without the factorisation code it stays below
// 256 registers, so TWO workgroups share a CU and one's barriers and fetch waits hide behind the other's MFMAs (with the
// fused code in the same kernel the allocation was 392 registers: one wave per SIMD, 48 % matrix-core occupancy,
// profiles/r02_cholesky_n10000_timeline.txt).  Tiles that lie wholly inside the matrix (all but the last tile row) take
// loads and stores without per-element guards; a wave whose 64 x 64 quadrant is the upper block of a diagonal tile idles.
// The accumulators hold -C, so the products are added as they come and the sign is restored with the store.
template <bool FUSE_POTRF>
__global__ void __launch_bounds__(256, FUSE_POTRF ? 1 : 2)
syrk_panel_kernel(double* __restrict__ A, int ld, int n, int col0, int kp, int row0, int tc0, int tc1, int ntr,
                  int* __restrict__ info, double* __restrict__ dinv) {
    __shared__ __attribute__((aligned(16))) double Ps[2][2][128][18];   // [buffer][I / J][row][k]
    // tile (ti, tj), tc0 <= tj < tc1, tj <= ti < ntr, numbered column by column
    int t = blockIdx.x, tj = tc0, off = 0;
    while (tj + 1 < tc1 && t >= off + (ntr - tj)) { off += ntr - tj; ++tj; }
    const int ti = tj + (t - off);
    const int ri = row0 + 128 * ti, rj = row0 + 128 * tj;            // first row of P_I / P_J (= first column of the C tile)
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    const bool full = ri + 128 <= n;                                 // rj <= ri: the tile needs no row / column guards
    const bool quad = ti > tj || wc <= wr;                           // this wave's quadrant is part of the lower block triangle
    // staging map: thread -> (row sr + 32 pass, 2 consecutive k), 8 threads per 128-byte row slice
    const int sr = tid >> 3, sk = (tid & 7) * 2;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 gi[4], gj[4];
    const double* pI = A + (size_t)(ri + sr) * ld + col0 + sk;
    const double* pJ = A + (size_t)(rj + sr) * ld + col0 + sk;
    const size_t rs = (size_t)32 * ld;
    auto fetch = [&](int k0) {
        if (full) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                gi[ps] = *reinterpret_cast<const f64x2*>(pI + ps * rs + k0);
                gj[ps] = *reinterpret_cast<const f64x2*>(pJ + ps * rs + k0);
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int r = sr + 32 * ps;
                gi[ps] = (ri + r < n) ? *reinterpret_cast<const f64x2*>(pI + ps * rs + k0) : f64x2{0.0, 0.0};
                gj[ps] = (rj + r < n) ? *reinterpret_cast<const f64x2*>(pJ + ps * rs + k0) : f64x2{0.0, 0.0};
            }
        }
    };
    auto stage = [&](int b) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = sr + 32 * ps;
            *reinterpret_cast<f64x2*>(&Ps[b][0][r][sk]) = gi[ps];
            *reinterpret_cast<f64x2*>(&Ps[b][1][r][sk]) = gj[ps];
        }
    };
    fetch(0);
    // accumulators start from -C (rows ri + 64 wr + 16 m + lk + 4 i, columns rj + 64 wc + 16 nn + lr)
    double* Cw = A + (size_t)(ri + 64 * wr + lk) * ld + rj + 64 * wc + lr;
    v4f64 acc[4][4];
    if (full && quad) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double* row = Cw + (size_t)(16 * m + 4 * i) * ld;
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) acc[m][nn][i] = -row[16 * nn];
            }
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = ri + 64 * wr + 16 * m + lk + 4 * i, c = rj + 64 * wc + 16 * nn + lr;
                    acc[m][nn][i] = (quad && r < n && c < n) ? -A[(size_t)r * ld + c] : 0.0;
                }
    }
    stage(0);
    __syncthreads();
    const int nslices = kp / 16;
    for (int sl = 0; sl < nslices; ++sl) {
        const int b = sl & 1;
        if (sl + 1 < nslices) fetch(16 * (sl + 1));
        if (quad) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double a[4], bb[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = Ps[b][0][64 * wr + 16 * m + lr][4 * kk + lk];
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) bb[nn] = Ps[b][1][64 * wc + 16 * nn + lr][4 * kk + lk];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bb[nn], acc[m][nn], 0, 0, 0);
            }
        }
        if (sl + 1 < nslices) stage(b ^ 1);
        __syncthreads();
    }
    // The first tile holds the diagonal block of the NEXT panel in the quadrant of wave 0: it is complete now, so it is
    // factorised on the spot (potrf_tile_lds_wg, all four waves) instead of by a launch of its own.
    if (FUSE_POTRF && t == 0 && tc0 == 0 && dinv) {
        double (*Ls)[66] = reinterpret_cast<double (*)[66]>(&Ps[0][0][0][0]);     // 64 x 66 doubles: fits the staging buffers
        if (w == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                    for (int i = 0; i < 4; ++i) Ls[16 * m + lk + 4 * i][16 * nn + lr] = -acc[m][nn][i];
        }
        __syncthreads();
        const double rinv = potrf_tile_lds_wg<66>(Ls, tid, row0 / 64, info);
        if (w == 0) {
            dinv[row0 + l] = rinv;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[m][nn][i] = -Ls[16 * m + lk + 4 * i][16 * nn + lr];
        }
    }
    // only block columns <= block rows belong to the factorisation: the upper 64-block of a diagonal tile is left alone
    if (full && quad) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double* row = Cw + (size_t)(16 * m + 4 * i) * ld;
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) row[16 * nn] = -acc[m][nn][i];
            }
    } else if (quad) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = ri + 64 * wr + 16 * m + lk + 4 * i, c = rj + 64 * wc + 16 * nn + lr;
                    if (r < n && c < n) A[(size_t)r * ld + c] = -acc[m][nn][i];
                }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Task-queue factorisation for up to 256 block columns (default for N = 961..16384): ONE launch instead of 2 nb - 1.
// The launch-per-step path above is a chain of dependent kernels: per block column a panel solve (11 us) and an update with
// the next diagonal factorisation fused in (19.6 us), each behind a kernel boundary -- 1.02 ms at N = 2000 for 2.67 GFLOP.
// Here the same 64 x 64 tile operations are TASKS in a static topological order; persistent workgroups draw the next task
// index from one atomic counter, wait (bounded) until the tile versions it depends on have been published, run it and
// publish its own tile version.  A workgroup only ever waits for tasks with a smaller index, and every drawn task is held by
// a running workgroup, so the queue cannot deadlock even when not all workgroups are resident.  Hand-off between workgroups:
// tiles are written with write-through (sc1) stores, every wave drains its stores, one barrier, then ONE lane publishes
// the tile's version with an sc1 store; readers poll the version words and read the tiles with sc1 loads
// (cdna_hip_programming.md Guideline 16, R1 with sc1 loads in place of the acquire).
//   CHAIN(k)      k >= 1: solve tile (k, k-1) against L[k-1,k-1], publish it, apply it to tile (k, k) and factorise that
//                 tile on the spot -- the whole critical path of a block column in ONE workgroup without leaving LDS;
//                 CHAIN(0) factorises tile (0, 0).
//   TRSM(i, k)    i >= k + 2: the other tiles of the panel.
//   UPDATE(i,j,k) tile (i, j) -= tile (i, k) tile (j, k)^T for i >= j > k except (k+1, k+1).
// Order per block column k: CHAIN(k+1) first, then the panel solves, then the updates of column k+1 (the next chain's
// inputs), then the rest -- the chain never queues behind bulk updates.  ver[i][j] = number of steps applied to tile (i, j);
// j + 1 means final.
// type & 255: 0 CHAIN(k), 1 TRSM(i,k), 2 UPDATE(i,j,k..k+cnt-1) with cnt = type >> 8 consecutive block columns (chol_build_tasks),
//             4 UPDATE2 = UPDATE(i,j,..) and UPDATE(i+1,j,..) in one task (eight-wave kernel), 5 UPDATE4 = the 2 x 2 block of tiles
//             (i,j), (i+1,j), (i,j+1), (i+1,j+1), i >= j + 1
struct CholTask { int type, i, j, k; };
// Every coherent load / store of the queue names the GLOBAL address space: inside the non-inlined phase functions, and in the batched
// kernel (whose matrix pointers are loaded from a table), the pointers are generic to the compiler and the accesses became flat_load /
// flat_store -- the slab stores of the diagonal factorisation took 0.1 us each.
typedef __attribute__((address_space(1))) unsigned long long* ct_gptr64;
typedef __attribute__((address_space(1))) int* ct_gptr32;
__device__ inline ct_gptr64 ct_g64(const double* p) { return (ct_gptr64)(unsigned long long*)const_cast<double*>(p); }
__device__ inline ct_gptr32 ct_g32(const int* p) { return (ct_gptr32)const_cast<int*>(p); }
// ... and the tiles, slabs and inverse blocks that are handed on go out in 16-byte pieces: a coherent (write-through) store is one fabric
// write per lane, and an 8-byte one costs 2.7x the time per byte of a 16-byte one (MI355X_MICROARCH.md).  A 64-row block at `base` with
// row stride ld as a buffer: (row, column) -> byte offset.
typedef unsigned int ct_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ct_u32x2 __attribute__((ext_vector_type(2)));
__device__ inline __amdgpu_buffer_rsrc_t ct_block_rsrc(double* base, int ld) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (unsigned)(63 * ld + 64) * 8u, 0x00020000);
}
// the pair (r, c), (r, c + 1) of a block, c even; lower = only what lies on or below the diagonal of the block
__device__ inline void ct_store_pair(__amdgpu_buffer_rsrc_t rs, int ld, int r, int c, ct_u32x4 v, bool lower) {
    const unsigned off = (unsigned)(r * ld + c) * 8u;
    if (!lower || c + 1 <= r) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
    else if (c == r) { ct_u32x2 h = {v.x, v.y}; __builtin_amdgcn_raw_buffer_store_b64(h, rs, off, 0, 16); }
}
// Batched queue (chol_tasks8_batch_kernel): bits 16.. of `type` say which matrix of the batch the task belongs to; the matrices are
// independent, each with its own tile versions, slab counters, status word and reciprocal diagonal.
struct CholMat { double* A; double* dinv; double* linv; int* ver; int* sver; int* info; int ld, nb; };
#define ALABI_CHOL_TASKS_MAX_NB 256   // default upper end of the one-launch task queue (N <= 16384); beyond: panels of 8 block columns
#define ALABI_CHOL_W8_MIN_NB 3    // block columns from which the queue runs eight waves per workgroup (chol_tasks8_kernel): every size it takes
#define ALABI_CHOL_UPDATE4_MIN_NB 100 // block columns from which the far updates take 2 x 2 tiles per task (UPDATE4; below: UPDATE2)
#define ALABI_CHOL_PLAIN_MIN 4   // block columns per UPDATE from which its operands are read with ordinary loads behind one acquire


template <int NT>
__device__ inline void tile_load_sc1(double (*T)[66], const double* __restrict__ src, int ld, int tid) {
#pragma unroll
    for (int e_ = 0; e_ < 4096 / NT; ++e_) {
        const int e = tid + NT * e_, r = e >> 6, c = e & 63;
        T[r][c] = __longlong_as_double((long long)__hip_atomic_load(ct_g64(src + (size_t)r * ld + c),
                                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
}
template <int NT>
__device__ inline void tile_store_sc1(double* __restrict__ dst, int ld, double (*T)[66], int tid, bool lower_only) {
    // every LDS read first, unconditionally, then the stores: written as "if (lower) store(T[r][c])" the compiler reads, waits and stores
    // element by element under the predicate -- 16 LDS round trips in a row
    const __amdgpu_buffer_rsrc_t rs = ct_block_rsrc(dst, ld);
    ct_u32x4 v[2048 / NT];
#pragma unroll
    for (int e_ = 0; e_ < 2048 / NT; ++e_) {
        const int e = tid + NT * e_;
        v[e_] = *reinterpret_cast<const ct_u32x4*>(&T[e >> 5][2 * (e & 31)]);
    }
#pragma unroll
    for (int e_ = 0; e_ < 2048 / NT; ++e_) asm volatile("" : "+v"(v[e_]));
#pragma unroll
    for (int e_ = 0; e_ < 2048 / NT; ++e_) {
        const int e = tid + NT * e_;
        ct_store_pair(rs, ld, e >> 5, 2 * (e & 31), v[e_], lower_only);
    }
}
// every wave has drained its stores and passed the barrier before ONE lane publishes the version
__device__ inline void publish_version(int* ver, int value, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(ct_g32(ver), value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Module-scope LDS, named directly by the non-inlined phase functions (as pointer arguments they would degrade to generic
// pointers).  The panel solve and the diagonal factorisation are separate noinline functions: inlined into the task loop their
// live ranges merge with the loop's and the serial recurrences fill up with AGPR moves (8.6 / 13.1 us instead of 5 / 9).
__shared__ double ct_pool[8 * 64 * 34];                               // one array: four 64 x 66 tiles, or (UPDATE2 / UPDATE4) six / eight 64 x 34 half tiles
#define ct_T0 (reinterpret_cast<double (*)[66]>(ct_pool))
#define ct_T1 (reinterpret_cast<double (*)[66]>(ct_pool + 64 * 66))
#define ct_T2 (reinterpret_cast<double (*)[66]>(ct_pool + 2 * 64 * 66))   // CHAIN: the diagonal tile, parked while the panel tile is solved
#define ct_T3 (reinterpret_cast<double (*)[66]>(ct_pool + 3 * 64 * 66))   // UPDATE over several block columns: second operand pair (T2, T3)
__shared__ int ct_task_s[16];                                          // the task loop's words (chol_tasks_body) + [9]: slabs of L[kk,kk] seen by a solve
#ifdef ALABI_CHOL_LOG
// Event log of the CHAIN tasks (tools/run_chol_log.sh): 10-ns time stamps written with plain stores by thread 0 -- no read-modify-write on
// the chain, unlike the ALABI_CHOL_PROF counters.  [k][0] drawn, [1] dependencies met, [2] tiles in LDS, [3..6] slab s of L[k-1,k-1] seen,
// [7] solve + diagonal update done, [8] panel tile published, [9] factorisation starts, [10..13] slab recurrence s done, [14] last inverse
// block out, [15] tile stored and published.
__device__ long long g_chain_log[256][32];   // [16 + 2 s] / [17 + 2 s]: factorisation past barrier A / B of slab s
__shared__ int ct_log_kb;
#define CT_LOG(i) do { if (threadIdx.x == 0) g_chain_log[ct_log_kb][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CT_LOGW(i) do { if ((threadIdx.x & 63) == 0) g_chain_log[ct_log_kb][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CT_LOGW(i) do { } while (0)
#define CT_LOG(i) do { } while (0)
#endif
// Lower-triangle tile j of the 4 x 4 grid of 16 x 16 tiles of a diagonal tile: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) .. (3,3)
__device__ inline void ct_diag_tile(int j, int& rt, int& ct) { rt = j >= 6 ? 3 : j >= 3 ? 2 : j >= 1 ? 1 : 0; ct = j - rt * (rt + 1) / 2; }
// What a panel solve needs of L[kk,kk] travels through the column's SLAB BUFFER, sbuf[4][64][16] doubles (32 KB per block column, beside the
// matrix): slab s holds, in rows 16 s .. 16 s + 15, the INVERSE of the slab's diagonal block and below them the slab's columns of L (rows above
// are unused) -- contiguous, in 16-byte pieces, piece e of a slab = row e >> 3, columns 2 (e & 7) ..  The solve never reads L[s,s] itself; in
// ct_T0 the inverse stands in its place.
#define ALABI_CHOL_TAG 0x7FF8DEADu   // both 32-bit halves of a "not written yet" double of the slab buffer: a NaN no arithmetic produces
template <int NT>
__device__ inline ct_u32x4 ct_slab_piece(__amdgpu_buffer_rsrc_t rs, int q, int e) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(q * 1024 + 2 * e) * 8u, 0, 16);
}
__device__ inline void ct_slab_piece_put(int q, int e, ct_u32x4 v) { *reinterpret_cast<ct_u32x4*>(&ct_T0[e >> 3][16 * q + 2 * (e & 7)]) = v; }
// the slabs [s0, s1) into ct_T0 in ONE memory round trip (a coherent load takes 0.7-1 us, whatever it fetches); run-time bounds: one copy of the code
template <int NT>
__device__ inline void ct_fetch_slabs(const double* __restrict__ sbuf, int tid, int s0, int s1) {
    constexpr int NE = 512 / NT;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(sbuf), 0, 32768u, 0x00020000);
    ct_u32x4 v[4][NE];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q >= s0 && q < s1) {
#pragma unroll
            for (int e_ = 0; e_ < NE; ++e_) {
                const int e = tid + NT * e_;
                if ((e >> 3) >= 16 * q) v[q][e_] = ct_slab_piece<NT>(rs, q, e);
            }
        }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q >= s0 && q < s1) {
#pragma unroll
            for (int e_ = 0; e_ < NE; ++e_) {
                const int e = tid + NT * e_;
                if ((e >> 3) >= 16 * q) ct_slab_piece_put(q, e, v[q][e_]);
            }
        }
}
// Panel solve X L_kk^T = B of the tile in ct_T1 ENTIRELY ON THE MATRIX CORES (round 4; before: a 16-step recurrence per slab in one
// wave, 1.8 us per slab, 7.2 us per tile -- 30 % of the workgroup time of a batch of N = 1600 matrices and the tail of every CHAIN).
// The diagonal factorisation publishes, per 16-column slab s, the inverse of the slab's 16 x 16 diagonal block (ct_potrf_publish:
// wave 1 runs the slab's recurrence on the block's rows and on the rows of the identity beside wave 0, so the inverse costs the
// chain nothing); with it
//     X_s = (B_s - sum_{u<s} X_u L_su^T) inv(L_ss)^T.
// Wave w (< 4) owns rows 16 w .. of the tile and works on the TRANSPOSE, Y = X^T: Y_s = inv(L_ss) (B_s^T - sum_u L_su Y_u).  Then the
// result of a product (C/D layout: row 4 i + (lane >> 4), column lane & 15) is, register i for k-step i, exactly the B operand of the
// next one (B[k = lane >> 4][n = lane & 15]), so the four slab steps chain in registers: 4 + 4 (3 - s) matrix-core instructions per slab
// and wave, 40 per tile = 1.1 us, no cross-lane traffic and no barrier between the slabs.  The slabs of L[kk,kk] are taken as they are
// published (sver[kk] = slabs available; all that are there in ONE fetch when the tile is final).  Error of a slab: that of a product
// with the explicit inverse of a 16 x 16 block, eps cond(L_ss) -- the blocks are small, tests hold ||L L^T - K|| <= 1e-12 ||K||.
// (Measured and not kept: the LAST inverse block polled itself -- pre-filled with a tag by the assembly kernel, valid once it differs -- instead
// of through the slab counter, one memory round trip instead of three behind the producer's last store: N = 2000 0.492 vs 0.495 ms.  The event
// log (ALABI_CHOL_LOG) shows why: the next CHAIN task gets its own tiles only 3 us before the previous factorisation ends -- they come from
// the single-column updates behind the previous panel solve -- and then works through the slabs at two round trips each, poll and fetch:
// slab 2 is in LDS 2 us AFTER that end, whatever the last block does.)
// DIAG (CHAIN): tile (k,k), parked in ct_T2, takes - X X^T slab by slab behind the solve (its ten lower 16 x 16 tiles dealt to all
// waves, accumulators in registers) and ends up in ct_T0 for the factorisation; the solved tile is written to Xdst while the last
// of that runs.  Returns false when a wait ran out (err set, every thread leaves).
// TAG (single matrix): no counter is polled at all.  The assembly kernel fills the slab buffers with a tag; the solve requests ALL FOUR slabs
// at once when it starts, and a slab counts as there when none of its pieces carries the tag any more (8 bytes at a time; a piece that
// does is requested again) -- one memory round trip behind the producer's stores instead of three (drain + counter, poll, fetch), and the
// slabs that were there already cost no round trip of their own.  (The event log, ALABI_CHOL_LOG, had shown the chain's next step getting its
// own tiles only 3 us before the previous factorisation ended, and then working through the slabs at two round trips each: the last slab was in
// LDS 4 us after that end.)  !TAG (batch): sver[kk] = slabs published so far, polled; everything that is there fetched in one round trip.
template <int NT, bool DIAG, bool TAG>
__device__ __attribute__((noinline)) bool ct_solve(int ld, const double* __restrict__ sbuf, int* sver, int* err,
                                                   int spin_limit, int ntasks, double* __restrict__ Xdst) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
    constexpr int NW = NT / 64, NQ = (10 + NW - 1) / NW;       // lower 16 x 16 tiles of the diagonal tile per wave: 3 (four waves) / 2 (eight)
    constexpr int NE = 512 / NT;                               // 16-byte pieces of a slab per thread
    v4f64 Y[4], dacc[NQ];
    int have = 0;                                              // slabs of L[kk,kk] in ct_T0
    ct_u32x4 pv[4][NE];
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(sbuf), 0, 32768u, 0x00020000);
    if constexpr (TAG) {
        if (tid == 0) ct_task_s[15] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e_ = 0; e_ < NE; ++e_) {
                const int e = tid + NT * e_;
                if ((e >> 3) >= 16 * q) pv[q][e_] = ct_slab_piece<NT>(prs, q, e);
            }
    }
    auto slab = [&](auto s_tag) -> bool {
        constexpr int S = decltype(s_tag)::value;
        if constexpr (TAG) {
            int spins = 0;
            for (;;) {
                bool good = true;
#pragma unroll
                for (int e_ = 0; e_ < NE; ++e_) {
                    const int e = tid + NT * e_;
                    const ct_u32x4 v = pv[S][e_];
                    if ((e >> 3) >= 16 * S && ((v.x == ALABI_CHOL_TAG && v.y == ALABI_CHOL_TAG) || (v.z == ALABI_CHOL_TAG && v.w == ALABI_CHOL_TAG))) good = false;
                }
                if (__all(good)) break;
                if (++spins > spin_limit || ((spins & 63) == 0 && __hip_atomic_load(ct_g32(err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    if (l == 0) { __hip_atomic_store(ct_g32(err), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ct_task_s[4] = ntasks; ct_task_s[15] = 1; }
                    break;
                }
                asm volatile("" ::: "memory");                   // (a fresh load every time round: the builtin is not volatile)
#pragma unroll
                for (int e_ = 0; e_ < NE; ++e_) {
                    const int e = tid + NT * e_;
                    if ((e >> 3) >= 16 * S) pv[S][e_] = ct_slab_piece<NT>(prs, S, e);
                }
            }
#pragma unroll
            for (int e_ = 0; e_ < NE; ++e_) {
                const int e = tid + NT * e_;
                if ((e >> 3) >= 16 * S) ct_slab_piece_put(S, e, pv[S][e_]);
            }
            __syncthreads();                                   // the slab -- and at S = 0 the caller's tiles -- are in LDS
            if (ct_task_s[15] != 0) return false;
            if constexpr (DIAG) CT_LOG(3 + S);
            have = S + 1;
        }
        if (have <= S) {
            if (tid == 0) {
                int v, spins = 0;
                while ((v = __hip_atomic_load(ct_g32(sver), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < S + 1) {
                    if (++spins > spin_limit || ((spins & 63) == 0 && __hip_atomic_load(ct_g32(err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        __hip_atomic_store(ct_g32(err), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ct_task_s[4] = ntasks;
                        v = -1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                ct_task_s[9] = v;
            }
            __syncthreads();
            const int got = ct_task_s[9];
            if (got < 0) return false;
            have = got < 4 ? got : 4;                          // everything that is there, in one round trip
            ct_fetch_slabs<NT>(sbuf, tid, S, have);
            __syncthreads();                                   // the slab(s) -- and at S = 0 the caller's tiles -- are in LDS
            if constexpr (DIAG) { for (int q_ = S; q_ < have; ++q_) CT_LOG(3 + q_); }
        }
        if (w < 4) {
            if (S == 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) Y[t][i] = ct_T1[16 * w + lr][16 * t + 4 * i + lk];
            }
            v4f64 Z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(ct_T0[16 * S + lr][16 * S + 4 * kk + lk], Y[S][kk], Z, 0, 0, 0);
#pragma unroll
            for (int t = S + 1; t < 4; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    Y[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ct_T0[16 * t + lr][16 * S + 4 * kk + lk], Z[kk], Y[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) ct_T1[16 * w + lr][16 * S + 4 * i + lk] = Z[i];
        }
        if constexpr (DIAG) {
            __syncthreads();                                   // slab S of X is in ct_T1 for all 64 rows (and nobody reads ct_T0's slab S any more)
            if (S == 3) tile_store_sc1<NT>(Xdst, ld, ct_T1, tid, false);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int j = w + NW * q;
                if (j < 10) {
                    int rt, ct;
                    ct_diag_tile(j, rt, ct);
                    if (S == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) dacc[q][i] = ct_T2[16 * rt + lk + 4 * i][16 * ct + lr];
                    }
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) {
                        const int ks = 4 * S + kq;
                        dacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ct_T1[16 * rt + lr][4 * ks + lk], ct_T1[16 * ct + lr][4 * ks + lk], dacc[q], 0, 0, 0);
                    }
                    if (S == 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) ct_T0[16 * rt + lk + 4 * i][16 * ct + lr] = dacc[q][i];
                    }
                }
            }
        }
        return true;
    };
    if (!slab(std::integral_constant<int, 0>{})) return false;
    if (!slab(std::integral_constant<int, 1>{})) return false;
    if (!slab(std::integral_constant<int, 2>{})) return false;
    if (!slab(std::integral_constant<int, 3>{})) return false;
    if constexpr (!DIAG) {
        __syncthreads();
        tile_store_sc1<NT>(Xdst, ld, ct_T1, tid, false);
    }
    return true;
}
// The diagonal factorisation of CHAIN(k) (potrf_tile_lds_wg on ct_T0) that hands its result on SLAB BY SLAB: the 16 columns
// of a slab are final for all 64 rows as soon as wave 0 has run the slab's recurrence, and the panel solve of the next chain
// task consumes L[k,k] in exactly that order -- so wave 3, idle while wave 0 runs the next recurrence, writes the slab (and
// its 16 reciprocals) through to memory and, one barrier later when its stores have drained, publishes sver[k] = slab + 1.
// The next CHAIN task then solves slab s while this one factorises slab s + 1 .. 3 instead of starting after the whole tile.
#ifdef ALABI_CHOL_PROF
__device__ long long g_potrf_prof[8];                          // 10-ns ticks inside wave 0's slab recurrences, slabs; [2] start -> barrier A, [3] A -> B, [4] last slab incl. its stores, [5] count
#endif
// The inverses of the slabs' 16 x 16 diagonal blocks, which the matrix-core panel solves multiply by (ct_solve), cost the chain nothing:
//   slabs 1..3: lanes 0..15 of wave 0 -- rows above the slab, idle in its recurrence -- carry the rows of the identity through the SAME
//     recurrence (x L_ss^T = e_i by forward substitution) and come out as the rows of inv(L_ss)^T; the last block, all the next panel solve
//     waits for at the end, goes out at once, the others with their slab;
//   slab 0 (no idle lanes): wave 0 gives up rows 48..63 for the identity, and wave 1 runs the same recurrence beside it for those rows
//     (lanes 16..31; its lanes 0..15 repeat rows 0..15, the source of the multipliers, bit for bit).
// The function must stay within the caller-saved registers and call nothing: with a substitution of ~215 live registers inside it (or a
// call to one) every CHAIN task saved and restored up to 79 registers through scratch memory -- the factorisation took 12-13 us instead of 9.
// (Also measured and not kept: ONE non-inlined copy of the recurrence for every slab, with the slab offset at run time: 1.52 instead of
// 1.26 us per slab.)
// Inside the function the waves do not meet at barriers but follow each other through LDS words (ct_task_s[10..14]; every wait bounded):
//   wave 0          the four slab recurrences; before recurrence s it waits until tile column s carries slab s - 1 ([12])
//   wave 1          slab 0: rows 48..63 behind a copy of rows 0..15 ([13]: 1 = rows read, 2 = rows written back); then owner A
//   owners A, B, C  (waves 1, 2 and 6 -- with four waves 3) the rank-16 updates of the trailing 16 x 16 tiles, every slab of a tile by ONE
//                   wave in order: A (1,1) (2,2), B (2,1) (3,2), C (3,1) (3,3).  When slab s is in LDS ([11] = s + 1) an owner first updates its
//                   tile of column s + 1 -- all the next recurrence needs; counted in [12] -- then its tiles further right, under that recurrence
//   waves 3 (and 7) write slab s -- rows 16 s.., its columns (lower part), the inverse of its diagonal block, its reciprocals -- through to
//                   memory in 16-byte pieces, wait for the stores to drain ([10]: wave 7's half) and publish sver = s + 1 ([14]: inverse
//                   blocks read, before wave 0 reuses the buffer)
// With two barriers per slab instead -- recurrence | stores + every trailing tile | next recurrence -- all eight waves waited 0.6-1.0 us
// per slab for the two storing waves (write-through stores hold the issuing wave), the factorisation took 9.9 us for 4.8 us of recurrences.
// Spinning waves share no SIMD with the recurrences of waves 0 and 1 (waves 4 and 5 sleep at the final barrier).
#define CT_FLAG(i) (*(volatile __attribute__((address_space(3))) int*)&ct_task_s[i])
__device__ inline bool ct_flag_wait(int i, int want, bool sleep) {
    int spins = 0;
    while (CT_FLAG(i) < want) {
        if (++spins > (1 << 22)) return false;                 // (a protocol error, not a slow neighbour: every wave here makes progress on its own)
        if (sleep) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ inline void ct_flag_set(int i, int v, int lane) {      // (LDS operations of one wave are executed in order: the data first)
    asm volatile("" ::: "memory");
    if (lane == 0) CT_FLAG(i) = v;
}
__device__ inline void ct_flag_add(int i, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&ct_task_s[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __attribute__((noinline)) double ct_potrf_publish(int kb, int* info, double* __restrict__ D, int ld, double* __restrict__ dinv,
                                                            int* sver, double* __restrict__ linv, int* err) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool eight = blockDim.x == 512;
    double* const inv_s = ct_pool + 3 * 64 * 66;              // T3: inverses of the slabs' diagonal blocks, [slab & 1][n][k]
    kb = __builtin_amdgcn_readfirstlane(kb);
    if (tid < 5) ct_task_s[10 + tid] = 0;
    __syncthreads();
    bool ok = true;
    // one owner's share of slab S: its tile of column S + 1 first (counted), then its tiles further right
    auto owner_slab = [&](auto s_tag, int which) {
        constexpr int S = decltype(s_tag)::value;
        if (!ct_flag_wait(11, S + 1, true)) { ok = false; return; }
        const int t1r = which + 1, t2r = which == 0 ? 2 : 3, t2c = which == 2 ? 3 : 2;    // (t1r, 1) and (t2r, t2c)
        if (S == 0) {
            tile_update_16<66>(ct_T0, 16 * t1r, 16, ct_T0, 16 * t1r, ct_T0, 16, 0, lane);
            ct_flag_add(12, lane);
            tile_update_16<66>(ct_T0, 16 * t2r, 16 * t2c, ct_T0, 16 * t2r, ct_T0, 16 * t2c, 0, lane);
        } else if (t2c == S + 1) {
            tile_update_16<66>(ct_T0, 16 * t2r, 16 * t2c, ct_T0, 16 * t2r, ct_T0, 16 * t2c, 16 * S, lane);
            ct_flag_add(12, lane);
        } else if (t2c > S + 1) {
            tile_update_16<66>(ct_T0, 16 * t2r, 16 * t2c, ct_T0, 16 * t2r, ct_T0, 16 * t2c, 16 * S, lane);
        }
    };
    // the storing waves' share of slab S
    auto store_slab = [&](auto s_tag) {
        constexpr int S = decltype(s_tag)::value;
        constexpr int c0 = 16 * S;
        if (!ct_flag_wait(11, S + 1, true)) { ok = false; return; }
        if (w == 7) CT_LOGW(16 + 4 * S);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(linv, 0, 32768u, 0x00020000);   // the column's slab buffer
        ct_u32x4 v[8];                                         // every LDS read first (see tile_store_sc1); piece e = lane + 64 p_: row e >> 3 >= c0
        double dl = 1.0;
#pragma unroll
        for (int p_ = 2 * S; p_ < 8; ++p_) {
            const int e = lane + 64 * p_, r = e >> 3, c = 2 * (e & 7);
            if (p_ < 2 * S + 2) { if (w == 3) v[p_] = *reinterpret_cast<const ct_u32x4*>(&inv_s[256 * (S & 1) + (r - c0) * 16 + c]); }   // rows c0 .. c0 + 15: the inverse block (wave 3 alone: [14])
            else v[p_] = *reinterpret_cast<const ct_u32x4*>(&ct_T0[r][c0 + c]);
        }
        if (w == 3) dl = ct_T0[c0 + (lane & 15)][c0 + (lane & 15)];
#pragma unroll
        for (int p_ = 2 * S; p_ < 8; ++p_) asm volatile("" : "+v"(v[p_]));
        if (w == 3) ct_flag_set(14, S + 1, lane);
        if (w == 7) CT_LOGW(17 + 4 * S);
#pragma unroll
        for (int p_ = 2 * S; p_ < 8; ++p_) {
            const int e = lane + 64 * p_;
            if (p_ < 2 * S + 2 ? w == 3 : (!eight || (p_ & 1) == (w >> 2))) __builtin_amdgcn_raw_buffer_store_b128(v[p_], rs, (unsigned)(S * 1024 + 2 * e) * 8u, 0, 16);
        }
        if (w == 3 && lane < 16)
            __hip_atomic_store(ct_g64(dinv + kb * 64 + c0 + lane), (unsigned long long)__double_as_longlong(potrf_dinv(dl)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 7) CT_LOGW(18 + 4 * S);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (w == 7) CT_LOGW(19 + 4 * S);
        if (w == 7) ct_flag_set(10, S + 1, lane);
        else {
            if (eight && !ct_flag_wait(10, S + 1, true)) { ok = false; return; }
            if (lane == 0) __hip_atomic_store(ct_g32(sver), S + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (S == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // in front of the "4" below
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    if (w == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c0 = 16 * s;
            if (s > 0 && ok) ok = ct_flag_wait(12, s == 1 ? 3 : s == 2 ? 5 : 6, false);   // tile column s carries slab s - 1
            double a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = ct_T0[lane][c0 + j];
            {                                                  // rows of the identity: lanes 0..15 (slab 0: lanes 48..63, wave 1 has those rows)
                const int il = s > 0 ? lane : lane - 48;
#pragma unroll
                for (int j = 0; j < 16; ++j) a[j] = (il >= 0 && il < 16) ? (j == il ? 1.0 : 0.0) : a[j];
            }
            potrf_slab(a, c0);
            CT_LOG(10 + s);
            if (s == 0 && ok) ok = ct_flag_wait(13, 1, false);        // wave 1 has read rows 0..15 before they are written back
            if (s >= 2 && ok) ok = ct_flag_wait(14, s - 1, false);    // wave 3 has read block s - 2 out of this half of the buffer
            if (s > 0 ? lane < 16 : lane >= 48) {
#pragma unroll
                for (int j = 0; j < 16; ++j) inv_s[256 * (s & 1) + j * 16 + (lane & 15)] = a[j];   // identity lane i holds row i of inv(L_ss)^T
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) ct_T0[lane][c0 + j] = a[j];
            }
            if (s == 0 && ok) ok = ct_flag_wait(13, 2, false);        // rows 48..63 of slab 0 are in LDS too
            if (s < 3) ct_flag_set(11, s + 1, lane);
            else {                                             // the last inverse block is all the next panel solve waits for: out at once
                const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(linv, 0, 32768u, 0x00020000);
                ct_u32x4 iv[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) iv[q] = *reinterpret_cast<const ct_u32x4*>(&inv_s[256 + 2 * (lane + 64 * q)]);
#pragma unroll
                for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(iv[q]));
#pragma unroll
                for (int q = 0; q < 2; ++q)                    // slab 3 of the slab buffer, rows 48..63
                    __builtin_amdgcn_raw_buffer_store_b128(iv[q], rl, (unsigned)(3 * 1024 + 48 * 16 + 2 * (lane + 64 * q)) * 8u, 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CT_LOG(14);
            }
        }
    } else if (w == 1) {
        {                                                      // rows 48..63 of slab 0 (lanes 16..31) behind a copy of rows 0..15 (lanes 0..15)
            double a[16];
            const int row = (lane & 16) ? 48 + (lane & 15) : (lane & 15);
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = ct_T0[row][j];
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(a[j]));   // (the loads have landed)
            ct_flag_set(13, 1, lane);
            potrf_slab(a, 0);
            if (lane >= 16 && lane < 32) {
#pragma unroll
                for (int j = 0; j < 16; ++j) ct_T0[row][j] = a[j];
            }
            ct_flag_set(13, 2, lane);
        }
        owner_slab(I0{}, 0); owner_slab(I1{}, 0);
    } else if (w == 2) {
        owner_slab(I0{}, 1); owner_slab(I1{}, 1);
    } else if (eight ? w == 6 : w == 3) {
        owner_slab(I0{}, 2); if (!eight) store_slab(I0{});
        owner_slab(I1{}, 2); if (!eight) store_slab(I1{});
        owner_slab(I2{}, 2); if (!eight) store_slab(I2{});
    } else if (w == 3 || w == 7) {
        store_slab(I0{}); store_slab(I1{}); store_slab(I2{});
    }
    if (!ok && lane == 0) __hip_atomic_store(ct_g32(err), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // every slab and every inverse block is out (wave 3 drained slab 2 before it published it, wave 0 block 3 above): the panel solves
    // need nothing else of this tile -- not the last diagonal block, which goes out with the whole tile behind this
    if (tid == 0) __hip_atomic_store(ct_g32(sver), 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w != 0) return 1.0;
    const double lll = ct_T0[lane][lane];
    const int bad = potrf_first_bad(lll);
    if (bad != 0 && lane == 0) atomicCAS(info, 0, kb * 64 + bad);
    return potrf_dinv(lll);
}
// A tile in flight: all 16 loads of a thread are issued before the first one is consumed (several tiles are fetched
// back to back and only then written to LDS: one memory round trip instead of one per tile)
template <int NT> struct TileRegs { unsigned long long v[4096 / NT]; };
template <int NT>
__device__ inline void tile_fetch(TileRegs<NT>& r, const double* __restrict__ src, int ld, int tid) {
#pragma unroll
    for (int e_ = 0; e_ < 4096 / NT; ++e_) {
        const int e = tid + NT * e_;
        r.v[e_] = __hip_atomic_load(ct_g64(src + (size_t)(e >> 6) * ld + (e & 63)), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int NT>
__device__ inline void tile_put(double (*T)[66], const TileRegs<NT>& r, int tid) {
#pragma unroll
    for (int e_ = 0; e_ < 4096 / NT; ++e_) {
        const int e = tid + NT * e_;
        T[e >> 6][e & 63] = __longlong_as_double((long long)r.v[e_]);
    }
}

// The same through 16-byte write-through-coherent (sc1) buffer loads: half the load instructions and twice the bytes per request
// (8-byte sc1 accesses run at 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md) -- the operand stream of the bulk updates.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NT> struct TileRegs16 { u32x4 v[2048 / NT]; };
// PLAIN: ordinary (L2-cached) loads -- valid behind an agent-scope acquire of the hand-off that published the tile (Guideline 16)
template <bool PLAIN, int NT>
__device__ inline void tile_fetch16(TileRegs16<NT>& r, __amdgpu_buffer_rsrc_t rs, unsigned tile_bytes, int ld, int tid) {
#pragma unroll
    for (int e_ = 0; e_ < 2048 / NT; ++e_) {
        const int e = tid + NT * e_;
        r.v[e_] = __builtin_amdgcn_raw_buffer_load_b128(rs, tile_bytes + (unsigned)(((e >> 5) * ld + 2 * (e & 31)) * 8), 0, PLAIN ? 0 : 16);
    }
}
// element e_ of a tile's registers alone: the grouped updates spread the fetch and the LDS write of the next operands over the
// k-steps of the current block column (one of each per k-step pair) instead of issuing them as a burst around the barrier
template <bool PLAIN, int NT>
__device__ inline void tile_fetch16_one(TileRegs16<NT>& r, int e_, __amdgpu_buffer_rsrc_t rs, unsigned tile_bytes, int ld, int tid) {
    const int e = tid + NT * e_;
    r.v[e_] = __builtin_amdgcn_raw_buffer_load_b128(rs, tile_bytes + (unsigned)(((e >> 5) * ld + 2 * (e & 31)) * 8), 0, PLAIN ? 0 : 16);
}
template <int NT>
__device__ inline void tile_put16_one(double (*T)[66], const TileRegs16<NT>& r, int e_, int tid) {
    const int e = tid + NT * e_;
    *reinterpret_cast<u32x4*>(&T[e >> 5][2 * (e & 31)]) = r.v[e_];
}
template <int NT>
__device__ inline void tile_put16(double (*T)[66], const TileRegs16<NT>& r, int tid) {
#pragma unroll
    for (int e_ = 0; e_ < 2048 / NT; ++e_) {
        const int e = tid + NT * e_;
        *reinterpret_cast<u32x4*>(&T[e >> 5][2 * (e & 31)]) = r.v[e_];
    }
}

// NT = 256: four waves, one per SIMD, up to 512 registers per lane (the shape the chain-bound sizes were tuned on).
// NT = 512 (round 3, many block columns): four HELPER waves join for the UPDATE tasks -- two matrix-core waves per SIMD (66-70
// instead of 56-59 TFLOP/s of v_mfma_f64_16x16x4, tools/micro/mfma_f64_rate), wave w owning rows 16 (w & 3).., columns
// 32 (w >> 2).. of the tile -- and for every tile load / store; in the serial parts of CHAIN and TRSM tasks they only keep the
// barriers company, so the chain runs as fast as with four waves (two workgroups of four waves per CU were measured instead:
// the grouped updates gained 27 %, but every recurrence that shared its SIMD with the other workgroup's matrix-core
// instructions took 1.5-1.8x as long and the singles waited five times longer for their inputs; N = 10000 9.86 -> 9.59 ms only).
// BATCH: the queue holds the interleaved task lists of many independent matrices (the hyper-parameter search: candidates x folds,
// gp_utils.py:511-700).  A task names its matrix (`mats`); the queue is cut into `nlists` lists, each with a head counter of its own
// on a 128-byte line of its own (ctl[32 q]; one word saturates at ~88 draws per microsecond, MI355X_MICROARCH.md `dequeue`) and each
// holding whole matrices, so a matrix's tiles stay in one XCD's L2: a workgroup starts on the list of its XCD and moves on to the
// next list when one is exhausted.  Every list is a topological order of its own tasks and a workgroup only waits for tasks in
// front of the one it drew, each of them drawn by a running workgroup: no deadlock, whatever the placement.
template <int NT, bool BATCH>
__device__ __forceinline__ void chol_tasks_body(double* __restrict__ A_, int ld_, int nb_, const CholTask* __restrict__ tasks, int ntasks,
                                                int* __restrict__ ctl, int* __restrict__ info_, double* __restrict__ dinv_, int spin_limit,
                                                const CholMat* __restrict__ mats, const int* __restrict__ list_off, int nlists) {
    double (*T0)[66] = ct_T0; double (*T1)[66] = ct_T1; double (*T2)[66] = ct_T2; double (*T3)[66] = ct_T3;
    int (&task_s)[16] = ct_task_s;
    // per task in a batch, fixed otherwise
    double* A = A_; int ld = ld_, nb = nb_; int* info = info_; double* dinv = dinv_;
    int* head = ctl; int* err = ctl + 1; int* ver = ctl + 2;          // ver[i * nb + j]
    int* sver = ctl + 2 + nb * nb;                                    // sver[k]: slabs of L[k,k] published so far (0..4)
    double* linv = BATCH ? nullptr : reinterpret_cast<double*>(ctl + ((2 + nb * nb + nb + 130 + 1) & ~1));   // the slab buffers, [nb][4][64][16] (ct_fetch_slabs)
    int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
    __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(A, 0, (unsigned)ld * (unsigned)ld * 8u, 0x00020000);
    if constexpr (BATCH) {
        if (tid == 0) {                                               // task_s[6]: current list, [7]: lists found exhausted so far
            int xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            task_s[6] = (xcc & 15) % nlists;
            task_s[7] = 0;
        }
    }
    // (Measured and not kept, round 3: CHAIN(k) applying block column k-2 to its panel tile itself instead of waiting for the
    // one-column UPDATE(k,k-1,k-2) task -- N = 2000 0.55 -> 0.58 ms with four waves, 0.54 -> 0.55 with eight: the period of the
    // chain is set by the 64-pivot factorisation handing its slabs to the next panel solve, not by that task.)
    // (Round 4, again with the matrix-core panel solves, tools/experiments/chol_chain_fused_lookahead.patch: the event log shows the next CHAIN task
    // receiving its tiles only 2.6 us before the previous factorisation ends -- publish, poll, fetch, 64 k-steps, store, publish, poll, fetch
    // behind the previous panel solve -- so the fused task reaches its dependencies 2.8 us earlier, but its own unpipelined update loop takes
    // 4.7 us against the 1 us the tile fetch took: N = 2000 0.518 instead of 0.488 ms.  A pipelined loop would gain ~0.9 us of 13 per step.)
    // (Measured and not kept: a grouped UPDATE drawing the NEXT task under its last block column, to take the queue's atomic round
    // trip off the workgroup's path -- still deadlock-free, and the four-wave kernel gained 3 % at N >= 5000, but the eight-wave
    // kernel lost 1-9 % at every size: a CHAIN task drawn ahead waits for its holder.)
    for (;;) {
        {
            // the per-thread tile offsets of every task type must not be hoisted out of the task loop (the compiler then keeps ~115
            // loop-invariant addresses alive and, with 256 registers per lane, spills them): re-derived per task from an opaque tid
            asm volatile("" : "+v"(tid));
            w = tid >> 6; l = tid & 63; lr = l & 15; lk = l >> 4;
        }
        __syncthreads();                                              // the previous task is done with LDS and task_s
        if (tid == 0) {
            int idx;
            if constexpr (BATCH) {
                int cur = task_s[6], gone = task_s[7];
                for (;;) {
                    if (gone >= nlists) { idx = ntasks; break; }
                    const int beg = list_off[cur], len = list_off[cur + 1] - beg;
                    idx = atomicAdd(ctl + 32 * cur, 1);
                    if (idx < len) { idx += beg; break; }
                    ++gone; cur = cur + 1 < nlists ? cur + 1 : 0;
                }
                task_s[6] = cur; task_s[7] = gone;
            } else {
                idx = atomicAdd(head, 1);
            }
            task_s[4] = idx;
            if (idx < ntasks) {
                const CholTask t = tasks[idx];
                task_s[0] = t.type & 255; task_s[1] = t.i; task_s[2] = t.j; task_s[3] = t.k; task_s[5] = (t.type >> 8) & 255; task_s[8] = t.type >> 16;
            }
        }
        __syncthreads();
        if (task_s[4] >= ntasks) return;
        const int type = task_s[0], ti = task_s[1], tj = task_s[2], tk = task_s[3];
#ifdef ALABI_CHOL_LOG
        const long long log_t0 = __builtin_amdgcn_s_memrealtime();
#endif
        const int tcnt = task_s[5] > 0 ? task_s[5] : 1;                 // UPDATE: block columns tk .. tk + tcnt - 1
        if constexpr (BATCH) {                                        // this task's matrix (wave-uniform: scalar loads)
            const CholMat cm = mats[__builtin_amdgcn_readfirstlane(task_s[8])];
            A = cm.A; ld = cm.ld; nb = cm.nb; ver = cm.ver; sver = cm.sver; info = cm.info; dinv = cm.dinv; linv = cm.linv;
            arsrc = __builtin_amdgcn_make_buffer_rsrc(A, 0, (unsigned)ld * (unsigned)ld * 8u, 0x00020000);
        }
#ifdef ALABI_CHOL_PROF
        const long long pw0 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- dependencies: up to eight (tile, version) pairs, polled by lanes 0..7 of wave 0
        if (w == 0) {
            int di_ = 0, dj_ = 0, need = 0;                           // lane 0 / 1 / 2
            if (type == 0) {                                          // CHAIN(k): tile (k,k-1) and (k,k) at k-1
                // (L[k-1,k-1] is NOT waited for here: its slabs are taken one by one below)
                if (l == 1) { di_ = tk; dj_ = tk - 1; need = tk - 1; }
                if (l == 2) { di_ = tk; dj_ = tk; need = tk - 1; }
                if (tk == 0) need = 0;
                if (tk == 0) { di_ = 0; dj_ = 0; }
            } else if (type == 1) {                                   // TRSM(i,k): tile (i,k) at k (L[k,k] is taken slab by slab)
                if (l == 1) { di_ = ti; dj_ = tk; need = tk; }
            } else {                                                  // UPDATE(i,j,k..kl): (i,kl), (j,kl) final (then so are the
                const int kl = tk + tcnt - 1;                         // panels before them), tile (i,j) at version k
                if (l == 0) { di_ = ti; dj_ = kl; need = kl + 1; }
                if (l == 1) { di_ = tj; dj_ = kl; need = kl + 1; }
                if (l == 2) { di_ = ti; dj_ = tj; need = tk; }
                if (type >= 4) {                                      // UPDATE2 / UPDATE4: the same for tile row i + 1
                    if (l == 3) { di_ = ti + 1; dj_ = kl; need = kl + 1; }
                    if (l == 4) { di_ = ti + 1; dj_ = tj; need = tk; }
                }
                if (type == 5) {                                      // UPDATE4: and for tile column j + 1
                    if (l == 5) { di_ = tj + 1; dj_ = kl; need = kl + 1; }
                    if (l == 6) { di_ = ti; dj_ = tj + 1; need = tk; }
                    if (l == 7) { di_ = ti + 1; dj_ = tj + 1; need = tk; }
                }
            }
            const bool active = l < 8 && need > 0;
            int spins = 0, ok = 1;
            while (true) {
                int have = need;
                if (active) have = __hip_atomic_load(ct_g32(ver + di_ * nb + dj_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(have >= need)) break;
                if (++spins > spin_limit || ((spins & 63) == 0 && __hip_atomic_load(ct_g32(err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok && l == 0) { __hip_atomic_store(ct_g32(err), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); task_s[4] = ntasks; }
            // an UPDATE over a whole group of block columns streams its operand tiles with ordinary loads (they can hit in the XCD's
            // L2, where the neighbouring tasks of the same tile column have just put them; write-through-coherent loads always go
            // out to the fabric, and the bulk updates are bound by exactly that traffic): one acquire per task makes that valid
            if ((type == 2 && tcnt >= ALABI_CHOL_PLAIN_MIN) || type >= 4) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
#ifdef ALABI_CHOL_PROF
        if (tid == 0 && type == 0 && tk > 0) reinterpret_cast<long long*>(ctl + ((2 + nb * nb + nb + 1) & ~1))[7] += __builtin_amdgcn_s_memrealtime() - pw0;
#endif
        if (task_s[4] >= ntasks) return;                              // a wait ran out: every workgroup leaves at its next check
        if (type == 2) {
            // ---------------- UPDATE(i, j, k .. k + tcnt - 1): C(i,j) -= sum_k' A(i,k') A(j,k')^T, accumulated in registers over
            // the whole range (C is read and written ONCE per task); the operand tiles of column k' + 1 are in flight while the
            // matrix cores work on column k' (two LDS operand pairs).  Measured and not kept: operands two columns ahead (a
            // second register set; as part of this kernel it spills, as a function of its own the call costs every task more than
            // the deeper prefetch gains -- the grouped tasks were no faster, so the fetch latency is not what bounds them).
#ifdef ALABI_CHOL_PROF
            const long long u0 = __builtin_amdgcn_s_memrealtime();
            long long u1 = u0, u2 = u0;
#endif
            auto update_range = [&](auto plain_tag) {
                constexpr bool PL = decltype(plain_tag)::value;
                constexpr int NN = NT == 512 ? 2 : 4;                  // 16 x 16 tiles per wave: 16 rows x (64 or 32) columns
                const int wr = w & 3, c0w = NT == 512 ? 32 * (w >> 2) : 0;
                TileRegs16<NT> ra, rb;
                const unsigned row_i = (unsigned)(ti * 64) * (unsigned)ld * 8u, row_j = (unsigned)(tj * 64) * (unsigned)ld * 8u;
                tile_fetch16<PL, NT>(ra, arsrc, row_i + (unsigned)tk * 512u, ld, tid);
                tile_fetch16<PL, NT>(rb, arsrc, row_j + (unsigned)tk * 512u, ld, tid);
                double* C = A + (size_t)(ti * 64 + 16 * wr) * ld + tj * 64 + c0w;
                v4f64 acc[NN];
    #pragma unroll
                for (int n = 0; n < NN; ++n)
    #pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[n][i] = __longlong_as_double((long long)__hip_atomic_load(
                            ct_g64(C + (size_t)(lk + 4 * i) * ld + 16 * n + lr), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT));
                tile_put16<NT>(T0, ra, tid); tile_put16<NT>(T1, rb, tid);
                // Column k' + 1 waits in LDS and column k' + 2 is in flight while the matrix cores work on column k': the registers of
                // a fetch are written to the other operand pair at the START of the next iteration (its last readers passed the
                // barrier before) and refilled at once, so no wave waits for memory or for the LDS writes between two columns.
                // The C tile has landed before the loop starts: otherwise the compiler's wait for it sits INSIDE the loop (vmcnt is one
                // in-order counter) and drains the operand prefetch of every iteration.
                if (tcnt > 1) {
                    tile_fetch16<PL, NT>(ra, arsrc, row_i + (unsigned)(tk + 1) * 512u, ld, tid);
                    tile_fetch16<PL, NT>(rb, arsrc, row_j + (unsigned)(tk + 1) * 512u, ld, tid);
                }
    #pragma unroll
                for (int n = 0; n < NN; ++n) asm volatile("" : "+v"(acc[n]));   // C is waited for HERE (the fetch above stays in flight)
                __syncthreads();
#ifdef ALABI_CHOL_PROF
                u1 = __builtin_amdgcn_s_memrealtime();
#endif
                // One block column: MORE = column c + 1 exists (its pieces go registers -> the other LDS pair), MORE2 = column c + 2 exists
                // (memory -> the same registers).  Compile-time flags, so that a column is straight-line code and the compiler can count
                // vmcnt exactly: behind a branch it waits for vmcnt(0) in front of every piece, i.e. for the load issued one k-step pair ago.
                auto column = [&](auto more_tag, auto more2_tag, int c) {
                    constexpr bool MORE = decltype(more_tag)::value, MORE2 = decltype(more2_tag)::value;
                    const unsigned col2 = (unsigned)(tk + c + 2) * 512u;
                    double (*Pa)[66] = (c & 1) ? T0 : T2;                 // the other pair: block column c + 1 goes there
                    double (*Pb)[66] = (c & 1) ? T1 : T3;
                    double (*Ta)[66] = (c & 1) ? T2 : T0;
                    double (*Tb)[66] = (c & 1) ? T3 : T1;
                    // software pipeline over pairs of k-steps: the LDS reads of pair kp + 1 are issued before the matrix-core instructions
                    // of pair kp (two register sets; sched_barrier keeps the compiler from sinking the reads next to their uses --
                    // it otherwise reads, waits, multiplies, and every pair of k-steps exposes one LDS round trip)
                    double pa[2][2], pb[2][2][NN];
                    auto lds_pair = [&](int set, int kp) {
    #pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            // volatile: ONE ds_read_b64 per operand (conflict-free with the row stride of 66: 2 LDS cycles).  Left to
                            // itself the compiler pairs them into ds_read2_b64, which is banked modulo 32 and serviced in groups of 16
                            // lanes: rows r and r + 8 collide, 16 LDS cycles per instruction -- the LDS then co-limits the loop
                            pa[set][h] = lds_read_b64(&Ta[16 * wr + lr][4 * (2 * kp + h) + lk]);
    #pragma unroll
                            for (int n = 0; n < NN; ++n)
                                pb[set][h][n] = lds_read_b64(&Tb[c0w + 16 * n + lr][4 * (2 * kp + h) + lk]);
                        }
                    };
                    lds_pair(0, 0);
    #pragma unroll
                    for (int kp = 0; kp < 8; ++kp) {
                        if (kp < 7) lds_pair((kp + 1) & 1, kp + 1);
                        // one piece of block column c + 1 and of column c + 2 per k-step pair (NE pieces per operand tile, 2 NE / 8 per
                        // pair) instead of a burst of LDS writes and loads around the barrier, when no wave has matrix-core work
                        {
                            constexpr int NE = 2048 / NT, PER = 2 * NE / 8;
    #pragma unroll
                            for (int q = PER * kp; q < PER * (kp + 1); ++q) {
                                if (q < NE) {
                                    if (MORE) tile_put16_one<NT>(Pa, ra, q, tid);
                                    if (MORE2) tile_fetch16_one<PL, NT>(ra, q, arsrc, row_i + col2, ld, tid);
                                } else {
                                    if (MORE) tile_put16_one<NT>(Pb, rb, q - NE, tid);
                                    if (MORE2) tile_fetch16_one<PL, NT>(rb, q - NE, arsrc, row_j + col2, ld, tid);
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                        for (int h = 0; h < 2; ++h)
    #pragma unroll
                            for (int n = 0; n < NN; ++n)
                                acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[kp & 1][h], pb[kp & 1][h][n], acc[n], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                for (int c = 0; c < tcnt; ++c) {
                    const bool more = c + 1 < tcnt;
                    if (c + 2 < tcnt) column(std::true_type{}, std::true_type{}, c);
                    else if (more) column(std::true_type{}, std::false_type{}, c);
                    else column(std::false_type{}, std::false_type{}, c);
                    if (more) __syncthreads();                            // pair (c + 1) is complete, pair c may be overwritten
                }
    #pragma unroll
                for (int n = 0; n < NN; ++n)
    #pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __hip_atomic_store(ct_g64(C + (size_t)(lk + 4 * i) * ld + 16 * n + lr),
                                           (unsigned long long)__double_as_longlong(acc[n][i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            if (tcnt >= ALABI_CHOL_PLAIN_MIN) update_range(std::true_type{}); else update_range(std::false_type{});
#ifdef ALABI_CHOL_PROF
            u2 = __builtin_amdgcn_s_memrealtime();
#endif
            publish_version(ver + ti * nb + tj, tk + tcnt, tid);
#ifdef ALABI_CHOL_PROF
            if (tid == 0) {   // 10-ns units, grouped updates [8..13], single-column updates [16..21]: wait for deps, first fetch + C, loop, store + publish, count, columns
                unsigned long long* up = reinterpret_cast<unsigned long long*>(ctl + ((2 + nb * nb + nb + 1) & ~1)) + (tcnt >= ALABI_CHOL_PLAIN_MIN ? 8 : 16);
                const long long u3 = __builtin_amdgcn_s_memrealtime();
                atomicAdd(up + 0, (unsigned long long)(u0 - pw0)); atomicAdd(up + 1, (unsigned long long)(u1 - u0));
                atomicAdd(up + 2, (unsigned long long)(u2 - u1)); atomicAdd(up + 3, (unsigned long long)(u3 - u2));
                atomicAdd(up + 4, 1ull); atomicAdd(up + 5, (unsigned long long)tcnt);
            }
#endif
        } else if (type == 5) {
            // ---------------- UPDATE4(i, j, k .. k + tcnt - 1): tiles (i, j), (i + 1, j), (i, j + 1), (i + 1, j + 1) in one task -- a 128 x 128
            // output, wave w owning rows 32 (w & 3) .., columns 64 (w >> 2) .. (2 x 4 accumulator tiles: six operand reads feed eight
            // matrix-core instructions), four operand tiles per block column for four output tiles, the fixed cost of a task once per
            // four tiles.  Eight 64 x 34 half tiles (two buffers of four) fill the pool; otherwise as UPDATE2.  i >= j + 1, so that
            // tile (i, j + 1) is in the lower triangle ((j + 1, j + 1) is a diagonal tile: its update is the whole symmetric tile).
            if constexpr (NT == 512) {
                double (*H)[34] = reinterpret_cast<double (*)[34]>(ct_pool);
                const int wr2 = w & 3, wc = w >> 2;
                const int prow = tid >> 3, pcol = 2 * (tid & 7);
                const unsigned rowb[4] = {(unsigned)(ti * 64 + prow) * (unsigned)ld * 8u, (unsigned)((ti + 1) * 64 + prow) * (unsigned)ld * 8u,
                                          (unsigned)(tj * 64 + prow) * (unsigned)ld * 8u, (unsigned)((tj + 1) * 64 + prow) * (unsigned)ld * 8u};
                u32x4 pc[8];
                auto request = [&](int p, int hs) {
                    pc[p] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, rowb[p >> 1] + (unsigned)(((tk + (hs >> 1)) * 64 + 32 * (hs & 1) + 16 * (p & 1) + pcol) * 8), 0, 0);
                };
                auto to_lds = [&](int p, int buf) {
                    *reinterpret_cast<u32x4*>(&H[(buf * 4 + (p >> 1)) * 64 + prow][16 * (p & 1) + pcol]) = pc[p];
                };
                const int nhs = 2 * tcnt;
#pragma unroll
                for (int p = 0; p < 8; ++p) request(p, 0);
                // C: rows 32 wr2 .. of the 128-row pair (tile i or i + 1), columns of tile j + wc
                double* C = A + (size_t)(ti * 64 + 32 * wr2) * ld + (tj + wc) * 64;
                v4f64 acc[2][4];
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[ri][n][i] = __longlong_as_double((long long)__hip_atomic_load(
                                ct_g64(C + (size_t)(16 * ri + lk + 4 * i) * ld + 16 * n + lr), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
                for (int p = 0; p < 8; ++p) to_lds(p, 0);
#pragma unroll
                for (int p = 0; p < 8; ++p) request(p, 1);
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(acc[ri][n]));        // C is waited for HERE
                __syncthreads();
                auto half_stage = [&](auto more_tag, auto more2_tag, int hs) {
                    constexpr bool MORE = decltype(more_tag)::value, MORE2 = decltype(more2_tag)::value;
                    const int buf = hs & 1;
                    double (*Ha)[34] = H + (buf * 4 + (wr2 >> 1)) * 64 + 32 * (wr2 & 1);   // this wave's 32 rows of A(i) or A(i+1)
                    double (*Hb)[34] = H + (buf * 4 + 2 + wc) * 64;                         // its 64 columns = the rows of A(j) or A(j+1)
                    double pa[2][2][2], pb[2][2][4];
                    auto lds_pair = [&](int set, int kp) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int kk = 4 * (2 * kp + h) + lk;
                            pa[set][h][0] = lds_read_b64(&Ha[lr][kk]); pa[set][h][1] = lds_read_b64(&Ha[16 + lr][kk]);
#pragma unroll
                            for (int n = 0; n < 4; ++n) pb[set][h][n] = lds_read_b64(&Hb[16 * n + lr][kk]);
                        }
                    };
                    lds_pair(0, 0);
#pragma unroll
                    for (int kp = 0; kp < 4; ++kp) {
                        if (kp < 3) lds_pair((kp + 1) & 1, kp + 1);
#pragma unroll
                        for (int p = 2 * kp; p < 2 * kp + 2; ++p) {
                            if (MORE) to_lds(p, buf ^ 1);
                            if (MORE2) request(p, hs + 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                                for (int n = 0; n < 4; ++n)
                                    acc[ri][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[kp & 1][h][ri], pb[kp & 1][h][n], acc[ri][n], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                for (int hs = 0; hs < nhs; ++hs) {
                    const bool more = hs + 1 < nhs;
                    if (hs + 2 < nhs) half_stage(std::true_type{}, std::true_type{}, hs);
                    else if (more) half_stage(std::true_type{}, std::false_type{}, hs);
                    else half_stage(std::false_type{}, std::false_type{}, hs);
                    if (more) __syncthreads();
                }
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            __hip_atomic_store(ct_g64(C + (size_t)(16 * ri + lk + 4 * i) * ld + 16 * n + lr),
                                               (unsigned long long)__double_as_longlong(acc[ri][n][i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < 4)
                __hip_atomic_store(ct_g32(ver + (ti + (tid & 1)) * nb + tj + (tid >> 1)), tk + tcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (type == 4) {
            // ---------------- UPDATE2(i, j, k .. k + tcnt - 1): the grouped update of tiles (i, j) AND (i + 1, j) in one task (eight-wave
            // kernel only).  A 128 x 64 output: wave w owns rows 32 (w & 3) .., columns 32 (w >> 2) .. (2 x 2 accumulator tiles: two A
            // and two B operand reads feed four matrix-core instructions, 1.0 LDS read per instruction instead of 1.5), the operand
            // tile A(j, k') is fetched once for both rows, and the fixed cost of a task (queue draw, dependency poll, first fetch, C round
            // trip, publish: 4.3 us) is paid once per two tiles.  Three operand tiles per block column do not fit twice beside each other
            // in 135 KB, so a stage is HALF a block column (32 k-values): six 64 x 34 half tiles = two buffers in the pool; while the
            // k-steps of half-stage s run, half-stage s + 1 goes from registers to the other buffer and s + 2 from memory into the same
            // registers, piece by piece (six 16-byte pieces per thread: vmcnt(5) in front of each).  Every output element receives its
            // k-steps in the same order as in UPDATE: the same bits.
            if constexpr (NT == 512) {
                double (*H)[34] = reinterpret_cast<double (*)[34]>(ct_pool);                  // half tile q: rows 64 q .. 64 q + 63
                const int wr2 = w & 3, wc = w >> 2;
                const int prow = tid >> 3, pcol = 2 * (tid & 7);                             // this thread's piece of a half tile: 16 bytes
                const unsigned rowb[3] = {(unsigned)(ti * 64 + prow) * (unsigned)ld * 8u, (unsigned)((ti + 1) * 64 + prow) * (unsigned)ld * 8u,
                                          (unsigned)(tj * 64 + prow) * (unsigned)ld * 8u};
                u32x4 pc[6];
                auto request = [&](int p, int hs) {                                          // half-stage hs = 2 (block column) + half
                    pc[p] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, rowb[p >> 1] + (unsigned)(((tk + (hs >> 1)) * 64 + 32 * (hs & 1) + 16 * (p & 1) + pcol) * 8), 0, 0);
                };
                auto to_lds = [&](int p, int buf) {
                    *reinterpret_cast<u32x4*>(&H[(buf * 3 + (p >> 1)) * 64 + prow][16 * (p & 1) + pcol]) = pc[p];
                };
                const int nhs = 2 * tcnt;
#pragma unroll
                for (int p = 0; p < 6; ++p) request(p, 0);
                double* C = A + (size_t)(ti * 64 + 32 * wr2) * ld + tj * 64 + 32 * wc;      // rows 32 wr2 .. of the 128-row pair
                v4f64 acc[2][2];
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[ri][n][i] = __longlong_as_double((long long)__hip_atomic_load(
                                ct_g64(C + (size_t)(16 * ri + lk + 4 * i) * ld + 16 * n + lr), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
                for (int p = 0; p < 6; ++p) to_lds(p, 0);
#pragma unroll
                for (int p = 0; p < 6; ++p) request(p, 1);                                    // (nhs >= 2 always)
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(acc[ri][n]));        // C is waited for HERE
                __syncthreads();
                auto half_stage = [&](auto more_tag, auto more2_tag, int hs) {
                    constexpr bool MORE = decltype(more_tag)::value, MORE2 = decltype(more2_tag)::value;
                    const int buf = hs & 1;
                    double (*Ha)[34] = H + (buf * 3 + (wr2 >> 1)) * 64 + 32 * (wr2 & 1);   // this wave's 32 rows of A(i) or A(i+1)
                    double (*Hb)[34] = H + (buf * 3 + 2) * 64 + 32 * wc;                    // its 32 columns = rows of A(j)
                    double pa[2][2][2], pb[2][2][2];
                    auto lds_pair = [&](int set, int kp) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int kk = 4 * (2 * kp + h) + lk;
                            pa[set][h][0] = lds_read_b64(&Ha[lr][kk]); pa[set][h][1] = lds_read_b64(&Ha[16 + lr][kk]);
                            pb[set][h][0] = lds_read_b64(&Hb[lr][kk]); pb[set][h][1] = lds_read_b64(&Hb[16 + lr][kk]);
                        }
                    };
                    lds_pair(0, 0);
#pragma unroll
                    for (int kp = 0; kp < 4; ++kp) {
                        if (kp < 3) lds_pair((kp + 1) & 1, kp + 1);
#pragma unroll
                        for (int p = 2 * kp; p < 2 * kp + 2 && p < 6; ++p) {
                            if (MORE) to_lds(p, buf ^ 1);
                            if (MORE2) request(p, hs + 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                                for (int n = 0; n < 2; ++n)
                                    acc[ri][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[kp & 1][h][ri], pb[kp & 1][h][n], acc[ri][n], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                for (int hs = 0; hs < nhs; ++hs) {
                    const bool more = hs + 1 < nhs;
                    if (hs + 2 < nhs) half_stage(std::true_type{}, std::true_type{}, hs);
                    else if (more) half_stage(std::true_type{}, std::false_type{}, hs);
                    else half_stage(std::false_type{}, std::false_type{}, hs);
                    if (more) __syncthreads();
                }
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            __hip_atomic_store(ct_g64(C + (size_t)(16 * ri + lk + 4 * i) * ld + 16 * n + lr),
                                               (unsigned long long)__double_as_longlong(acc[ri][n][i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(ct_g32(ver + ti * nb + tj), tk + tcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ct_g32(ver + (ti + 1) * nb + tj), tk + tcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (type == 1) {
            // ---------------- TRSM(i, k)
            {   // (16-byte coherent loads: half the instructions of the 8-byte form, and the faster rate per byte)
                TileRegs16<NT> rb;
                tile_fetch16<false, NT>(rb, arsrc, (unsigned)(ti * 64) * (unsigned)ld * 8u + (unsigned)tk * 512u, ld, tid);
                tile_put16<NT>(T1, rb, tid);
            }
            if (!ct_solve<NT, false, !BATCH>(ld, linv + (size_t)tk * 4096, sver + tk, err, spin_limit, ntasks,
                                     A + (size_t)(ti * 64) * ld + tk * 64)) return;
            publish_version(ver + ti * nb + tk, tk + 1, tid);
        } else {
            // ---------------- CHAIN(k)
            double* D = A + (size_t)(tk * 64) * ld + tk * 64;
#ifdef ALABI_CHOL_LOG
            if (tid == 0) { ct_log_kb = tk; g_chain_log[tk][0] = log_t0; g_chain_log[tk][1] = __builtin_amdgcn_s_memrealtime(); }
            __syncthreads();
#endif
#ifdef ALABI_CHOL_PROF
            long long* prof = reinterpret_cast<long long*>(ctl + ((2 + nb * nb + nb + 1) & ~1));
            const long long p0 = __builtin_amdgcn_s_memrealtime();
            long long p1 = p0, p2 = p0, p3 = p0, p4 = p0;
#endif
            if (tk > 0) {
                {
                    TileRegs16<NT> rb, rc;
                    tile_fetch16<false, NT>(rb, arsrc, (unsigned)(tk * 64) * (unsigned)ld * 8u + (unsigned)(tk - 1) * 512u, ld, tid);
                    tile_fetch16<false, NT>(rc, arsrc, (unsigned)(tk * 64) * (unsigned)ld * 8u + (unsigned)tk * 512u, ld, tid);
                    tile_put16<NT>(T1, rb, tid); tile_put16<NT>(T2, rc, tid);
                }
#ifdef ALABI_CHOL_PROF
                p1 = __builtin_amdgcn_s_memrealtime();
#endif
                CT_LOG(2);
                // the panel solve on the matrix cores, slab by slab as CHAIN(k-1) publishes the slabs of L[k-1,k-1] and the inverses of their
                // diagonal blocks (bounded wait each); tile (k,k) -= X X^T follows it one slab behind and ends up in T0 (ct_solve)
                if (!ct_solve<NT, true, !BATCH>(ld, linv + (size_t)(tk - 1) * 4096, sver + tk - 1, err,
                                        spin_limit, ntasks, A + (size_t)(tk * 64) * ld + (tk - 1) * 64)) return;
#ifdef ALABI_CHOL_PROF
                p2 = __builtin_amdgcn_s_memrealtime();
#endif
                CT_LOG(7);
                publish_version(ver + tk * nb + (tk - 1), tk, tid);      // the solved panel tile is final: updates of column k can start
                CT_LOG(8);
#ifdef ALABI_CHOL_PROF
                p3 = __builtin_amdgcn_s_memrealtime();
#endif
            } else {
                tile_load_sc1<NT>(T0, D, ld, tid);
            }
            __syncthreads();
#ifdef ALABI_CHOL_PROF
            p4 = __builtin_amdgcn_s_memrealtime();
#endif
            CT_LOG(9);
            const double rinv = ct_potrf_publish(tk, info, D, ld, dinv, sver + tk, linv + (size_t)tk * 4096, err);
#ifdef ALABI_CHOL_PROF
            const long long p5 = __builtin_amdgcn_s_memrealtime();
#endif
            if (w == 0) __hip_atomic_store(ct_g64(dinv + tk * 64 + l),
                                           (unsigned long long)__double_as_longlong(rinv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tile_store_sc1<NT>(D, ld, T0, tid, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(ct_g32(sver + tk), 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ct_g32(ver + tk * nb + tk), tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            CT_LOG(15);
#ifdef ALABI_CHOL_PROF
            if (tid == 0 && tk > 0) {   // 10-ns units: [0] loads [1] trsm [2] store+publish panel [3] mfma+park [4] potrf [5] store+publish diag [6] count [7] wait for deps
                const long long p6 = __builtin_amdgcn_s_memrealtime();
                prof[0] += p1 - p0; prof[1] += p2 - p1; prof[2] += p3 - p2; prof[3] += p4 - p3; prof[4] += p5 - p4; prof[5] += p6 - p5; prof[6] += 1;
            }
#endif
        }
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
chol_tasks_kernel(double* __restrict__ A, int ld, int nb, const CholTask* __restrict__ tasks, int ntasks, int* __restrict__ ctl,
                  int* __restrict__ info, double* __restrict__ dinv, int spin_limit) {
    chol_tasks_body<256, false>(A, ld, nb, tasks, ntasks, ctl, info, dinv, spin_limit, nullptr, nullptr, 1);
}
__global__ void __launch_bounds__(512)
chol_tasks8_kernel(double* __restrict__ A, int ld, int nb, const CholTask* __restrict__ tasks, int ntasks, int* __restrict__ ctl,
                   int* __restrict__ info, double* __restrict__ dinv, int spin_limit) {
    chol_tasks_body<512, false>(A, ld, nb, tasks, ntasks, ctl, info, dinv, spin_limit, nullptr, nullptr, 1);
}
// ctl: [32 q] head of list q (q < nlists <= 8), [1] time-out flag
__global__ void __launch_bounds__(512)
chol_tasks8_batch_kernel(const CholMat* __restrict__ mats, const CholTask* __restrict__ tasks, int ntasks, const int* __restrict__ list_off,
                         int nlists, int* __restrict__ ctl, int spin_limit) {
    chol_tasks_body<512, true>(nullptr, 64, 1, tasks, ntasks, ctl, nullptr, nullptr, spin_limit, mats, list_off, nlists);
}

static int tiles_in_cols(int ntr, int tc0, int tc1) {
    int n = 0;
    for (int tj = tc0; tj < tc1; ++tj) n += ntr - tj;
    return n;
}

// Task list of the queue kernel for nb block columns: a static topological order, drawn from one counter.
//   step k (block column k is final once its tasks are done):
//     CHAIN(k+1)                      solve tile (k+1,k), update and factorise tile (k+1,k+1), all in one workgroup
//     TRSM(i,k), i >= k+2             the rest of panel k
//     UPDATE(i,j,k) with ONE block column for the tile columns j = k+1 (inputs of CHAIN(k+2) and of panel k+1) .. k+near
//     UPDATE(i,j,[far(j), k]) for tile column j = k+1+near: everything it has not received yet, in one task
//     UPDATE(i,j,[k+1-gk, k]) for the tile columns beyond, whenever a group of gk block columns is complete
//   far(j) = gk * floor((j - near) / gk) (0 below).  A tile far from the chain takes the block columns in groups of gk -- C is
//   read and written once per group instead of once per block column, and a task carries gk x 64 matrix-core instructions per wave
//   against its fixed cost (queue draw, dependency poll, first fetch, C round trip: 3.5 us against 2.7 us per block column) --,
//   catches up in one task when the chain is near + 1 columns away, and from then on takes every block column as soon as it is
//   final, so that nothing the chain needs waits for a group to fill.
// Every task depends only on tasks before it in the list (tests/test_abi.py replays the order on the host).
// (Round 3, measured and not kept: a batch of grouped updates dealt over the following gk steps -- a share per step, in front of or
// behind its one-column updates, each column's share flushed before anything else touches the column -- so that the one-column
// tasks find their panel tiles solved instead of waiting 4.4 us each: N = 3072 0.84 -> 0.81 ms, but 8192 5.0 -> 5.55 and 10000
// 8.1 -> 8.6-8.7 either way; a batch in one piece keeps the operand tiles of a tile column in the XCDs' L2s while they are used.)
static void chol_build_tasks(int nb, int gk, int near, bool two, std::vector<CholTask>& t) {
    // 2 x 2 tiles per grouped update from ALABI_CHOL_UPDATE4_MIN_NB block columns on (measured: N = 3072 0.87 -> 0.94 ms, 5000 1.82 -> 1.88,
    // 8192 5.09 -> 4.97, 10000 8.36 -> 8.13, 16000 28.9 -> 27.5: the big tasks pay when the trailing matrix is wide)
    bool four = two && nb >= ALABI_CHOL_UPDATE4_MIN_NB;
    if (const char* e = getenv("ALABI_CHOL_UPDATE4")) four = two && e[0] == '1';
    auto far = [&](int j) { return (j - near) < 0 ? 0 : (j - near) / gk * gk; };
    t.clear();
    t.push_back({0, 0, 0, 0});
    for (int k = 0; k + 1 < nb; ++k) {
        t.push_back({0, k + 1, k + 1, k + 1});
        for (int i = k + 2; i < nb; ++i) t.push_back({1, i, k, k});
        for (int i = k + 2; i < nb; ++i) t.push_back({2 | (1 << 8), i, k + 1, k});
        for (int j = k + 2; j < nb && j <= k + near; ++j)
            for (int i = j; i < nb; ++i) t.push_back({2 | (1 << 8), i, j, k});
        const int jc = k + 1 + near;                          // catches up: block columns [far(jc), k]
        if (jc < nb && far(jc) <= k)                          // (one tile per task: as UPDATE2 pairs these cost 6 % at N = 10000 -- the chain is near)
            for (int i = jc; i < nb; ++i) t.push_back({2 | ((k + 1 - far(jc)) << 8), i, jc, far(jc)});
        if ((k + 1) % gk == 0)
            for (int j = k + 1 + near; j < nb; ++j) {
                if (far(j) < k + 1) continue;                 // (j = k+1+near has far(j) = k+1 here: its catch-up task above is empty)
                const int k0 = k + 1 - gk;
                if (four && j + 1 < nb) {
                    // tile columns j and j + 1 together: the diagonal tile (j, j) alone, then 2 x 2 blocks of tiles from row j + 1 on
                    t.push_back({2 | (gk << 8), j, j, k0});
                    int i = j + 1;
                    for (; i + 1 < nb; i += 2) t.push_back({5 | (gk << 8), i, j, k0});
                    if (i < nb) { t.push_back({2 | (gk << 8), i, j, k0}); t.push_back({2 | (gk << 8), i, j + 1, k0}); }
                    ++j;
                    continue;
                }
                for (int i = j; i < nb; ++i) {
                    if (two && i + 1 < nb) { t.push_back({4 | (gk << 8), i, j, k0}); ++i; }            // tiles (i, j) and (i + 1, j)
                    else t.push_back({2 | (gk << 8), i, j, k0});
                }
            }
    }
}

// The list for a matrix that shares the queue with many others (chol_batch_build).  There the chip is kept busy by the OTHER
// matrices, so nothing has to be fed to a matrix's own chain early and no tile takes a block column on its own: every tile receives
// full groups of gk block columns while the chain is far, and ONE catch-up task brings it up to date at the last moment --
//   off-diagonal (i, j): at step j - 1 (then TRSM(i, j) / CHAIN(j) can solve it), diagonal (j, j): at step j - 2 (CHAIN(j) applies
//   column j - 1 itself); the catch-up covers [gk floor(c / gk), c] for catch-up step c, the groups before it are complete.
// With gk >= nb this is the left-looking factorisation: every tile is read and written once.  Against the list above (near = 4):
// 276 single-column tasks fewer per matrix of 25 block columns, each of which paid a task's fixed cost (queue draw, dependency
// poll, first fetch, C round trip) for 64 matrix-core instructions per wave.  Every tile still receives its block columns in
// ascending order inside register accumulators: the same bits.
static void chol_build_tasks_batch(int nb, int gk, bool two, bool four, std::vector<CholTask>& t) {
    t.clear();
    t.push_back({0, 0, 0, 0});
    auto grouped = [&](int j, int i0, int cnt, int k0) {                 // tiles (i, j), i = i0 .. nb - 1, block columns k0 .. k0 + cnt - 1
        for (int i = i0; i < nb; ++i) {
            if (two && cnt >= 2 && i + 1 < nb) { t.push_back({4 | (cnt << 8), i, j, k0}); ++i; }
            else t.push_back({2 | (cnt << 8), i, j, k0});
        }
    };
    for (int k = 0; k + 1 < nb; ++k) {
        t.push_back({0, k + 1, k + 1, k + 1});
        for (int i = k + 2; i < nb; ++i) t.push_back({1, i, k, k});
        const int c0 = k / gk * gk, cc = k + 1 - c0;                      // catch-ups of this step: block columns [c0, k]
        grouped(k + 1, k + 2, cc, c0);                                    // column k + 1 below its diagonal tile
        if (k + 2 < nb) t.push_back({2 | (cc << 8), k + 2, k + 2, c0});   // diagonal tile (k + 2, k + 2)
        if ((k + 1) % gk == 0) {                                          // a group is complete: everything whose catch-up is still ahead
            const int k0 = k + 1 - gk;
            grouped(k + 2, k + 3, gk, k0);                                // column k + 2 without its diagonal tile (caught up above)
            for (int j = k + 3; j < nb; ++j) {
                if (four && gk >= 2 && j + 1 < nb) {                      // tile columns j and j + 1: (j, j) alone, then 2 x 2 blocks
                    t.push_back({2 | (gk << 8), j, j, k0});
                    int i = j + 1;
                    for (; i + 1 < nb; i += 2) t.push_back({5 | (gk << 8), i, j, k0});
                    if (i < nb) { t.push_back({2 | (gk << 8), i, j, k0}); t.push_back({2 | (gk << 8), i, j + 1, k0}); }
                    ++j;
                    continue;
                }
                grouped(j, j, gk, k0);
            }
        }
    }
}

// Block columns per far update and width of the near band, by size: measured in tools/prof_cholesky.py
static void chol_task_shape(int nb, int* gk, int* near) {
    // measured (profiles/r03_cholesky_task_shapes.txt): N = 2000: (4,4) 0.585 ms, (8,4) 0.578, (16,3) 0.627; N = 3072: (4,2) 0.98,
    // (16,3) 0.96; N = 5000: (4,2) 2.18, (8,4) 2.06, (16,4) 1.98; N = 8192: (16,3) 5.87, (32,3) 6.19; N = 10000: (8,2) 10.3, (16,3) 9.91
    *gk = nb < 40 ? 4 : nb < 64 ? 8 : 16; *near = nb < 100 ? 4 : 3;
    if (const char* e = getenv("ALABI_CHOL_GK")) { const int v = atoi(e); if (v >= 1 && v <= 64) *gk = v; }
    if (const char* e = getenv("ALABI_CHOL_NEAR")) { const int v = atoi(e); if (v >= 1 && v <= 16) *near = v; }
}

static bool chol_tasks_w8(int nb) {                       // eight waves per workgroup (chol_tasks8_kernel)?
    bool w8 = nb >= ALABI_CHOL_W8_MIN_NB;
    if (const char* e3 = getenv("ALABI_CHOL_W8")) w8 = e3[0] == '1';
    return w8;
}
static bool chol_tasks_two(int nb, int gk) {              // grouped updates of two tiles per task (UPDATE2; eight-wave kernel, gk >= 2)
    bool two = chol_tasks_w8(nb) && gk >= 2;
    if (const char* e = getenv("ALABI_CHOL_UPDATE2")) two = two && e[0] != '0';
    return two;
}

extern "C" int alabi_debug_chol_tasks(int nb, int* out, int cap) {   // host only: the list as (type, i, j, k) quadruples; returns the count
    int gk, near;
    chol_task_shape(nb, &gk, &near);
    std::vector<CholTask> t;
    chol_build_tasks(nb, gk, near, chol_tasks_two(nb, gk), t);
    if (out)
        for (size_t q = 0; q < t.size() && (int)q < cap; ++q) { out[4 * q] = t[q].type; out[4 * q + 1] = t[q].i; out[4 * q + 2] = t[q].j; out[4 * q + 3] = t[q].k; }
    return (int)t.size();
}

// The list on the device, built once per (device, nb, shape) and kept.
static int chol_task_list(int nb, const CholTask** dev, int* count) {
    static std::mutex mu;
    static std::map<std::array<int, 4>, std::pair<CholTask*, int>> cache;
    int device = 0, gk, near;
    ALABI_HIP_CHECK(hipGetDevice(&device));
    chol_task_shape(nb, &gk, &near);
    const bool two = chol_tasks_two(nb, gk);
    std::lock_guard<std::mutex> lk(mu);
    const char* e4 = getenv("ALABI_CHOL_UPDATE4");
    const std::array<int, 4> key{device, nb, gk, near + (two ? 64 : 0) + (e4 ? (e4[0] == '1' ? 128 : 256) : 0)};
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<CholTask> t;
        chol_build_tasks(nb, gk, near, two, t);
        CholTask* d = nullptr;
        ALABI_HIP_CHECK(hipMalloc(&d, t.size() * sizeof(CholTask)));
        ALABI_HIP_CHECK(hipMemcpy(d, t.data(), t.size() * sizeof(CholTask), hipMemcpyHostToDevice));
        it = cache.emplace(key, std::make_pair(d, (int)t.size())).first;
    }
    *dev = it->second.first; *count = it->second.second;
    return ALABI_OK;
}

// 1 when the queue kernel was launched (the caller reads gp->chol_ctl[1] after its synchronisation: non-zero = a wait ran out,
// the matrix is in an undefined state and must be assembled and factorised again on the launch-per-step path).
// Will the task queue factorise this matrix?  If so its control words exist and *ctl_ints says how many the assembly kernel
// has to clear (queue head, time-out flag, tile versions, slab counters).
int cholesky_tasks_prepare(alabi_gp* gp, hipStream_t s, int* ctl_ints_out) {
    *ctl_ints_out = 0;
    const int nb = gp->Npad / 64;
    // Default for 3..256 block columns (N = 129..16384; ALABI_CHOL_TASKS=0 forces it off).  With the eight-wave
    // kernel of round 3 (from 40 block columns on): N = 5000 1.79 ms, 8192 5.08, 10000 8.5 (39.2 TFLOP/s), 11000 10.9 (panels of 8:
    // 12.6), 12000 13.6 (15.4), 14000 20.6 (22.9), 16000 29.7 (30.6) -- profiles/r03_cholesky_w8_vs_w4.txt.  Before that: measured
    // (tools/prof_chol_tasks.py, assembly included; round 3, updates over groups of block columns): N = 1024 0.33 ms (0.46 launch
    // per step), 2000 0.59 (0.89), 3072 0.96 (1.39), 4096 1.40 (2.05), 5000 2.0 (2.9), 6000 2.86 (3.96), 8192 5.9 (6.9),
    // 10000 9.9 (10.8 panels of 8), 12000 15.9 (15.4), 16000 35.1 (30.5): from 11000 on the rank-512 panel path is ahead.
    const char* env = getenv("ALABI_CHOL_TASKS");
    const bool forced_on = env && env[0] == '1', forced_off = env && env[0] == '0';
    // From 3 block columns on since the end of round 4 (tools/prof_chol_small.py, queue vs launch per step: N = 192 0.092 vs 0.112 ms, 512 0.155 vs
    // 0.230, 960 0.249 vs 0.408 -- with the matrix-core panel solves the queue wins at every size; before, it started at 16 block columns).
    if (nb < 3 || nb > 256 || forced_off || (!forced_on && nb > ALABI_CHOL_TASKS_MAX_NB)) return ALABI_OK;
    const size_t ctl_ints = 2 + (size_t)nb * nb + nb + 130;            // + 130: alignment + phase timers of an ALABI_CHOL_PROF build
    // behind the control words (8-byte aligned; filled with the tag by the assembly kernel): the slab buffers, [nb][4][64][16] doubles
    auto total_ints = [](size_t b) { return ((2 + b * b + b + 130 + 1) & ~(size_t)1) + b * 8192; };
    if (gp->chol_ctl_ints < total_ints(nb)) {
        if (gp->chol_ctl) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); ALABI_HIP_CHECK(hipFree(gp->chol_ctl)); gp->chol_ctl = nullptr; }
        const size_t cap_nb = gp->n_cap / 64 < 256 ? gp->n_cap / 64 : 256;
        const size_t cap = total_ints(cap_nb), need = total_ints(nb);
        ALABI_HIP_CHECK(hipMalloc(&gp->chol_ctl, (cap > need ? cap : need) * sizeof(int)));
        gp->chol_ctl_ints = cap > need ? cap : need;
    }
    *ctl_ints_out = (int)ctl_ints;
    return ALABI_OK;
}

// 1 when the queue kernel was launched (the caller reads gp->chol_ctl[1] after its synchronisation: non-zero = a wait ran out,
// the matrix is in an undefined state and must be assembled and factorised again on the launch-per-step path).  The control
// words and gp->info were cleared by launch_assemble(gp, s, ctl_ints).
int launch_cholesky_tasks(alabi_gp* gp, hipStream_t s, int* launched) {
    *launched = 0;
    const int ld = gp->Npad, nb = gp->Npad / 64;
    const CholTask* tasks = nullptr;
    int ntasks = 0, st;
    if ((st = chol_task_list(nb, &tasks, &ntasks)) != ALABI_OK) return st;
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = ntasks < n_cu ? ntasks : n_cu;
    int spin = 1 << 18;
    if (const char* e2 = getenv("ALABI_CHOL_SPIN_LIMIT")) { const int v = atoi(e2); if (v > 0) spin = v; }
    // eight waves per workgroup from ALABI_CHOL_W8_MIN_NB block columns on (ALABI_CHOL_W8=0 / 1 forces four / eight)
    const bool w8 = chol_tasks_w8(nb);
    if (w8) hipLaunchKernelGGL(chol_tasks8_kernel, dim3(grid), dim3(512), 0, s, gp->L, ld, nb, tasks, ntasks, gp->chol_ctl, gp->info, gp->dinv, spin);
    else hipLaunchKernelGGL(chol_tasks_kernel, dim3(grid), dim3(256), 0, s, gp->L, ld, nb, tasks, ntasks, gp->chol_ctl, gp->info, gp->dinv, spin);
    ALABI_LAUNCH_CHECK();
#ifdef ALABI_CHOL_PROF
    {
        long long h[24];
        (void)hipMemcpyAsync(h, gp->chol_ctl + ((2 + nb * nb + nb + 1) & ~1), sizeof(h), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        for (int q = 8; q <= 16; q += 8)
            if (h[q + 4] > 0)
                fprintf(stderr, "[chol_tasks_kernel] per %s UPDATE (us): wait for deps %.2f, first fetch + C %.2f, loop %.2f (%.2f per block column), C store + publish %.2f (n=%lld, %.2f columns each)\n",
                        q == 8 ? "grouped" : "single-column", 0.01 * h[q] / h[q + 4], 0.01 * h[q + 1] / h[q + 4], 0.01 * h[q + 2] / h[q + 4],
                        0.01 * h[q + 2] / (h[q + 5] ? h[q + 5] : 1), 0.01 * h[q + 3] / h[q + 4], h[q + 4], (double)h[q + 5] / h[q + 4]);
        {
            long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpyFromSymbol(pp, HIP_SYMBOL(g_potrf_prof), sizeof(pp));
            if (pp[1] > 0) fprintf(stderr, "[chol_tasks_kernel] slab recurrence of the diagonal factorisation (wave 0): %.2f us per 16 pivots (n=%lld)\n", 0.01 * pp[0] / pp[1], pp[1]);
            if (pp[5] > 0) fprintf(stderr, "[chol_tasks_kernel] per factorisation (us): slabs 0-2 start -> barrier A %.2f each, A -> B %.2f each, last slab incl. its stores %.2f (n=%lld)\n",
                                   0.01 * pp[2] / (3 * pp[5]), 0.01 * pp[3] / (3 * pp[5]), 0.01 * pp[4] / pp[5], pp[5]);
            long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_potrf_prof), z, sizeof(z));
        }
        if (h[6] > 0)
            fprintf(stderr, "[chol_tasks_kernel] per CHAIN (us): wait %.2f loads %.2f trsm %.2f store+publish %.2f mfma %.2f potrf %.2f store+publish %.2f (n=%lld)\n",
                    0.01 * h[7] / h[6], 0.01 * h[0] / h[6], 0.01 * h[1] / h[6], 0.01 * h[2] / h[6], 0.01 * h[3] / h[6], 0.01 * h[4] / h[6],
                    0.01 * h[5] / h[6], h[6]);
    }
#endif
#ifdef ALABI_CHOL_LOG
    if (getenv("ALABI_CHOL_LOG_PRINT")) {
        static long long h[256][32];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_chain_log), sizeof(h));
        const char* name[15] = {"drawn -> deps met", "deps met -> tiles in LDS", "tiles in LDS -> slab 0 seen", "slab 0 -> slab 1 seen", "slab 1 -> slab 2 seen",
                                "slab 2 -> slab 3 seen", "slab 3 seen -> solve + diag update done", "-> panel tile published", "-> factorisation starts",
                                "-> recurrence 0 done", "-> recurrence 1 done", "-> recurrence 2 done", "-> recurrence 3 done", "-> last inverse block out",
                                "-> tile stored, published"};
        const int k0 = nb / 4, k1 = nb - 2;
        fprintf(stderr, "[chain log] nb = %d, means over CHAIN(%d..%d), us:\n", nb, k0, k1);
        for (int i = 0; i < 15; ++i) {
            double sum = 0;
            for (int k = k0; k <= k1; ++k) sum += 0.01 * (double)(h[k][i + 1] - h[k][i]);
            fprintf(stderr, "  %-44s %7.2f\n", name[i], sum / (k1 - k0 + 1));
        }
        {
            double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = k0; k <= k1; ++k) {
                const long long r3 = h[k - 1][13];              // the producer's last recurrence done
                t[0] += 0.01 * (double)(h[k][3] - r3); t[1] += 0.01 * (double)(h[k][4] - r3); t[2] += 0.01 * (double)(h[k][5] - r3); t[3] += 0.01 * (double)(h[k][6] - r3);
                t[4] += 0.01 * (double)(h[k][7] - r3); t[5] += 0.01 * (double)(h[k][9] - r3); t[6] += 0.01 * (double)(h[k - 1][14] - r3); t[7] += 0.01 * (double)(h[k][2] - r3);
            }
            const double n_ = k1 - k0 + 1;
            fprintf(stderr, "  relative to the END of the previous CHAIN's last recurrence: tiles in LDS %+.2f; slab 0 / 1 / 2 / 3 in LDS %+.2f %+.2f %+.2f %+.2f; solve + diag update done %+.2f; "
                            "factorisation starts %+.2f (the previous CHAIN's last inverse block drained at %+.2f)\n", t[7] / n_, t[0] / n_, t[1] / n_, t[2] / n_, t[3] / n_, t[4] / n_, t[5] / n_, t[6] / n_);
        }
        for (int sl = 0; sl < 3; ++sl) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int k = k0; k <= k1; ++k) { a0 += 0.01 * (double)(h[k][16 + 4 * sl] - h[k][10 + sl]); a1 += 0.01 * (double)(h[k][17 + 4 * sl] - h[k][16 + 4 * sl]); a2 += 0.01 * (double)(h[k][18 + 4 * sl] - h[k][17 + 4 * sl]); a3 += 0.01 * (double)(h[k][19 + 4 * sl] - h[k][18 + 4 * sl]); }
            const double n_ = k1 - k0 + 1;
            fprintf(stderr, "  slab %d, storing wave 7: sees the recurrence done after %.2f, LDS reads %.2f, stores issued %.2f, drained %.2f\n", sl, a0 / n_, a1 / n_, a2 / n_, a3 / n_);
        }
        double per = 0, hop = 0;
        for (int k = k0; k <= k1; ++k) { per += 0.01 * (double)(h[k + 1][9] - h[k][9]); hop += 0.01 * (double)(h[k + 1][6] - h[k][14]); }
        fprintf(stderr, "  period (factorisation start to start) %.2f; last inverse block out -> seen by the next CHAIN %.2f\n", per / (k1 - k0 + 1), hop / (k1 - k0 + 1));
    }
#endif
    *launched = 1;
    return ALABI_OK;
}

// The launch-per-step factorisation of ONE [Npad, Npad] matrix (any owner: a GP handle, a slot of a batch workspace).
int launch_cholesky_steps(double* L, int Npad, int* info, double* dinv, hipStream_t s) {
    const int ld = Npad, nb = Npad / 64;
    // (info was cleared by the assembly kernel, which always runs just before)
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(64), 0, s, L, ld, 0, info, dinv);
    // Block columns per panel (0: rank-64 updates of the whole trailing matrix).  Measured on MI355X (tools/prof_cholesky.py):
    // the panel path wins from about N = 8000 on (N = 10000: 14.3 -> 11.3 ms, N = 16000: 49.6 -> 31.4 ms with panels of 8;
    // panels of 4: 11.8 / 33.0 ms); below that the extra launches per panel cost more than the trailing traffic they save
    // (N = 5000: 3.2 vs 3.9 ms).
    int panel = nb >= 110 ? 8 : 0;                          // N >= 7040 (N = 7000: 5.33 vs 5.54 ms rank-64, 7500: 6.05 vs 6.59; 6500: 4.80 vs 4.73)
    if (const char* env = getenv("ALABI_CHOL_PANEL")) { const int v = atoi(env); if (v == 0 || v == 2 || v == 4 || v == 6 || v == 8) panel = v; }
    if (panel == 0) {
        for (int kb = 0; kb + 1 < nb; ++kb) {
            const int T = nb - kb - 1;
            hipLaunchKernelGGL(trsm_panel_kernel, dim3(T), dim3(256), 0, s, L, ld, kb, dinv);
            hipLaunchKernelGGL(syrk_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, s, L, ld, kb, info, dinv, 0, T);   // + potrf of block kb+1
        }
        ALABI_LAUNCH_CHECK();
        return ALABI_OK;
    }
    // Panels of `panel` block columns.  Inside a panel: trsm of block column kb, rank-64 update of the REST OF THE PANEL only
    // (with the factorisation of the next diagonal block fused in).  Behind it: one rank-(64 panel) update of the trailing
    // matrix, split for look-ahead -- the tile columns of the next panel on the caller's stream (the next panel's chain of
    // small launches waits only for them), everything further right on a second stream, overlapping that chain.
    // (per host thread: the cross-validation search factorises on several threads and streams at once)
    static thread_local hipStream_t side = nullptr;
    static thread_local hipEvent_t ev_panel[2] = {nullptr, nullptr}, ev_rest[2] = {nullptr, nullptr};
    const char* la_env = getenv("ALABI_CHOL_LOOKAHEAD");
    const bool lookahead = !(la_env && la_env[0] == '0');
    if (lookahead && !side) {
        // lowest priority: the bulk update fills every CU (two workgroups each); whenever one of them retires, the waiting
        // workgroups of the next panel's chain on the caller's stream are dispatched first
        int prio_least = 0, prio_greatest = 0;
        ALABI_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        ALABI_HIP_CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio_least));
        for (int i = 0; i < 2; ++i) {
            ALABI_HIP_CHECK(hipEventCreateWithFlags(&ev_panel[i], hipEventDisableTiming));
            ALABI_HIP_CHECK(hipEventCreateWithFlags(&ev_rest[i], hipEventDisableTiming));
        }
    }
    bool rest_pending = false;
    int pi = 0;
    for (int p0 = 0; p0 < nb; p0 += panel, ++pi) {
        const int pe = p0 + panel < nb ? p0 + panel : nb;
        for (int kb = p0; kb < pe; ++kb) {
            const int T = nb - kb - 1;
            if (T == 0) break;
            hipLaunchKernelGGL(trsm_panel_kernel, dim3(T), dim3(256), 0, s, L, ld, kb, dinv);
            const int jc = pe - kb - 1;
            if (jc > 0) {
                int tiles = 0;
                for (int tj = 0; tj < jc; ++tj) tiles += T - tj;
                hipLaunchKernelGGL(syrk_update_kernel, dim3(tiles), dim3(256), 0, s, L, ld, kb, info, dinv, jc, T);
            }
        }
        if (pe >= nb) break;
        const int row0 = 64 * pe, ntr = (Npad - row0 + 127) / 128, kp = 64 * (pe - p0);
        const int next_tc = (panel * 64) / 128;          // tile columns that make up the next panel
        const int tc_split = next_tc < ntr ? next_tc : ntr;
        if (lookahead) {
            // the rest of the PREVIOUS panel's update touched the tiles we are about to update: wait for it
            if (rest_pending) ALABI_HIP_CHECK(hipStreamWaitEvent(s, ev_rest[(pi + 1) & 1], 0));
            ALABI_HIP_CHECK(hipEventRecord(ev_panel[pi & 1], s));                       // panel pi is final
            hipLaunchKernelGGL(syrk_panel_kernel<true>, dim3(tiles_in_cols(ntr, 0, tc_split)), dim3(256), 0, s, L, ld, Npad, 64 * p0, kp,
                               row0, 0, tc_split, ntr, info, dinv);
            if (tc_split < ntr) {
                ALABI_HIP_CHECK(hipStreamWaitEvent(side, ev_panel[pi & 1], 0));
                hipLaunchKernelGGL(syrk_panel_kernel<false>, dim3(tiles_in_cols(ntr, tc_split, ntr)), dim3(256), 0, side, L, ld, Npad,
                                   64 * p0, kp, row0, tc_split, ntr, ntr, info, (double*)nullptr);
                ALABI_HIP_CHECK(hipEventRecord(ev_rest[pi & 1], side));
                rest_pending = true;
            } else {
                rest_pending = false;
            }
        } else {
            hipLaunchKernelGGL(syrk_panel_kernel<true>, dim3(tiles_in_cols(ntr, 0, ntr)), dim3(256), 0, s, L, ld, Npad, 64 * p0, kp, row0,
                               0, ntr, ntr, info, dinv);
        }
    }
    if (lookahead && rest_pending) ALABI_HIP_CHECK(hipStreamWaitEvent(s, ev_rest[(pi + 1) & 1], 0));
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_cholesky(alabi_gp* gp, hipStream_t s) { return launch_cholesky_steps(gp->L, gp->Npad, gp->info, gp->dinv, s); }

// ---------------------------------------------------------------------------------------------------------------------
// Batched task queue: B independent matrices in ONE launch (gp_batch.hip: the folds x candidates of the hyper-parameter search).
// A single matrix of 16..40 block columns leaves the chip idle -- its time is the chain of nb CHAIN tasks, ~17 us each, with 8 %
// matrix-core duty at N = 2000 -- so the task lists of many matrices are interleaved: the chains of different matrices run side by
// side on different workgroups and the bulk updates of one fill the gaps of another.
//   * matrix b goes to list b % nlists (one list per XCD: its tiles stay in that XCD's L2, its own head counter);
//   * inside a list the matrices advance in SLOTS: slot t holds step t - start_m of every matrix m of the list that is active,
//     start_m = floor(rank_m * stagger) -- with stagger = steps / P about P matrices per list are in flight at any time, in
//     different phases (the wide early steps of one beside the narrow late steps of another), and the working set stays
//     ~ P * nlists lower triangles instead of all B;
//   * within a slot the matrices closest to their end come first (their steps are short and chain-bound).
// Each matrix keeps the order of its own list, so every list remains a topological order.
// (Round 4, measured and not kept -- tools/experiments/chol_batch_trsm3_tasks.patch: two or three panel tiles per TRSM task, one wave's
// slab recurrence per tile side by side, bit-identical: 500 matrices of N = 1600 28.5-29.2 ms with one tile per task in that build,
// 27.6-27.9 with two / three -- but the build without the extra task type does 27.5-28.0: the new code path costs the kernel 15 more
// spilled registers, which takes back what the TRSM tasks gain; N = 8000 4.50 -> 4.38 ms per fit.)
// ALABI_BATCH_GK: block columns per group, ALABI_BATCH_LEFT=0: the single-matrix list instead.  Measured (tools/prof_batch_cv.py, 500
// matrices of N = 1600 per call, everything included): the single-matrix list 46.6 ms; this one 34.3 (gk 4), 31.9 (8), 31.8 (10), 32.1 (12),
// 35.5 (16), 36.9 (32 = left-looking) -- profiles/r04_batch_sweeps.txt
static int chol_batch_gk() {
    int gk = 10;
    if (const char* e = getenv("ALABI_BATCH_GK")) { const int v = atoi(e); if (v >= 1 && v <= 255) gk = v; }
    return gk;
}
static bool chol_batch_left() { const char* e = getenv("ALABI_BATCH_LEFT"); return !(e && e[0] == '0'); }
static void chol_batch_shape(int nb, std::vector<CholTask>& t) {
    if (chol_batch_left()) {
        const char* e4 = getenv("ALABI_CHOL_UPDATE4");
        chol_build_tasks_batch(nb, chol_batch_gk(), true, !(e4 && e4[0] == '0'), t);
        return;
    }
    int gk, near;
    chol_task_shape(nb, &gk, &near);
    chol_build_tasks(nb, gk, near, chol_tasks_two(nb, gk), t);
}
static int chol_batch_build(const std::vector<int>& nbs, int nlists, int window, std::vector<CholTask>& out, std::vector<int>& list_off) {
    std::map<int, std::pair<std::vector<CholTask>, std::vector<int>>> per_nb;       // nb -> (tasks, first task of every step)
    for (int nb : nbs) {
        if (per_nb.count(nb)) continue;
        auto& e = per_nb[nb];
        chol_batch_shape(nb, e.first);
        for (size_t q = 0; q < e.first.size(); ++q)
            if ((e.first[q].type & 255) == 0) e.second.push_back((int)q);            // a CHAIN task opens a step
        e.second.push_back((int)e.first.size());
    }
    out.clear();
    list_off.assign(nlists + 1, 0);
    for (int q = 0; q < nlists; ++q) {
        list_off[q] = (int)out.size();
        std::vector<int> mem;                                                        // matrices of this list
        for (int b = q; b < (int)nbs.size(); b += nlists) mem.push_back(b);
        if (mem.empty()) continue;
        std::vector<int> start(mem.size());
        int last_slot = 0;
        for (size_t r = 0; r < mem.size(); ++r) {
            const int steps = nbs[mem[r]];
            const double stagger = window > 0 ? (double)steps / window : 0.0;
            start[r] = (int)(r * stagger);
            if (start[r] + steps > last_slot) last_slot = start[r] + steps;
        }
        // Inside a slot the tasks go PHASE by phase over the active matrices -- every matrix's CHAIN, then every matrix's panel solves,
        // then every matrix's catch-up updates (what the NEXT step's CHAIN and panel solves read), then every matrix's grouped updates
        // -- not matrix by matrix: the updates of a step wait for the CHAIN and the panel solves of the SAME step, drawn moments before
        // them, and the next CHAIN waits for this step's catch-ups; with the other matrices' tasks in between, a task's inputs are
        // finished when a workgroup reaches it instead of holding that workgroup for up to a CHAIN's 17 us (ALABI_BATCH_PHASES=0:
        // matrix by matrix; =3: without the split of the updates).  A matrix's own tasks keep their order (its step is CHAIN, panel
        // solves, catch-ups, groups in that order already).
        const char* pe = getenv("ALABI_BATCH_PHASES");
        const int nphase = (pe && pe[0] == '0') ? 1 : ((pe && pe[0] == '3') || !chol_batch_left()) ? 3 : 4;
        for (int t = 0; t < last_slot; ++t)
            for (int phase = 0; phase < nphase; ++phase)
                for (size_t r = 0; r < mem.size(); ++r) {                           // lower rank = started earlier = closer to its end
                    const int b = mem[r], st = t - start[r];
                    if (st < 0 || st >= nbs[b]) continue;
                    const auto& e = per_nb[nbs[b]];
                    for (int x = e.second[st]; x < e.second[st + 1]; ++x) {
                        CholTask c = e.first[x];
                        const int ty = c.type & 255, k = st - 1;                    // step st factorises block column k + 1 = st
                        int ph = ty == 0 ? 0 : ty == 1 ? 1 : 2;
                        if (ph == 2 && nphase == 4 && !(c.j == k + 1 || (c.i == c.j && c.j == k + 2))) ph = 3;   // not a catch-up: a grouped update
                        if (nphase > 1 && ph != phase) continue;
                        c.type |= b << 16;
                        out.push_back(c);
                    }
                }
    }
    list_off[nlists] = (int)out.size();
    return (int)out.size();
}

extern "C" int alabi_debug_chol_batch_matrix_tasks(int nb, int* out, int cap) {   // host only: ONE matrix's list inside a batch
    std::vector<CholTask> t;
    chol_batch_shape(nb, t);
    if (out)
        for (size_t q = 0; q < t.size() && (int)q < cap; ++q) { out[4 * q] = t[q].type; out[4 * q + 1] = t[q].i; out[4 * q + 2] = t[q].j; out[4 * q + 3] = t[q].k; }
    return (int)t.size();
}

extern "C" int alabi_debug_chol_batch_tasks(int B, const int* nbs, int nlists, int window, int* out, int cap, int* list_off_out) {
    std::vector<int> v(nbs, nbs + B), lo;
    std::vector<CholTask> t;
    chol_batch_build(v, nlists, window, t, lo);
    if (out)
        for (size_t q = 0; q < t.size() && (int)q < cap; ++q) { out[4 * q] = t[q].type; out[4 * q + 1] = t[q].i; out[4 * q + 2] = t[q].j; out[4 * q + 3] = t[q].k; }
    if (list_off_out) for (int q = 0; q <= nlists; ++q) list_off_out[q] = lo[q];
    return (int)t.size();
}

void chol_batch_free(CholBatchQueue& q) {
    if (q.tasks) (void)hipFree(q.tasks);
    if (q.list_off) (void)hipFree(q.list_off);
    if (q.mats) (void)hipFree(q.mats);
    if (q.ctl) (void)hipFree(q.ctl);
    if (q.linv) (void)hipFree(q.linv);
    q = CholBatchQueue{};
}

// Queue B matrices (slot b: A[b] [ld[b], ld[b]] row-major with ld a multiple of 64, dinv[b] [ld[b]], info[b] [1], all device memory):
// builds (or reuses) the interleaved task list, uploads the matrix table, clears the control words -- everything on `s`.
int chol_batch_prepare(CholBatchQueue& q, int B, const int* ld, double* const* A, double* const* dinv, int* const* info, hipStream_t s) {
    if (B <= 0 || B >= 32768) return ALABI_BAD_ARGUMENT;
    int nlists = 8, window = 8;      // measured: window 2 45 ms, 3 40, 4 35.6, 5 33.2, 8 31.8, 0 (all at once) 31.4-32.4 per 500 matrices of N = 1600
    if (const char* e = getenv("ALABI_BATCH_LISTS")) { const int v = atoi(e); if (v >= 1 && v <= 8) nlists = v; }
    if (const char* e = getenv("ALABI_BATCH_WINDOW")) { const int v = atoi(e); if (v >= 0 && v <= 4096) window = v; }
    if (nlists > B) nlists = B;
    while (nlists > 1 && B < 4 * nlists && B % nlists != 0) nlists /= 2;   // few (large) matrices: equal shares per list (12 matrices of N = 8000: 8 lists 5.7 ms per fit, 4 lists 5.2)
    std::vector<int> nbs(B);
    size_t ver_ints = 0;
    for (int b = 0; b < B; ++b) {
        if (ld[b] <= 0 || ld[b] % 64 != 0 || ld[b] / 64 > 256) return ALABI_BAD_ARGUMENT;
        nbs[b] = ld[b] / 64;
        ver_ints += (size_t)nbs[b] * nbs[b] + nbs[b];
    }
    int shape_sig = 0;                                                    // the switches that shape a matrix's own list (tools: env sweeps)
    {
        int gk, near;
        chol_task_shape(nbs[0], &gk, &near);
        const char* e4 = getenv("ALABI_CHOL_UPDATE4");
        shape_sig = (chol_batch_left() ? 1 << 20 : 0) + ((getenv("ALABI_BATCH_PHASES") ? (getenv("ALABI_BATCH_PHASES")[0] & 7) : 7) << 21) + chol_batch_gk() * 4096 + gk * 64 + near * 4 + (e4 ? (e4[0] == '1' ? 1 : 3) : 0);
    }
    if (!(q.tasks && q.nbs == nbs && q.nlists == nlists && q.window == window && q.shape_sig == shape_sig)) {
        std::vector<CholTask> t;
        std::vector<int> lo;
        chol_batch_build(nbs, nlists, window, t, lo);
        if (t.size() > q.tasks_cap) {
            if (q.tasks) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(q.tasks); q.tasks = nullptr; q.tasks_cap = 0; }
            ALABI_HIP_CHECK(hipMalloc(&q.tasks, t.size() * sizeof(CholTask)));
            q.tasks_cap = t.size();
        }
        if (!q.list_off) ALABI_HIP_CHECK(hipMalloc(&q.list_off, 9 * sizeof(int)));
        ALABI_HIP_CHECK(hipStreamSynchronize(s));                         // a launch still reading the previous list
        ALABI_HIP_CHECK(hipMemcpy(q.tasks, t.data(), t.size() * sizeof(CholTask), hipMemcpyHostToDevice));
        ALABI_HIP_CHECK(hipMemcpy(q.list_off, lo.data(), (nlists + 1) * sizeof(int), hipMemcpyHostToDevice));
        q.ntasks = (int)t.size(); q.nbs = nbs; q.nlists = nlists; q.window = window; q.shape_sig = shape_sig;
    }
    if ((size_t)B > q.mats_cap) {
        if (q.mats) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(q.mats); q.mats = nullptr; }
        ALABI_HIP_CHECK(hipMalloc(&q.mats, (size_t)B * sizeof(CholMat)));
        q.mats_cap = B;
    }
    const size_t ctl_ints = 256 + ver_ints;
    if (ctl_ints > q.ctl_cap) {
        if (q.ctl) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(q.ctl); q.ctl = nullptr; }
        ALABI_HIP_CHECK(hipMalloc(&q.ctl, ctl_ints * sizeof(int)));
        q.ctl_cap = ctl_ints;
    }
    q.ctl_ints = ctl_ints;
    size_t linv_doubles = 0;                                              // the slab buffers, [nb][4][64][16] per matrix (not cleared: the batch polls the slab counters)
    for (int b = 0; b < B; ++b) linv_doubles += (size_t)nbs[b] * 4096;
    if (linv_doubles > q.linv_cap) {
        if (q.linv) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(q.linv); q.linv = nullptr; }
        ALABI_HIP_CHECK(hipMalloc(&q.linv, linv_doubles * sizeof(double)));
        q.linv_cap = linv_doubles;
    }
    std::vector<CholMat> hm(B);
    size_t off = 256, loff = 0;
    for (int b = 0; b < B; ++b) {
        hm[b].A = A[b]; hm[b].dinv = dinv[b]; hm[b].info = info[b]; hm[b].ld = ld[b]; hm[b].nb = nbs[b];
        hm[b].ver = q.ctl + off; hm[b].sver = q.ctl + off + (size_t)nbs[b] * nbs[b];
        hm[b].linv = q.linv + loff;
        off += (size_t)nbs[b] * nbs[b] + nbs[b];
        loff += (size_t)nbs[b] * 4096;
    }
    ALABI_HIP_CHECK(hipMemcpyAsync(q.mats, hm.data(), (size_t)B * sizeof(CholMat), hipMemcpyHostToDevice, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));                             // `hm` is a local
    ALABI_HIP_CHECK(hipMemsetAsync(q.ctl, 0, ctl_ints * sizeof(int), s));
    q.B = B;
    return ALABI_OK;
}

// One launch of the batched queue; the time-out flag is q.ctl[1] (read it after synchronising: non-zero = undefined matrices).
int chol_batch_launch(CholBatchQueue& q, hipStream_t s) {
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = q.ntasks < n_cu ? q.ntasks : n_cu;
    int spin = 1 << 18;
    if (const char* e2 = getenv("ALABI_CHOL_SPIN_LIMIT")) { const int v = atoi(e2); if (v > 0) spin = v; }
    hipLaunchKernelGGL(chol_tasks8_batch_kernel, dim3(grid), dim3(512), 0, s, q.mats, q.tasks, q.ntasks, q.list_off, q.nlists, q.ctl, spin);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
