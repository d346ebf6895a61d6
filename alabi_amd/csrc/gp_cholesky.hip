// Blocked right-looking Cholesky (lower, in place, fp64) for gfx950.
//
// Replaces scipy.linalg.cholesky inside george's BasicSolver.compute, reached from the
// reference at alabi/core.py:1158 (every active-learning iteration), :1430, :1577 and
// alabi/gp_utils.py:243.  Work: N^3/3 flops; the dense trailing update runs on the fp64
// matrix cores (v_mfma_f64_16x16x4_f64) with both 64x64 panels staged in LDS.
//
// Per 64-column block step kb:
//   1. potrf_diag   : ONE wavefront factorises A[kb,kb] left-looking: lane i keeps row i in
//                     registers, finished rows are published to LDS and re-read as broadcasts,
//                     so a column costs j independent FMAs + one sqrt/reciprocal and there is no
//                     workgroup barrier.  A non-positive pivot is reported as LAPACK's potrf
//                     `info` (1-based).  1/L_jj is kept (dinv) for every later triangular solve.
//   2. trsm_panel   : A[i,kb] <- A[i,kb] * L_kk^-T.  One lane owns one row (64 rows per wave):
//                     the substitution along the row needs no cross-lane traffic, L_kk is read
//                     from LDS as broadcasts, divisions are multiplications by dinv.
//   3. syrk_update  : A[i,j] -= A[i,kb] * A[j,kb]^T for kb < j <= i, one 64x64 tile per
//                     workgroup, 4 waves x (16 rows x 64 cols) x K=64 on MFMA.
// The matrix is [Npad, Npad] row-major with identity padding, so every block is full.
#include "common.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64)
potrf_diag_kernel(double* __restrict__ A, int ld, int kb, int* __restrict__ info, double* __restrict__ dinv) {
    __shared__ double Ls[64][65];
    const int lane = threadIdx.x;
    double* Ab = A + (size_t)(kb * 64) * ld + kb * 64;
    for (int r = 0; r < 64; ++r) Ls[r][lane] = Ab[(size_t)r * ld + lane];   // coalesced rows
    __syncthreads();
    double a[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) a[k] = Ls[lane][k];   // lane i <- row i (stride 65: conflict-free)
    __syncthreads();
    double my_rinv = 1.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        // v_i = A_ij - sum_{k<j} L_ik L_jk ; L_jk is row j, published in LDS at earlier columns
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const double ljk = Ls[j][k];
            if ((k & 3) == 0) s0 = fma(a[k], ljk, s0);
            else if ((k & 3) == 1) s1 = fma(a[k], ljk, s1);
            else if ((k & 3) == 2) s2 = fma(a[k], ljk, s2);
            else s3 = fma(a[k], ljk, s3);
        }
        const double v = a[j] - ((s0 + s1) + (s2 + s3));
        // pivot = v of lane j
        double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j),
                                      __builtin_amdgcn_readlane(__double2loint(v), j));
        if (!(piv > 0.0)) {  // also true for NaN
            if (lane == 0) atomicCAS(info, 0, kb * 64 + j + 1);
            piv = 1.0;
        }
        // 1/sqrt(piv) from the hardware estimate + two Newton steps (full fp64), then L_jj = piv * rinv with
        // one correction step: ~12 dependent ops instead of an IEEE sqrt followed by an IEEE division.
        double rinv = __builtin_amdgcn_rsq(piv);
        rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
        rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
        double ljj = piv * rinv;
        ljj = fma(0.5 * rinv, fma(-ljj, ljj, piv), ljj);
        rinv = fma(rinv, fma(-ljj, rinv, 1.0), rinv);   // rinv = 1 / L_jj to working precision
        a[j] = (lane == j) ? ljj : v * rinv;
        if (lane == j) my_rinv = rinv;
        Ls[lane][j] = a[j];          // rows >= j are final in column j (rows < j write unused upper entries)
        __syncthreads();             // single wave: orders the LDS write before the next column's reads
    }
    dinv[kb * 64 + lane] = my_rinv;
    for (int r = 0; r < 64; ++r)
        if (lane <= r) Ab[(size_t)r * ld + lane] = Ls[r][lane];
}

// X * L_kk^T = B for 64 rows: lane = row, serial along the row, no cross-lane traffic.
__global__ void __launch_bounds__(64)
trsm_panel_kernel(double* __restrict__ A, int ld, int kb, const double* __restrict__ dinv) {
    __shared__ double lkk[64][65];
    __shared__ double bs[64][65];
    __shared__ double di[64];
    const int lane = threadIdx.x;
    const double* Lb = A + (size_t)(kb * 64) * ld + kb * 64;
    double* Bb = A + (size_t)((kb + 1 + blockIdx.x) * 64) * ld + kb * 64;
    for (int r = 0; r < 64; ++r) {
        lkk[r][lane] = Lb[(size_t)r * ld + lane];
        bs[r][lane] = Bb[(size_t)r * ld + lane];
    }
    di[lane] = dinv[kb * 64 + lane];
    __syncthreads();
    double b[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) b[c] = bs[lane][c];
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const double x = b[c] * di[c];
        b[c] = x;
#pragma unroll
        for (int c2 = c + 1; c2 < 64; ++c2) b[c2] = fma(-x, lkk[c2][c], b[c2]);
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) bs[lane][c] = b[c];
    __syncthreads();
    for (int r = 0; r < 64; ++r) Bb[(size_t)r * ld + lane] = bs[r][lane];
}

// C[bi,bj] -= P[bi] * P[bj]^T with P[b] = A[b-block rows, kb-block cols].
__global__ void __launch_bounds__(256)
syrk_update_kernel(double* __restrict__ A, int ld, int kb) {
    __shared__ double Pi[64][66];
    __shared__ double Pj[64][66];
    int t = blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    int tj = t - ti * (ti + 1) / 2;
    const int bi = kb + 1 + ti, bj = kb + 1 + tj;
    const int tid = threadIdx.x;
    const double* Ai = A + (size_t)(bi * 64) * ld + kb * 64;
    const double* Aj = A + (size_t)(bj * 64) * ld + kb * 64;
    for (int e = tid; e < 4096; e += 256) {
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
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) C[(size_t)(lk + 4 * i) * ld + 16 * n + lr] = acc[n][i];
}

int launch_cholesky(alabi_gp* gp, hipStream_t s) {
    const int ld = gp->Npad, nb = gp->Npad / 64;
    ALABI_HIP_CHECK(hipMemsetAsync(gp->info, 0, sizeof(int), s));
    for (int kb = 0; kb < nb; ++kb) {
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(64), 0, s, gp->L, ld, kb, gp->info, gp->dinv);
        int T = nb - kb - 1;
        if (T > 0) {
            hipLaunchKernelGGL(trsm_panel_kernel, dim3(T), dim3(64), 0, s, gp->L, ld, kb, gp->dinv);
            hipLaunchKernelGGL(syrk_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, s, gp->L, ld, kb);
        }
    }
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
