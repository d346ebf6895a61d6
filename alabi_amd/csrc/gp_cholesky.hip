// Blocked right-looking Cholesky (lower, in place, fp64) for gfx950.
//
// Replaces scipy.linalg.cholesky inside george's BasicSolver.compute, reached from the
// reference at alabi/core.py:1158 (every active-learning iteration), :1430, :1577 and
// alabi/gp_utils.py:243.  Work: N^3/3 flops; the dense trailing update runs on the fp64
// matrix cores (v_mfma_f64_16x16x4_f64) with both 64x64 panels staged in LDS.
//
// Per 64-column block step kb:
//   1. potrf_diag   : one workgroup factorises A[kb,kb] in LDS (reports a non-positive
//                     pivot as LAPACK's potrf `info`, 1-based).
//   2. trsm_panel   : A[i,kb] <- A[i,kb] * L_kk^-T for every block row i > kb.  One
//                     wavefront owns 16 rows; the 4 lanes of a row exchange the solved
//                     entry with a wave shuffle, so the sweep needs no barrier.
//   3. syrk_update  : A[i,j] -= A[i,kb] * A[j,kb]^T for kb < j <= i, one 64x64 tile per
//                     workgroup, 4 waves x (16 rows x 64 cols) x K=64 on MFMA.
// The matrix is [Npad, Npad] row-major with identity padding, so every block is full.
#include "common.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
potrf_diag_kernel(double* __restrict__ A, int ld, int kb, int* __restrict__ info) {
    __shared__ double a[64][65];
    const int tid = threadIdx.x;
    double* Ab = A + (size_t)(kb * 64) * ld + kb * 64;
    for (int e = tid; e < 4096; e += 256) {
        int r = e >> 6, c = e & 63;
        a[r][c] = (c <= r) ? Ab[(size_t)r * ld + c] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
    for (int j = 0; j < 64; ++j) {
        double djj = a[j][j];
        if (!(djj > 0.0)) {  // also true for NaN
            if (tid == 0) atomicCAS(info, 0, kb * 64 + j + 1);
            djj = 1.0;
        }
        double ljj = sqrt(djj);
        __syncthreads();
        if (tid == j) a[j][j] = ljj;
        if (tid > j && tid < 64) a[tid][j] = a[tid][j] / ljj;
        __syncthreads();
        // trailing update inside the block: a[i][k] -= a[i][j] * a[k][j], j < k <= i
        if (tx > j) {
            double akj = a[tx][j];
            for (int i = j + 1 + ty; i < 64; i += 4) {
                if (tx <= i) a[i][tx] = fma(-a[i][j], akj, a[i][tx]);
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < 4096; e += 256) {
        int r = e >> 6, c = e & 63;
        if (c <= r) Ab[(size_t)r * ld + c] = a[r][c];
    }
}

// X * L_kk^T = B for a strip of 16 rows; lane = 4*row + g owns columns c == g (mod 4).
__global__ void __launch_bounds__(64)
trsm_panel_kernel(double* __restrict__ A, int ld, int kb) {
    __shared__ double lkk[64][65];
    const int lane = threadIdx.x;
    const double* Lb = A + (size_t)(kb * 64) * ld + kb * 64;
    for (int e = lane; e < 4096; e += 64) {
        int r = e >> 6, c = e & 63;
        lkk[r][c] = (c <= r) ? Lb[(size_t)r * ld + c] : 0.0;
    }
    __syncthreads();
    const int rl = lane >> 2, g = lane & 3;
    double* row = A + (size_t)((kb + 1) * 64 + blockIdx.x * 16 + rl) * ld + kb * 64;
    double b[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) b[t] = row[4 * t + g];
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const int owner = c & 3, t = c >> 2;
        double x = b[t] / lkk[c][c];
        x = __shfl(x, (lane & ~3) | owner, 64);
        if (g == owner) b[t] = x;
#pragma unroll
        for (int t2 = 0; t2 < 16; ++t2) {
            if (4 * t2 + 3 > c) {  // compile-time prune; exact test below
                int c2 = 4 * t2 + g;
                if (c2 > c) b[t2] = fma(-x, lkk[c2][c], b[t2]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) row[4 * t + g] = b[t];
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
    __syncthreads();
    const int w = tid >> 6, l = tid & 63;
    const int lr = l & 15, lk = l >> 4;
    double* C = A + (size_t)(bi * 64 + 16 * w) * ld + bj * 64;
    v4f64 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[n][i] = C[(size_t)(lk + 4 * i) * ld + 16 * n + lr];
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
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, s, gp->L, ld, kb, gp->info);
        int T = nb - kb - 1;
        if (T > 0) {
            hipLaunchKernelGGL(trsm_panel_kernel, dim3(T * 4), dim3(64), 0, s, gp->L, ld, kb);
            hipLaunchKernelGGL(syrk_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, s, gp->L, ld, kb);
        }
    }
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
