// Append ONE training point to an existing factorisation (rank-1 extension of the Cholesky factor and of the cached L^-1).
//
// Replaces the from-scratch refit the reference does after every active-learning iteration (alabi/core.py:1780 -> _fit_gp ->
// gp.compute, core.py:1097-1160) when the kernel hyper-parameters are the carried ones: with K' = [[K, k], [k^T, kss]],
//     l = L^-1 k = W k,   l_nn = sqrt(kss - |l|^2),   L' = [[L, 0], [l^T, l_nn]],   W' = [[W, 0], [-(l^T W) / l_nn, 1 / l_nn]],
// two matrix-vector products with the cached W = L^-1 instead of N^3/3 (+ the N^3/3 rebuild of W).  The new row takes the first
// padding row of the 64-aligned layout (identity there), so nothing moves; when the padding is used up the caller refits.
// Same arithmetic as the last row of a full factorisation up to the order of the sums.
#include "gp_device.hpp"

namespace alabi {

// k_i = amp f(r2(x_i, x_new)) for i < N (0 on the padding) -> kcol; the scaled coordinates of the new point -> Xt[:, N]
template <int D, bool GENERIC>
__global__ void __launch_bounds__(256)
append_kcol_kernel(double* __restrict__ Xt, int N, int Npad, const double* __restrict__ x_new, int d, DimVec inv_len,
                   double amp, KernelFn kf, double* __restrict__ kcol) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) q[k] = (k < d) ? x_new[k] * inv_len.v[k] : 0.0;
    if (i < Npad) {
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double df = Xt[(size_t)k * Npad + i] - q[k];
            r2 = fma(df, df, r2);
        }
        kcol[i] = (i < N) ? amp * radial<GENERIC>(r2, kf) : 0.0;
    }
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < D) Xt[(size_t)threadIdx.x * Npad + N] = q[threadIdx.x];
}

// l = W k for the rows of block rb: W tile-major, W[t][i][c] = L^-1[i][64 t + c] (exact zeros above the diagonal).
// Thread (row r = tid >> 2, quarter q = tid & 3) owns 16 consecutive columns of its row in every column tile: eight
// 16-byte loads per tile, four tiles in flight (a lone 8-byte load per iteration left the kernel at 140 GB/s).
__global__ void __launch_bounds__(256)
append_lrow_kernel(const double* __restrict__ W, const double* __restrict__ kcol, int N, int Npad, double* __restrict__ lrow) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int rb = blockIdx.x, r = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int i = rb * 64 + r;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll 4
    for (int t = 0; t <= rb; ++t) {
        const f64x2* wp = reinterpret_cast<const f64x2*>(W + ((size_t)t * Npad + i) * 64 + 16 * q);
        const f64x2* kp = reinterpret_cast<const f64x2*>(kcol + 64 * t + 16 * q);
        f64x2 wv[8], kv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { wv[e] = wp[e]; kv[e] = kp[e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) { a0 = fma(wv[e][0], kv[e][0], a0); a1 = fma(wv[e][1], kv[e][1], a1); }
    }
    double acc = a0 + a1;
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (q == 0) lrow[i] = (i < N) ? acc : 0.0;
}

// pivot of the new row; writes row N of L and dinv[N] when it is positive, reports LAPACK-style info otherwise
__global__ void __launch_bounds__(256)
append_pivot_kernel(double* __restrict__ L, double* __restrict__ dinv, const double* __restrict__ lrow, int N, int Npad,
                    double kss, int* __restrict__ info, double* __restrict__ scal) {
    __shared__ double scratch[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s = fma(lrow[i], lrow[i], s);
    s = block_sum(s, scratch);
    const double dd = kss - s;
    if (!(dd > 0.0)) {                                   // also NaN
        if (threadIdx.x == 0) { *info = N + 1; scal[0] = 0.0; }
        return;
    }
    const double lnn = sqrt(dd);
    for (int j = threadIdx.x; j < Npad; j += 256)
        L[(size_t)N * Npad + j] = (j < N) ? lrow[j] : (j == N ? lnn : 0.0);
    if (threadIdx.x == 0) { dinv[N] = 1.0 / lnn; scal[0] = lnn; *info = 0; }
}

// row N of W' for column tile t: -(l^T W)[64 t + c] / l_nn (c < N - 64 t), 1 / l_nn on the diagonal, 0 beyond.
// Thread (row group g = tid >> 2 of 64, quarter q) walks the rows 64 t + g, + 64, ... with 16 columns each (eight 16-byte
// loads per row, two rows in flight); the 64 row groups are then added per column through LDS in a fixed order.
__global__ void __launch_bounds__(256)
append_wrow_kernel(double* __restrict__ W, const double* __restrict__ lrow, int N, int Npad, const double* __restrict__ scal) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ double part[64][65];
    const double lnn = scal[0];
    if (!(lnn > 0.0)) return;                            // the pivot failed: leave W alone
    const int t = blockIdx.x, g = threadIdx.x >> 2, q = threadIdx.x & 3;
    double acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
#pragma unroll 2
    for (int i = 64 * t + g; i < N; i += 64) {
        const f64x2* wp = reinterpret_cast<const f64x2*>(W + ((size_t)t * Npad + i) * 64 + 16 * q);
        const double li = lrow[i];
        f64x2 wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = wp[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[2 * e] = fma(li, wv[e][0], acc[2 * e]); acc[2 * e + 1] = fma(li, wv[e][1], acc[2 * e + 1]); }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[g][16 * q + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        double u = 0.0;
#pragma unroll 8
        for (int gg = 0; gg < 64; ++gg) u += part[gg][c];
        const int col = 64 * t + c;
        W[((size_t)t * Npad + N) * 64 + c] = (col < N) ? -u / lnn : (col == N ? 1.0 / lnn : 0.0);
    }
}

int launch_append(alabi_gp* gp, const double* x_new, hipStream_t s) {
    const int N = gp->N, Npad = gp->Npad, nb = Npad / 64, db = dim_bucket(gp->d);
    const double amp = exp(gp->log_amp);
    // kss = k(x, x) + white noise; every kernel of the family has f(0) = 1
    const double kss = amp + exp(gp->log_wn);
    double* kcol = gp->work;                              // [Npad]
    double* lrow = gp->work + gp->n_cap;                  // [Npad]
    ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(gp->kf.type, hipLaunchKernelGGL((append_kcol_kernel<D, GENERIC>), dim3((Npad + 255) / 256),
        dim3(256), 0, s, gp->Xt, N, Npad, x_new, gp->d, gp->inv_len, amp, gp->kf, kcol)));
    hipLaunchKernelGGL(append_lrow_kernel, dim3(nb), dim3(256), 0, s, gp->winv, kcol, N, Npad, lrow);
    hipLaunchKernelGGL(append_pivot_kernel, dim3(1), dim3(256), 0, s, gp->L, gp->dinv, lrow, N, Npad, kss, gp->info, gp->red + 2);
    hipLaunchKernelGGL(append_wrow_kernel, dim3(nb), dim3(256), 0, s, gp->winv, lrow, N, Npad, gp->red + 2);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
