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

// l = W k for the rows of block rb: W tile-major, W[t][i][c] = L^-1[i][64 t + c] (exact zeros above the diagonal)
__global__ void __launch_bounds__(256)
append_lrow_kernel(const double* __restrict__ W, const double* __restrict__ kcol, int N, int Npad, double* __restrict__ lrow) {
    const int rb = blockIdx.x, w = threadIdx.x >> 6, c = threadIdx.x & 63;
    for (int r = 0; r < 16; ++r) {
        const int i = rb * 64 + 16 * w + r;
        double acc = 0.0;
        if (i < N)
            for (int t = 0; t <= rb; ++t) acc = fma(W[((size_t)t * Npad + i) * 64 + c], kcol[64 * t + c], acc);
        acc = wave_sum_dpp(acc);
        if (c == 63) lrow[i] = (i < N) ? acc : 0.0;
    }
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

// row N of W' for column tile t: -(l^T W)[64 t + c] / l_nn (c < N - 64 t), 1 / l_nn on the diagonal, 0 beyond
__global__ void __launch_bounds__(256)
append_wrow_kernel(double* __restrict__ W, const double* __restrict__ lrow, int N, int Npad, const double* __restrict__ scal) {
    __shared__ double part[4][64];
    const double lnn = scal[0];
    if (!(lnn > 0.0)) return;                            // the pivot failed: leave W alone
    const int t = blockIdx.x, g = threadIdx.x >> 6, c = threadIdx.x & 63;
    double acc = 0.0;
    for (int i = 64 * t + g; i < N; i += 4) acc = fma(lrow[i], W[((size_t)t * Npad + i) * 64 + c], acc);
    part[g][c] = acc;
    __syncthreads();
    if (g == 0) {
        const double u = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
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
