// Analytic gradient of the GP log marginal likelihood for gfx950.
//
// Replaces george's gp.grad_log_likelihood(y) reached from the reference's hyper-parameter fit
// (alabi/core.py:1261 via gp_utils.py:165):  with A = alpha alpha^T - K^-1,
//     d logL / d p = 0.5 tr(A dK/dp),      d logL / d mean = sum(alpha).
// K^-1 is never stored.  launch_factor_inverse leaves W = L^-1 in the workspace (tile-major); one workgroup per
// 64x64 block (ta >= tb) of the lower triangle forms its block of K^-1 = W^T W on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64, k running over the rows n >= 64 ta where both column tiles are non-zero), re-evaluates the
// kernel and its derivatives for the block from the scaled coordinates, and contracts on the spot; off-diagonal blocks
// count twice.  Partial sums per workgroup are reduced in a fixed order (bit-reproducible).
//
//   dK/d log_amp   = K - wn I                     dK/d log_wn = wn I
//   dK/d log_M_k   = amp f'(r2) (-D_k^2),         D_k the scaled coordinate difference
//   dK/d log_alpha = amp a f (u/(1+u) - log1p(u)), u = r2 / (2a)      (rational quadratic only)
#include "gp_device.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// f(r2) and f'(r2) of the four kernels (gp_device.hpp: radial)
__device__ inline void radial_with_derivative(double r2, KernelFn kf, double& f, double& df, double& dlog_alpha) {
    dlog_alpha = 0.0;
    if (kf.type == 0) { f = exp_neg_half(r2); df = -0.5 * f; return; }
    if (kf.type == 1) { const double r = sqrt(3.0 * r2), e = exp(-r); f = (1.0 + r) * e; df = -1.5 * e; return; }
    if (kf.type == 2) {
        const double r = sqrt(5.0 * r2), e = exp(-r);
        f = (1.0 + r + r * r / 3.0) * e; df = -(5.0 / 6.0) * (1.0 + r) * e; return;
    }
    const double u = 0.5 * r2 / kf.alpha, l = log1p(u);
    f = exp(-kf.alpha * l);
    df = -0.5 * f / (1.0 + u);
    dlog_alpha = kf.alpha * f * (u / (1.0 + u) - l);
}

// The 64 x 64 block (ta, tb), tb <= ta, of K^-1 = W^T W from the tile-major W = L^-1: the sum over 64-row slabs n0 >= 64 ta of
// Wa_slab^T Wb_slab.  Both slabs (64 x 64, row n, column a / b) are staged in LDS through registers one slab ahead (16-byte
// coalesced loads), wave w owns output rows 16w..16w+15, four 16-column tiles.  MFMA operand layout (gfx950, 16x16x4 f64):
// A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, C[(lane >> 4) + 4 i][lane & 15] in element i; with the slabs stored
// [k][i] both operands are conflict-free row reads.  256 threads; acc[c][i] = block[16 w + (lane >> 4) + 4 i][16 c + (lane & 15)].
__device__ inline void kinv_block(const double* __restrict__ W, int Npad, int ta, int tb, v4f64 (&acc)[4]) {
    __shared__ double Sa[64][66], Sb[64][66];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const f64x2* Wa2 = reinterpret_cast<const f64x2*>(W + (size_t)ta * Npad * 64);
    const f64x2* Wb2 = reinterpret_cast<const f64x2*>(W + (size_t)tb * Npad * 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = v4f64{0.0, 0.0, 0.0, 0.0};
    f64x2 pa[8], pb[8];
    {
        const size_t base = (size_t)(ta * 64) * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) { pa[i] = Wa2[base + tid + 256 * i]; pb[i] = Wb2[base + tid + 256 * i]; }
    }
    for (int n0 = ta * 64; n0 < Npad; n0 += 64) {
        __syncthreads();                               // the previous slab's MFMAs are done with Sa / Sb
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
            *reinterpret_cast<f64x2*>(&Sa[r][2 * c2]) = pa[i];
            *reinterpret_cast<f64x2*>(&Sb[r][2 * c2]) = pb[i];
        }
        __syncthreads();
        if (n0 + 64 < Npad) {
            const size_t base = (size_t)(n0 + 64) * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) { pa[i] = Wa2[base + tid + 256 * i]; pb[i] = Wb2[base + tid + 256 * i]; }
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double a = Sa[4 * ks + (l >> 4)][16 * w + (l & 15)];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Sb[4 * ks + (l >> 4)][16 * c + (l & 15)], acc[c], 0, 0, 0);
        }
    }
}

// solver.get_inverse() (reference: alabi/utility.py:610, the finite-difference acquisition gradient): K^-1 [N,N] row-major, both
// triangles, one workgroup per block (ta >= tb) of the lower triangle writing the block and its mirror image.
__global__ void __launch_bounds__(256)
kinv_write_kernel(const double* __restrict__ W, int N, int Npad, double* __restrict__ out) {
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int p = blockIdx.x;
    int ta = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while ((ta + 1) * (ta + 2) / 2 <= p) ++ta;
    while (ta * (ta + 1) / 2 > p) --ta;
    const int tb = p - ta * (ta + 1) / 2;
    v4f64 acc[4];
    kinv_block(W, Npad, ta, tb, acc);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ga = ta * 64 + 16 * w + (l >> 4) + 4 * i, gb = tb * 64 + 16 * c + (l & 15);
            if (ga >= N || gb >= N) continue;
            out[(size_t)ga * N + gb] = acc[c][i];
            if (ta != tb) out[(size_t)gb * N + ga] = acc[c][i];
        }
}

// partial[block][0] = sum A (K - wn I), [1] = trace part, [2] = log_alpha part, [3 + k] = log_M_k part
template <int D>
__global__ void __launch_bounds__(256)
grad_contract_kernel(const double* __restrict__ W, const double* __restrict__ Xt, const double* __restrict__ alpha,
                     int N, int Npad, double amp, KernelFn kf, double* __restrict__ partial) {
    __shared__ double xa_s[D][64], xb_s[D][64];
    __shared__ double al_a[64], al_b[64];
    __shared__ double scratch[16];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    // block index -> (ta, tb), tb <= ta
    const int p = blockIdx.x;
    int ta = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while ((ta + 1) * (ta + 2) / 2 <= p) ++ta;
    while (ta * (ta + 1) / 2 > p) --ta;
    const int tb = p - ta * (ta + 1) / 2;
    for (int e = tid; e < D * 64; e += 256) {
        xa_s[e >> 6][e & 63] = Xt[(size_t)(e >> 6) * Npad + ta * 64 + (e & 63)];
        xb_s[e >> 6][e & 63] = Xt[(size_t)(e >> 6) * Npad + tb * 64 + (e & 63)];
    }
    if (tid < 64) { al_a[tid] = alpha[ta * 64 + tid]; al_b[tid] = alpha[tb * 64 + tid]; }
    v4f64 acc[4];
    kinv_block(W, Npad, ta, tb, acc);
    __syncthreads();
    double s_amp = 0.0, s_tr = 0.0, s_al = 0.0, s_m[D];
#pragma unroll
    for (int k = 0; k < D; ++k) s_m[k] = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ra = 16 * w + (l >> 4) + 4 * i, rb = 16 * c + (l & 15);
            const int ga = ta * 64 + ra, gb = tb * 64 + rb;
            if (ga >= N || gb >= N) continue;
            const double A = al_a[ra] * al_b[rb] - acc[c][i];
            double r2 = 0.0, dk2[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double df = xa_s[k][ra] - xb_s[k][rb];
                dk2[k] = df * df;
                r2 += dk2[k];
            }
            double f, fp, fa;
            radial_with_derivative(r2, kf, f, fp, fa);
            s_amp = fma(A, amp * f, s_amp);
            s_al = fma(A, amp * fa, s_al);
            const double g = -A * amp * fp;
#pragma unroll
            for (int k = 0; k < D; ++k) s_m[k] = fma(g, dk2[k], s_m[k]);
            if (ga == gb) s_tr += A;
        }
    const double wgt = (ta == tb) ? 1.0 : 2.0;
    double* out = partial + (size_t)p * (D + 3);
    double v;
    v = block_sum(s_amp, scratch); if (tid == 0) out[0] = wgt * v;
    v = block_sum(s_tr, scratch);  if (tid == 0) out[1] = v;
    v = block_sum(s_al, scratch);  if (tid == 0) out[2] = wgt * v;
#pragma unroll
    for (int k = 0; k < D; ++k) { v = block_sum(s_m[k], scratch); if (tid == 0) out[3 + k] = wgt * v; }
}

// grad = [d/d mean, d/d log_wn, d/d log_amp, d/d log_alpha, d/d log_M_0 ...]; the partial sums are added in block order
__global__ void __launch_bounds__(256)
grad_final_kernel(const double* __restrict__ partial, int nblocks, int stride, const double* __restrict__ alpha, int N,
                  double wn, int d, double* __restrict__ grad) {
    __shared__ double scratch[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s += alpha[i];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) grad[0] = s;
    for (int q = 0; q < 3 + d; ++q) {
        double t = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 256) t += partial[(size_t)b * stride + q];
        t = block_sum(t, scratch);
        const int slot = (q == 0) ? 2 : (q == 1) ? 1 : (q == 2) ? 3 : q + 1;
        if (threadIdx.x == 0) grad[slot] = 0.5 * t * (q == 1 ? wn : 1.0);
    }
}

// W = L^-1 of the current factor: the cache the variance path keeps per factor, or the variance workspace when there is no room
static int factor_inverse_source(alabi_gp* gp, hipStream_t s, const double** Wsrc) {
    int st = ensure_winv(gp, s);
    if (st == ALABI_OK) { *Wsrc = gp->winv; return ALABI_OK; }
    if (st != ALABI_NOT_COMPUTED) return st;
    if ((st = launch_factor_inverse(gp, s)) != ALABI_OK) return st;
    *Wsrc = gp->ws;
    return ALABI_OK;
}

int launch_get_inverse(alabi_gp* gp, double* out, hipStream_t s) {
    const double* Wsrc = nullptr;
    int st = factor_inverse_source(gp, s, &Wsrc);
    if (st != ALABI_OK) return st;
    const int nb = gp->Npad / 64;
    hipLaunchKernelGGL(kinv_write_kernel, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, Wsrc, gp->N, gp->Npad, out);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_grad_log_likelihood(alabi_gp* gp, double* grad_dev, hipStream_t s) {
    const double* Wsrc = nullptr;
    int st = factor_inverse_source(gp, s, &Wsrc);
    if (st != ALABI_OK) return st;
    const int nb = gp->Npad / 64, nblocks = nb * (nb + 1) / 2;
    const int db = dim_bucket(gp->d);
    const size_t need = (size_t)nblocks * (db + 3) * sizeof(double);
    if (need > gp->scan_bytes) {
        if (gp->scan) {
            ALABI_HIP_CHECK(hipStreamSynchronize(s));
            ALABI_HIP_CHECK(hipFree(gp->scan));
            gp->scan = nullptr; gp->scan_bytes = 0;
        }
        ALABI_HIP_CHECK(hipMalloc(&gp->scan, need));
        gp->scan_bytes = need;
    }
    const double amp = exp(gp->log_amp), wn = exp(gp->log_wn);
    ALABI_DISPATCH_DIM(db, hipLaunchKernelGGL(grad_contract_kernel<D>, dim3(nblocks), dim3(256), 0, s, Wsrc, gp->Xt, gp->alpha,
                                              gp->N, gp->Npad, amp, gp->kf, gp->scan));
    hipLaunchKernelGGL(grad_final_kernel, dim3(1), dim3(256), 0, s, gp->scan, nblocks, db + 3, gp->alpha, gp->N, wn, gp->d,
                       grad_dev);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
