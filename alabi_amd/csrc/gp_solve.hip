// alpha = K^-1 (y - mean) by blocked forward / backward substitution with the Cholesky
// factor, plus the two scalar reductions (log-determinant and r^T alpha).
//
// Replaces george's GP._compute_alpha (scipy cho_solve) reached from the reference through
// the first gp.predict after each gp.compute (alabi/core.py:85, :1441, :1486), and
// solver.log_determinant / gp.log_likelihood (alabi/core.py:1248, gp_utils.py:139).
//
// One launch per 64-row block step.  Every workgroup of step kb re-solves the 64x64
// diagonal system in one wavefront (lane i holds row i, the solved entry is broadcast with
// a wave shuffle) and then applies its own 64x64 block of the update, so a step is ONE
// kernel and the steps are ordered by the stream.  2 N^2 flops, latency bound: once per refit.
#include "gp_device.hpp"

namespace alabi {

// Forward step kb: z_kb = L_kk^-1 r_kb ; r_i -= L[i,kb] z_kb for i > kb.
// blockIdx.x = i - kb.  r is updated in place below block kb; z is written to `z`.
__global__ void __launch_bounds__(256)
trsv_fwd_step_kernel(const double* __restrict__ L, int ld, int kb, const double* __restrict__ dinv,
                     double* __restrict__ r, double* __restrict__ z) {
    __shared__ double lkk[64][65];
    __shared__ double zs[64];
    const int tid = threadIdx.x;
    const double* Lb = L + (size_t)(kb * 64) * ld + kb * 64;
    for (int e = tid; e < 4096; e += 256) {
        int rr = e >> 6, c = e & 63;
        lkk[rr][c] = (c <= rr) ? Lb[(size_t)rr * ld + c] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        double v = r[kb * 64 + tid];
        const double di = dinv[kb * 64 + tid];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const double zj = lane_bcast(v * di, j);   // v_readlane with a static lane: no LDS round trip
            if (tid == j) v = zj;
            if (tid > j) v = fma(-lkk[tid][j], zj, v);
        }
        zs[tid] = v;
        if (blockIdx.x == 0) z[kb * 64 + tid] = v;
    }
    __syncthreads();
    if (blockIdx.x == 0) return;
    const int i = kb + blockIdx.x;
    // r_i[row] -= sum_c L[i*64+row][kb*64+c] * zs[c]; thread = (row, quarter of c)
    const int row = tid >> 2, q = tid & 3;
    const double* Lrow = L + (size_t)(i * 64 + row) * ld + kb * 64 + q * 16;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s = fma(Lrow[c], zs[q * 16 + c], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (q == 0) r[i * 64 + row] -= s;
}

// Backward step kb: a_kb = L_kk^-T z_kb ; z_i -= L[kb,i]^T a_kb for i < kb.  blockIdx.x = i
// for i < kb, and blockIdx.x == kb is the block that publishes a_kb.
__global__ void __launch_bounds__(256)
trsv_bwd_step_kernel(const double* __restrict__ L, int ld, int kb, const double* __restrict__ dinv,
                     double* __restrict__ z, double* __restrict__ alpha) {
    __shared__ double lkk[64][65];
    __shared__ double as[64];
    __shared__ double part[4][64];
    const int tid = threadIdx.x;
    const double* Lb = L + (size_t)(kb * 64) * ld + kb * 64;
    for (int e = tid; e < 4096; e += 256) {
        int rr = e >> 6, c = e & 63;
        lkk[rr][c] = (c <= rr) ? Lb[(size_t)rr * ld + c] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        double v = z[kb * 64 + tid];
        const double di = dinv[kb * 64 + tid];
#pragma unroll
        for (int j = 63; j >= 0; --j) {
            const double aj = lane_bcast(v * di, j);
            if (tid == j) v = aj;
            if (tid < j) v = fma(-lkk[j][tid], aj, v);  // (L^T)[tid][j] = L[j][tid]
        }
        as[tid] = v;
        if ((int)blockIdx.x == kb) alpha[kb * 64 + tid] = v;
    }
    __syncthreads();
    if ((int)blockIdx.x == kb) return;
    const int i = blockIdx.x;
    // z_i[r] -= sum_c L[kb*64+c][i*64+r] * as[c]; lanes run along r (coalesced rows of L)
    const int rr = tid & 63, q = tid >> 6;
    const double* Lc = L + (size_t)(kb * 64 + q * 16) * ld + i * 64 + rr;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s = fma(Lc[(size_t)c * ld], as[q * 16 + c], s);
    part[q][rr] = s;
    __syncthreads();
    if (tid < 64) z[i * 64 + tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

__global__ void __launch_bounds__(256)
residual_kernel(const double* __restrict__ y, int N, int Npad, double mean, double* __restrict__ r) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < Npad) r[n] = (n < N) ? y[n] - mean : 0.0;
}

// red[0] = 2 sum log L_ii ; red[1] = sum (y_i - mean) alpha_i   (single workgroup, fixed order)
__global__ void __launch_bounds__(256)
gp_reduce_kernel(const double* __restrict__ L, int ld, int N, const double* __restrict__ y,
                 double mean, const double* __restrict__ alpha, int have_alpha,
                 double* __restrict__ red) {
    __shared__ double s0[256], s1[256];
    int tid = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = tid; i < N; i += 256) {
        a += log(L[(size_t)i * ld + i]);
        if (have_alpha) b = fma(y[i] - mean, alpha[i], b);
    }
    s0[tid] = a; s1[tid] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) { s0[tid] += s0[tid + w]; s1[tid] += s1[tid + w]; }
        __syncthreads();
    }
    if (tid == 0) { red[0] = 2.0 * s0[0]; red[1] = s1[0]; }
}

int launch_alpha(alabi_gp* gp, hipStream_t s) {
    const int ld = gp->Npad, nb = gp->Npad / 64;
    double* r = gp->work;            // [Npad] residual, consumed by the forward sweep
    double* z = gp->work + gp->n_cap;  // [Npad] forward solution, consumed by the backward sweep
    hipLaunchKernelGGL(residual_kernel, dim3((gp->Npad + 255) / 256), dim3(256), 0, s, gp->y, gp->N,
                       gp->Npad, gp->mean, r);
    for (int kb = 0; kb < nb; ++kb)
        hipLaunchKernelGGL(trsv_fwd_step_kernel, dim3(nb - kb), dim3(256), 0, s, gp->L, ld, kb, gp->dinv, r, z);
    for (int kb = nb - 1; kb >= 0; --kb)
        hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3(kb + 1), dim3(256), 0, s, gp->L, ld, kb, gp->dinv, z, gp->alpha);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_reductions(alabi_gp* gp, hipStream_t s) {
    hipLaunchKernelGGL(gp_reduce_kernel, dim3(1), dim3(256), 0, s, gp->L, gp->Npad, gp->N, gp->y,
                       gp->mean, gp->alpha, gp->has_alpha ? 1 : 0, gp->red);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
