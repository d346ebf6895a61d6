// L^-1 by recursive block inversion (gfx950, fp64 MFMA):   inv([[A, 0], [B, C]]) = [[A^-1, 0], [-C^-1 B A^-1, C^-1]].
//
// W = L^-1 is what the variance product, the likelihood gradient, the query gradients and the append step multiply with
// (gp_predict.hip: ensure_winv).  Pushing the identity through the substitution kernel builds it as Npad/16 independent
// chains of up to nb (nb + 1) / 2 dependent stages (0.6 ms at N = 2000); here the 64 x 64 diagonal blocks are inverted first
// (one wavefront each, all in parallel) and every level of the recursion is two batched products of independent 64 x 64
// tiles on the matrix cores -- log2(nb) levels, no chain longer than one block row of tiles.
//   level with block size m (m / 64 = mt tiles), pair p = blocks [2 p mt, 2 p mt + 2 mt):
//     T^T = A^-T B^T    k-tiles j .. mt-1   (A^-1 is lower triangular)   -> scratch in the unused upper block (1, 2) of R
//     X21 = -C^-1 T     k-tiles 0 .. i      (C^-1 is lower triangular)   -> block (2, 1) of R
// nb need not be a power of two: a trailing block C may be shorter than A (or absent); tiles beyond nb are skipped.
// R is row-major [Npad][Npad] (the variance workspace); the result is re-tiled to the tile-major layout of the consumers
// with exact zeros above the diagonal.
#include "gp_device.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// X = L_kk^-1 for every diagonal block: lane i owns row i of X (x_i L = e_i), right-looking from column 63 down:
// x_ij = b_j / L_jj, then b_k -= x_ij L_jk for k < j.  L_jk is uniform (LDS broadcast); two dependent operations per column.
__global__ void __launch_bounds__(64)
inv_diag_kernel(const double* __restrict__ L, int ld, const double* __restrict__ dinv, double* __restrict__ R) {
    __shared__ double Ls[64][66];
    const int lane = threadIdx.x, kb = blockIdx.x;
    const double* Lb = L + (size_t)(kb * 64) * ld + kb * 64;
#pragma unroll
    for (int r = 0; r < 64; ++r) Ls[r][lane] = Lb[(size_t)r * ld + lane];
    const double* di = dinv + kb * 64;
    __syncthreads();
    double b[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) b[j] = (j == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 63; j >= 0; --j) {
        const double x = b[j] * di[j];
        b[j] = x;
#pragma unroll
        for (int k = 0; k < j; ++k) b[k] = fma(-x, Ls[j][k], b[k]);
    }
    double* Rb = R + (size_t)(kb * 64 + lane) * ld + kb * 64;
#pragma unroll
    for (int j = 0; j < 64; j += 2) *reinterpret_cast<f64x2*>(Rb + j) = f64x2{b[j], b[j + 1]};
}

// One 64 x 64 tile of C = alpha * op(A) op(B) for every pair of a level; op = transpose when TA / TB.  Global tiles are always
// read row-wise (coalesced 16-byte loads) and the transposition happens in the LDS read pattern of the MFMA operands.
//   kmode 1: k-tiles ti .. mt-1      kmode 2: k-tiles 0 .. ti
// `bound_tile0` / `pair_tiles` / `bound_cols`: the global tile index of ti = 0 (rows) or tj = 0 (columns) in pair 0 and the
// tile distance between pairs -- tiles whose bounded index reaches nb do not exist (a trailing block may be shorter than
// its partner).  The next k-tile is requested (clamped, so unconditionally) while the current one is multiplied.
template <bool TA, bool TB>
__global__ void __launch_bounds__(256)
inv_gemm_kernel(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, double* __restrict__ C, int ldc,
                size_t pair_stride, int mt, int kmode, double alpha, int bound_tile0, int pair_tiles, int bound_cols, int nb) {
    __shared__ double As[64][66];
    __shared__ double Bs[64][66];
    const int tj = blockIdx.x, ti = blockIdx.y, p = blockIdx.z;
    if (bound_tile0 + p * pair_tiles + (bound_cols ? tj : ti) >= nb) return;
    A += (size_t)p * pair_stride; B += (size_t)p * pair_stride; C += (size_t)p * pair_stride;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int k_lo = (kmode == 1) ? ti : 0, k_hi = (kmode == 1) ? mt : ti + 1;
    v4f64 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = v4f64{0.0, 0.0, 0.0, 0.0};
    f64x2 pa[8], pb[8];
#define ALABI_INV_REQUEST(K)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                           \
        const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;                                                 \
        pa[i] = *reinterpret_cast<const f64x2*>(A + (size_t)((TA ? (K) : ti) * 64 + r) * lda + (TA ? ti : (K)) * 64 + 2 * c2); \
        pb[i] = *reinterpret_cast<const f64x2*>(B + (size_t)((TB ? tj : (K)) * 64 + r) * ldb + (TB ? (K) : tj) * 64 + 2 * c2); \
    }
    ALABI_INV_REQUEST(k_lo)
    for (int k = k_lo; k < k_hi; ++k) {
        __syncthreads();                                   // the previous tile's MFMAs are done with As / Bs
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
            *reinterpret_cast<f64x2*>(&As[r][2 * c2]) = pa[i];
            *reinterpret_cast<f64x2*>(&Bs[r][2 * c2]) = pb[i];
        }
        const int kn = (k + 1 < k_hi) ? k + 1 : k;
        ALABI_INV_REQUEST(kn)
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double a = TA ? As[4 * ks + lk][16 * w + lr] : As[16 * w + lr][4 * ks + lk];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const double b = TB ? Bs[16 * n + lr][4 * ks + lk] : Bs[4 * ks + lk][16 * n + lr];
                acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
            }
        }
    }
#undef ALABI_INV_REQUEST
    double* Ct = C + (size_t)(ti * 64 + 16 * w) * ldc + tj * 64;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) Ct[(size_t)(lk + 4 * i) * ldc + 16 * n + lr] = alpha * acc[n][i];
}

// R (row-major, lower triangle valid) -> dst[t][n][c] = L^-1[n][64 t + c], exact zeros above the diagonal
__global__ void __launch_bounds__(256)
inv_retile_kernel(const double* __restrict__ R, double* __restrict__ dst, int Npad) {
    const size_t n_el = (size_t)Npad * Npad;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n_el; e += (size_t)gridDim.x * 256) {
        const size_t t = e / ((size_t)Npad * 64), rem = e % ((size_t)Npad * 64);
        const size_t n = rem >> 6, col = t * 64 + (rem & 63);
        dst[e] = (col <= n) ? R[n * Npad + col] : 0.0;
    }
}

// dst (tile-major) = L^-1 through the row-major scratch R (Npad^2 doubles, distinct from dst)
int launch_factor_inverse_dnc(alabi_gp* gp, double* R, double* dst, hipStream_t s) {
    const int Npad = gp->Npad, nb = Npad / 64, ld = Npad;
    hipLaunchKernelGGL(inv_diag_kernel, dim3(nb), dim3(64), 0, s, gp->L, ld, gp->dinv, R);
    for (int mt = 1; mt < nb; mt *= 2) {
        const int m = mt * 64;
        const int pairs = (nb + 2 * mt - 1) / (2 * mt);
        const size_t pair_stride = (size_t)2 * m * ld + 2 * m;
        // T^T = A^-T B^T (m x rows-of-C), k-tiles ti .. mt-1:  A^-1 = R[r0.., r0..] read transposed, B = L[(r0 + m).., r0..] read
        // transposed, T^T -> R[r0.., (r0 + m)..] -- the unused upper block, which has exactly this shape also when C is short
        hipLaunchKernelGGL((inv_gemm_kernel<true, true>), dim3(mt, mt, pairs), dim3(256), 0, s, R, ld, gp->L + (size_t)m * ld, ld,
                           R + m, ld, pair_stride, mt, 1, 1.0, mt, 2 * mt, 1, nb);
        // X21 = -C^-1 T, k-tiles 0 .. ti:  C^-1 = R[(r0 + m).., (r0 + m)..], T^T read transposed, X21 -> R[(r0 + m).., r0..]
        hipLaunchKernelGGL((inv_gemm_kernel<false, true>), dim3(mt, mt, pairs), dim3(256), 0, s, R + (size_t)m * ld + m, ld, R + m, ld,
                           R + (size_t)m * ld, ld, pair_stride, mt, 2, -1.0, mt, 2 * mt, 0, nb);
    }
    hipLaunchKernelGGL(inv_retile_kernel, dim3(1024), dim3(256), 0, s, R, dst, Npad);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
