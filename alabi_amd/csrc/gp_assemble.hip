// Squared-exponential (ARD) kernel-matrix assembly for gfx950.
//
// Replaces george's kernel.get_value(x) inside gp.compute (reference call site
// alabi/core.py:1158; kernel built at alabi/core.py:1000 and gp_utils.py:230-231):
//     K[i,j] = exp(log_amp) * exp(-0.5 * sum_k (x_ik - x_jk)^2 / exp(log_M_k)) + wn * delta_ij
//
// Layout: the training inputs are first re-laid out as Xt[k][n] = X[n][k] * exp(-0.5 log_M_k)
// (SoA, coalesced along n), so every later kernel computes r^2 = sum_k (a_k - b_k)^2 directly.
// K is written as the LOWER triangle of an [Npad, Npad] row-major matrix (Npad = N rounded
// up to 64); the padding is the identity so blocked kernels need no edge cases.
// HBM-write bound: 8 * Npad^2 / 2 bytes per assembly.
#include "gp_device.hpp"

namespace alabi {

__global__ void __launch_bounds__(256)
prepare_inputs_kernel(const double* __restrict__ X, int N, int Npad, int d, int dbucket,
                      DimVec inv_len, double* __restrict__ Xt) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    for (int k = 0; k < dbucket; ++k) {  // rows d..dbucket-1 and columns N..Npad-1 are zero
        double v = 0.0;
        if (n < N && k < d) v = X[(size_t)n * d + k] * inv_len.v[k];
        Xt[(size_t)k * Npad + n] = v;
    }
}

// One 64x64 tile of the lower triangle per workgroup.  Every thread owns a 4 x 4 micro-tile (rows 4 (tid >> 4) + a, columns
// (tid & 15) + 16 b): per coordinate it reads four scaled row and four scaled column values from LDS for sixteen
// difference-square-accumulate pairs -- 0.5 LDS reads per pair instead of 2 with one element per thread and step (round 3: the
// kernel was bound by exactly those reads, 0.39 ms at N = 10000, d = 20, four times the time its 400 MB of writes take).  Sixteen
// lanes store 128 contiguous bytes of a row.  Same operations per element in the same order: the same bits as before.
// (assemble_tile in gp_device.hpp: shared with the batched assembly of gp_batch.hip, which must produce the same bits)
__global__ void __launch_bounds__(256)
assemble_lower_kernel(const double* __restrict__ Xt, int N, int Npad, int d, double amp, double wn, KernelFn kf,
                      double* __restrict__ K, int* __restrict__ zero, int zero_ints, int* __restrict__ info) {
    __shared__ double xi[ALABI_MAX_DIM][64];
    __shared__ double xj[ALABI_MAX_DIM][64];
    // the factorisation's control words (task-queue head, time-out flag, tile versions) and its status word start at zero:
    // cleared here instead of by two memset nodes in front of the factorisation kernel
    for (int i = blockIdx.x * 256 + threadIdx.x; i < zero_ints; i += gridDim.x * 256) zero[i] = 0;
    // ... and the slab buffers behind them (gp_cholesky.hip: [nb][4][64][16] doubles) start as the tag the panel solves poll them for
    if (zero_ints > 0) {
        const int nbt = Npad / 64, base = (zero_ints + 1) & ~1;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nbt * 8192; i += gridDim.x * 256) zero[base + i] = (int)0x7FF8DEADu;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *info = 0;
    // linear tile id -> (bi >= bj)
    int t = blockIdx.x;
    int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    int bj = t - bi * (bi + 1) / 2;
    assemble_tile(xi, xj, Xt, N, Npad, d, amp, wn, kf, K, bi, bj);
}

// Rectangular kernel.get_value(x1, x2): both inputs raw [n,d] row-major.
__global__ void __launch_bounds__(256)
kernel_matrix_kernel(const double* __restrict__ X1, int n1, const double* __restrict__ X2, int n2,
                     int d, double amp, DimVec inv_len, KernelFn kf, double* __restrict__ K) {
    __shared__ double xi[ALABI_MAX_DIM][64];
    __shared__ double xj[ALABI_MAX_DIM][64];
    int bi = blockIdx.y, bj = blockIdx.x;
    int tid = threadIdx.x;
    for (int e = tid; e < d * 64; e += 256) {
        int c = e / d, k = e % d;  // contiguous reads of row-major inputs
        int gi = bi * 64 + c, gj = bj * 64 + c;
        xi[k][c] = gi < n1 ? X1[(size_t)gi * d + k] * inv_len.v[k] : 0.0;
        xj[k][c] = gj < n2 ? X2[(size_t)gj * d + k] * inv_len.v[k] : 0.0;
    }
    __syncthreads();
    int c = tid & 63;
    int gc = bj * 64 + c;
    if (gc >= n2) return;
    for (int r = tid >> 6; r < 64; r += 4) {
        int gr = bi * 64 + r;
        if (gr >= n1) break;
        double r2 = 0.0;
        for (int k = 0; k < d; ++k) {
            double df = xi[k][r] - xj[k][c];
            r2 = fma(df, df, r2);
        }
        K[(size_t)gr * n2 + gc] = amp * radial(r2, kf);
    }
}

int launch_prepare_inputs(alabi_gp* gp, const double* X, int N, hipStream_t s) {
    int Npad = gp->Npad;
    hipLaunchKernelGGL(prepare_inputs_kernel, dim3((Npad + 255) / 256), dim3(256), 0, s, X, N, Npad,
                       gp->d, dim_bucket(gp->d), gp->inv_len, gp->Xt);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_assemble(alabi_gp* gp, hipStream_t s, int zero_ctl_ints) {
    int nb = gp->Npad / 64;
    int tiles = nb * (nb + 1) / 2;
    hipLaunchKernelGGL(assemble_lower_kernel, dim3(tiles), dim3(256), 0, s, gp->Xt, gp->N, gp->Npad,
                       gp->d, exp(gp->log_amp), exp(gp->log_wn), gp->kf, gp->L, gp->chol_ctl, zero_ctl_ints, gp->info);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_kernel_matrix(const double* X1, int n1, const double* X2, int n2, int d, double amp,
                         const DimVec& inv_len, KernelFn kf, double* K, hipStream_t s) {
    dim3 grid((n2 + 63) / 64, (n1 + 63) / 64);
    hipLaunchKernelGGL(kernel_matrix_kernel, grid, dim3(256), 0, s, X1, n1, X2, n2, d, amp, inv_len, kf, K);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
