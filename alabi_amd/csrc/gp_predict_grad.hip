// Predictive mean / variance AND their gradients with respect to the query point, for small batches (gfx950, fp64).
//
// Replaces the reference's finite-difference kernel gradients and explicit K^-1 behind the acquisition optimiser:
//   alabi/utility.py:511-555  numerical_kernel_gradient  (2 d extra kernel rows per call, step 1e-6)
//   alabi/utility.py:558-583  grad_gp_mean_prediction    d mu / d x  =  (d k*/d x)^T alpha
//   alabi/utility.py:586-623  grad_gp_var_prediction     d var / d x = -2 (d k*/d x)^T K^-1 k*   (get_inverse(): O(N^3) per call)
// Here d k*/d x is closed form and K^-1 k* = W^T (W k*) with the cached W = L^-1 (gp_predict.hip: ensure_winv), so one
// gradient costs two passes over the lower triangle of W (N^2 fp64 MFMA flops each) for up to 16 query points at once.
//
// Three launches per group of <= 16 queries:
//   pgrad_v_kernel      v = W k*            one workgroup per block row (K* blocks evaluated in place), also |v|^2 partials
//   pgrad_z_kernel      z = W^T v           one workgroup per (column tile, row part); A and B operands straight from HBM/L2
//   pgrad_final_kernel  mu, var, d mu, d var per query: one pass over the training points with closed-form d k / d x
// All sums run in a fixed order (no atomics), so results are reproducible run to run.
#include "gp_device.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// f(r2) and f'(r2) of the four radial families (gp_device.hpp: radial)
template <bool GENERIC>
__device__ inline void radial_and_slope(double r2, KernelFn kf, double& f, double& df) {
    if (!GENERIC || kf.type == 0) { f = exp_neg_half(r2); df = -0.5 * f; return; }
    if (kf.type == 1) { const double r = sqrt(3.0 * r2), e = exp(-r); f = (1.0 + r) * e; df = -1.5 * e; return; }
    if (kf.type == 2) {
        const double r = sqrt(5.0 * r2), e = exp(-r);
        f = (1.0 + r + r * r / 3.0) * e; df = -(5.0 / 6.0) * (1.0 + r) * e; return;
    }
    const double u = 0.5 * r2 / kf.alpha;
    f = exp(-kf.alpha * log1p(u));
    df = -0.5 * f / (1.0 + u);
}

// v[(64 kb + r) * 16 + q] = sum_{j <= kb} W[kb, j] K*_j ; partial[kb * 16 + q] = sum over the block's rows of v^2.
// MFMA operand layout: A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15], D row (lane >> 4) + 4 i, column lane & 15.
template <int D, bool GENERIC>
__global__ void __launch_bounds__(256)
pgrad_v_kernel(const double* __restrict__ W, const double* __restrict__ Xt, int N, int Npad, const double* __restrict__ Xs,
               int d, int M, DimVec inv_len, double amp, KernelFn kf, double* __restrict__ v, double* __restrict__ partial) {
    __shared__ double Wt[64][66];
    __shared__ double Ks[64][18];
    __shared__ double qs[16][D];
    __shared__ double red[4][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int kb = blockIdx.x;
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
#pragma unroll
    for (int i = 0; i < 4; ++i) v[(size_t)(kb * 64 + 16 * w + lk + 4 * i) * 16 + lr] = acc[i];
    double ss = fma(acc[0], acc[0], fma(acc[1], acc[1], fma(acc[2], acc[2], acc[3] * acc[3])));
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) red[w][lane] = ss;
    __syncthreads();
    if (tid < 16) partial[(size_t)kb * 16 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// zpart[p][(64 t + c) * 16 + q] = sum over the rows i of part p (i >= 64 t) of W[i, 64 t + c] v[i, q].
// Tile-major W: column tile t is a contiguous [Npad][64] slab, so lane (m, k) reads W_t[i + k][16 w + m]: four 128-byte
// row segments per wave and step; v[i + k][q] is one contiguous 512-byte line.  Two accumulators hide the MFMA latency.
__global__ void __launch_bounds__(256)
pgrad_z_kernel(const double* __restrict__ W, const double* __restrict__ v, int Npad, int parts, double* __restrict__ zpart) {
    const int t = blockIdx.x / parts, p = blockIdx.x % parts;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int nrows = Npad - 64 * t;
    const int per = ((nrows + parts - 1) / parts + 7) & ~7;          // rows per part, a multiple of 8
    int r0 = 64 * t + p * per, r1 = r0 + per;
    if (r1 > Npad) r1 = Npad;
    const double* Wt = W + (size_t)t * Npad * 64 + 16 * w + lr;
    v4f64 a0 = v4f64{0.0, 0.0, 0.0, 0.0}, a1 = a0;
    for (int i = r0; i < r1; i += 8) {                               // r1 - r0 is a multiple of 8 (nrows is a multiple of 64)
        const double wa = Wt[(size_t)(i + lk) * 64], wb = Wt[(size_t)(i + 4 + lk) * 64];
        const double va = v[(size_t)(i + lk) * 16 + lr], vb = v[(size_t)(i + 4 + lk) * 16 + lr];
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, va, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wb, vb, a1, 0, 0, 0);
    }
    double* out = zpart + (size_t)p * Npad * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(size_t)(64 * t + 16 * w + lk + 4 * i) * 16 + lr] = a0[i] + a1[i];
}

// One workgroup per query q: var = amp - sum_kb partial, mu = mean + sum_i alpha_i k_i,
// d mu / d x_c = sum_i alpha_i dk_i/dx_c,  d var / d x_c = -2 sum_i z_i dk_i/dx_c,  with
// dk_i/dx_c = amp f'(r2_i) 2 (x_c - X_ic) / M_c  (coordinates in Xt / qs are already divided by sqrt(M_c)).
template <int D, bool GENERIC>
__global__ void __launch_bounds__(256)
pgrad_final_kernel(const double* __restrict__ Xt, const double* __restrict__ alpha, int N, int Npad,
                   const double* __restrict__ Xs, int d, DimVec inv_len, double amp, double mean, KernelFn kf,
                   const double* __restrict__ zpart, int parts, const double* __restrict__ partial, int nb,
                   double* __restrict__ mu, double* __restrict__ var, double* __restrict__ dmu, double* __restrict__ dvar) {
    __shared__ double qs[D];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid < D) qs[tid] = (tid < d) ? Xs[(size_t)q * d + tid] * inv_len.v[tid] : 0.0;
    __syncthreads();
    double sm[D], sv[D], smu = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) { sm[k] = 0.0; sv[k] = 0.0; }
#pragma unroll 2
    for (int i = tid; i < N; i += 256) {
        double df[D], r2 = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { df[k] = qs[k] - Xt[(size_t)k * Npad + i]; r2 = fma(df[k], df[k], r2); }
        double f, fp;
        radial_and_slope<GENERIC>(r2, kf, f, fp);
        double z = 0.0;
        for (int p = 0; p < parts; ++p) z += zpart[((size_t)p * Npad + i) * 16 + q];
        const double a = alpha[i];
        smu = fma(a, amp * f, smu);
        const double g = 2.0 * amp * fp;
        const double ga = a * g, gz = z * g;
#pragma unroll
        for (int k = 0; k < D; ++k) { sm[k] = fma(ga, df[k], sm[k]); sv[k] = fma(gz, df[k], sv[k]); }
    }
    // 2 D + 1 sums over the workgroup: one DPP reduction per value inside each wave (result in lane 63), one LDS exchange,
    // then thread k adds the four wave totals of value k in a fixed order -- one barrier instead of two per value.
    __shared__ double red[4][2 * D + 1];
    const int lane = tid & 63, wv = tid >> 6;
    {
        const double t = wave_sum_dpp(smu);
        if (lane == 63) red[wv][2 * D] = t;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double tm = wave_sum_dpp(sm[k]);
        const double tv = wave_sum_dpp(sv[k]);
        if (lane == 63) { red[wv][k] = tm; red[wv][D + k] = tv; }
    }
    __syncthreads();
    if (tid < 2 * D + 1) {
        const double t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        if (tid == 2 * D) { if (mu) mu[q] = mean + t; }
        else if (tid < D) { if (tid < d) dmu[(size_t)q * d + tid] = t * inv_len.v[tid]; }
        else if (tid - D < d) dvar[(size_t)q * d + (tid - D)] = -2.0 * t * inv_len.v[tid - D];
    }
    if (var && tid == 64) {
        double sacc = 0.0;
        for (int kb = 0; kb < nb; ++kb) sacc += partial[(size_t)kb * 16 + q];
        var[q] = amp - sacc;
    }
}

static int ensure_pgrad(alabi_gp* gp, size_t bytes, hipStream_t s) {
    if (gp->pgrad_bytes >= bytes) return ALABI_OK;
    if (gp->pgrad) {
        ALABI_HIP_CHECK(hipStreamSynchronize(s));
        ALABI_HIP_CHECK(hipFree(gp->pgrad));
        gp->pgrad = nullptr; gp->pgrad_bytes = 0;
    }
    ALABI_HIP_CHECK(hipMalloc(&gp->pgrad, bytes));
    gp->pgrad_bytes = bytes;
    return ALABI_OK;
}

int launch_predict_grad(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var, double* dmu, double* dvar,
                        hipStream_t s) {
    const int Npad = gp->Npad, nb = Npad / 64, db = dim_bucket(gp->d), d = gp->d;
    if (db < 0) return ALABI_BAD_ARGUMENT;
    int st = ensure_winv(gp, s);
    if (st != ALABI_OK) return st == ALABI_NOT_COMPUTED ? ALABI_HIP_ERROR : st;   // no room for L^-1: nothing to fall back on
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n_cu <= 0) n_cu = 256;
    int parts = 1;
    while (parts < 8 && nb * parts < 2 * n_cu) parts *= 2;
    if (parts > nb) parts = nb > 0 ? nb : 1;
    // scratch: v [Npad][16] | partial [nb][16] | zpart [parts][Npad][16]
    const size_t nv = (size_t)Npad * 16, np = (size_t)nb * 16, nz = (size_t)parts * Npad * 16;
    if ((st = ensure_pgrad(gp, (nv + np + nz) * sizeof(double), s)) != ALABI_OK) return st;
    double* v = gp->pgrad; double* partial = v + nv; double* zpart = partial + np;
    const double amp = exp(gp->log_amp);
    for (long long m0 = 0; m0 < M; m0 += 16) {
        const int mc = (int)((M - m0 < 16) ? (M - m0) : 16);
        const double* xs = Xs + (size_t)m0 * d;
        ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(gp->kf.type, {
            hipLaunchKernelGGL((pgrad_v_kernel<D, GENERIC>), dim3(nb), dim3(256), 0, s, gp->winv, gp->Xt, gp->N, Npad, xs, d, mc,
                               gp->inv_len, amp, gp->kf, v, partial);
            hipLaunchKernelGGL(pgrad_z_kernel, dim3(nb * parts), dim3(256), 0, s, gp->winv, v, Npad, parts, zpart);
            hipLaunchKernelGGL((pgrad_final_kernel<D, GENERIC>), dim3(mc), dim3(256), 0, s, gp->Xt, gp->alpha, gp->N, Npad, xs, d,
                               gp->inv_len, amp, gp->mean, gp->kf, zpart, parts, partial, nb, mu ? mu + m0 : nullptr,
                               var ? var + m0 : nullptr, dmu + (size_t)m0 * d, dvar + (size_t)m0 * d);
        }));
    }
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
