// Batched GP prediction for gfx950: mu* = k(X*,X) alpha + m and var* = k** - |L^-1 k*|^2.
//
// Replaces george's gp.predict(y, X*, return_var=) reached from the reference at
// alabi/core.py:85, :95 (cached likelihood), :1441, :1486 (surrogate_log_likelihood),
// :1601 (acquisition), :1812/:1828 (bookkeeping).  K* is never materialised in HBM.
//
// mean, small M  : one workgroup per query, lanes along the training points (the same device
//                  function the ensemble sampler uses for a walker's log-probability).
// mean, large M  : 64 queries per workgroup, lanes along queries (no cross-lane reduction);
//                  256 training points at a time staged in LDS and broadcast to the lanes.
//                  Bound by the fp64 vector/transcendental rate: N (2d+2) flops + N exp per
//                  query, ~8(d+1) bytes of HBM per query.
// variance       : per tile of 64 queries a blocked forward substitution V = L^-1 K*^T.  For
//                  each 64-row block kb: the K* block is evaluated on the fly, the
//                  off-diagonal part  C = K*_kb - sum_{j<kb} L[kb,j] V_j  runs on the fp64
//                  matrix cores (v_mfma_f64_16x16x4_f64, L block and V block staged in LDS),
//                  the 64x64 diagonal solve runs as 16 columns per wave with 4 lanes per
//                  column exchanging the solved entry by wave shuffle.  V_j tiles live in a
//                  per-workgroup HBM workspace (they do not fit in 160 KB of LDS) and are
//                  re-read through L2.  N^2 flops per query (MFMA bound), L streamed once per
//                  64-query tile.
#include <cstdlib>
#include "gp_device.hpp"

namespace alabi {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int D>
__global__ void __launch_bounds__(256)
predict_mean_rowwise_kernel(const double* __restrict__ Xt, const double* __restrict__ alpha, int Npad,
                            const double* __restrict__ Xs, int d, DimVec inv_len, double amp, double mean,
                            KernelFn kf, double* __restrict__ mu) {
    __shared__ double q[ALABI_MAX_DIM];
    __shared__ double scratch[16];
    const long long m = blockIdx.x;
    if (threadIdx.x < D) q[threadIdx.x] = (threadIdx.x < d) ? Xs[m * d + threadIdx.x] * inv_len.v[threadIdx.x] : 0.0;
    __syncthreads();
    double s = gp_kernel_dot_block<D>(Xt, alpha, Npad, q, scratch, kf);
    if (threadIdx.x == 0) mu[m] = fma(amp, s, mean);
}

template <int D, bool GENERIC>
__global__ void __launch_bounds__(256)
predict_mean_tile_kernel(const double* __restrict__ Xt, const double* __restrict__ alpha, int Npad,
                         const double* __restrict__ Xs, int d, long long M, DimVec inv_len, double amp,
                         double mean, KernelFn kf, double* __restrict__ mu) {
    __shared__ double xt[D][256];
    __shared__ double al[256];
    __shared__ double part[4][64];
    const int tid = threadIdx.x, c = tid & 63, w = tid >> 6;
    const long long m = (long long)blockIdx.x * 64 + c;
    double q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) q[k] = (m < M && k < d) ? Xs[m * d + k] * inv_len.v[k] : 0.0;
    double acc = 0.0;
    for (int n0 = 0; n0 < Npad; n0 += 256) {
        __syncthreads();
        const int n = n0 + tid;
#pragma unroll
        for (int k = 0; k < D; ++k) xt[k][tid] = (n < Npad) ? Xt[(size_t)k * Npad + n] : 0.0;
        al[tid] = (n < Npad) ? alpha[n] : 0.0;
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < 64; ++j) {
            const int nn = w * 64 + j;
            double r2 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double df = xt[k][nn] - q[k];
                r2 = fma(df, df, r2);
            }
            acc = fma(al[nn], radial<GENERIC>(r2, kf), acc);
        }
    }
    part[w][c] = acc;
    __syncthreads();
    if (tid < 64 && m < M) mu[m] = fma(amp, (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]), mean);
}

// IDENT = true turns the same blocked forward substitution into a triangular inversion: the right-hand side of
// tile t is columns 64t..64t+63 of the identity, the V tiles are the OUTPUT (ws[t] = L^-1[:, 64t:64t+64] as
// [Npad][64], rows above block t are never written nor read), and nothing else is produced.
template <int D, bool IDENT = false>
__global__ void __launch_bounds__(256, 2)
predict_var_kernel(const double* __restrict__ L, const double* __restrict__ dinv, const double* __restrict__ Xt,
                   const double* __restrict__ alpha, int N, int Npad, const double* __restrict__ Xs, int d, long long M, DimVec inv_len,
                   double amp, double mean, double* __restrict__ ws, double* __restrict__ mu,
                   double* __restrict__ var, int split, KernelFn kf) {
    // `split` is always 0.  The `split != k` tests below are opaque to the compiler and put the GEMM step and the
    // unrolled diagonal solve into basic blocks of their own: as ONE block the register allocator hoists the
    // solve's LDS reads into the GEMM phase and spills 100 VGPRs to scratch inside the hot loop
    // (measured on MI355X: 11.3 ms -> 6.9 ms per 65536 queries at N=2000).
    __shared__ double As[64][66];   // L[kb,j] block, then L[kb,kb]
    __shared__ double Vs[64][80];   // V_j tile (MFMA B operand), then the C tile
    __shared__ double xtr[D][64];   // scaled coordinates of training block kb
    __shared__ double alb[64];
    __shared__ double dis[64];      // 1 / L_rr of block kb
    const int tid = threadIdx.x, c = tid & 63, w = tid >> 6;
    const int lr = c & 15, lk = c >> 4;  // MFMA lane decomposition within the wave
    const int nb = Npad / 64, ld = Npad;
    double* V = ws + (size_t)blockIdx.x * Npad * 64;
    const long long ntiles = (M + 63) / 64;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m = tile * 64 + c;
        if (IDENT) V = ws + (size_t)tile * Npad * 64;
        const int kb0 = IDENT ? (int)tile : 0;       // first block row with a non-zero right-hand side
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = (!IDENT && m < M && k < d) ? Xs[m * d + k] * inv_len.v[k] : 0.0;
        double mu_acc = 0.0, ss = 0.0;
        for (int kb = kb0; kb < nb; ++kb) {
            __syncthreads();   // the previous block's solve has finished with As / Vs / xtr / alb / dis
            // Software pipeline: the 64x64 L block and V tile of step j+1 travel HBM/L2 -> registers (16 B per
            // lane, 8 + 8 loads) while step j runs on the matrix cores; the first stage is issued here so
            // that the K* evaluation below hides it.
            f64x2 pa[8], pv[8];
            {
                const double* Lb = L + (size_t)(kb * 64) * ld + (kb > kb0 ? kb0 * 64 : kb * 64);
                const double* V0 = V + (size_t)(kb0 * 64) * 64;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
                    pa[i] = *reinterpret_cast<const f64x2*>(Lb + (size_t)r * ld + 2 * c2);
                    if (kb > kb0 && split != 1) pv[i] = reinterpret_cast<const f64x2*>(V0)[e];
                }
            }
            for (int e = tid; e < D * 64; e += 256) xtr[e >> 6][e & 63] = Xt[(size_t)(e >> 6) * Npad + kb * 64 + (e & 63)];
            if (tid < 64) { alb[tid] = alpha[kb * 64 + tid]; dis[tid] = dinv[kb * 64 + tid]; }
            __syncthreads();
            // K* block: thread (column c, wave w) evaluates rows 16w .. 16w+15
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int r = w * 16 + i;
                double r2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double df = xtr[k][r] - q[k];
                    r2 = fma(df, df, r2);
                }
                double kv = (kb * 64 + r < N) ? amp * radial(r2, kf) : 0.0;
                if (IDENT) kv = (kb == kb0 && r == c) ? 1.0 : 0.0;
                mu_acc = fma(kv, alb[r], mu_acc);
                Vs[r][c] = kv;
            }
            __syncthreads();
            v4f64 acc[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[n][i] = Vs[16 * w + lk + 4 * i][16 * n + lr];
            for (int j = kb0; j < kb; ++j) {
                __syncthreads();       // every wave is done reading As / Vs
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
                    *reinterpret_cast<f64x2*>(&As[r][2 * c2]) = pa[i];
                    *reinterpret_cast<f64x2*>(&Vs[r][2 * c2]) = pv[i];
                }
                __syncthreads();
                {   // next stage: L[kb, j+1] and V_{j+1}, or the diagonal block L[kb, kb] after the last step
                    const int jn = j + 1;
                    const double* Lb = L + (size_t)(kb * 64) * ld + jn * 64;
                    const double* Vj = V + (size_t)(jn * 64) * 64;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
                        if (split != 2) pa[i] = *reinterpret_cast<const f64x2*>(Lb + (size_t)r * ld + 2 * c2);
                        if (jn < kb && split != 1) pv[i] = reinterpret_cast<const f64x2*>(Vj)[e];
                    }
                }
                if (split != 3)
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    double a = -As[16 * w + lr][4 * ks + lk];
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        double b = Vs[4 * ks + lk][16 * n + lr];
                        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) Vs[16 * w + lk + 4 * i][16 * n + lr] = acc[n][i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // pa holds L[kb,kb]; only its lower triangle is read below
                const int e = tid + 256 * i, r = e >> 5, c2 = e & 31;
                *reinterpret_cast<f64x2*>(&As[r][2 * c2]) = pa[i];
            }
            __syncthreads();
            // diagonal solve: wave w owns columns 16w..16w+15; lane (col lr, group lk) holds rows == lk (mod 4)
            const int col = 16 * w + lr;
            double v[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = Vs[4 * t + lk][col];
            if (split != 4)
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                const int owner = r & 3, t = r >> 2;
                double x = v[t] * dis[r];
                x = __shfl(x, lr + 16 * owner, 64);
                if (lk == owner) v[t] = x;
#pragma unroll
                for (int t2 = 0; t2 < 16; ++t2) {
                    if (4 * t2 + 3 > r) {
                        const int r2 = 4 * t2 + lk;
                        if (r2 > r) v[t2] = fma(-As[r2][r], x, v[t2]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                V[(size_t)(kb * 64 + 4 * t + lk) * 64 + col] = v[t];
                ss = fma(v[t], v[t], ss);
            }
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (IDENT) continue;
        {
            const long long mc = tile * 64 + 16 * w + lr;
            if (lk == 0 && mc < M) var[mc] = amp - ss;
        }
        __syncthreads();
        Vs[w][c] = mu_acc;
        __syncthreads();
        if (tid < 64 && m < M) mu[m] = ((Vs[0][c] + Vs[1][c]) + (Vs[2][c] + Vs[3][c])) + mean;
    }
}

int launch_predict_mean(alabi_gp* gp, const double* Xs, long long M, double* mu, hipStream_t s) {
    if (M <= 0) return ALABI_OK;
    const int db = dim_bucket(gp->d);
    const double amp = exp(gp->log_amp);
    if (M <= 4096) {
        ALABI_DISPATCH_DIM(db, hipLaunchKernelGGL(predict_mean_rowwise_kernel<D>, dim3((unsigned)M), dim3(256), 0, s,
                                                  gp->Xt, gp->alpha, gp->Npad, Xs, gp->d, gp->inv_len, amp,
                                                  gp->mean, gp->kf, mu));
    } else {
        const long long tiles = (M + 63) / 64;
        if (tiles > 0x7fffffffLL) return ALABI_BAD_ARGUMENT;
        ALABI_DISPATCH_DIM(db, ALABI_DISPATCH_KERNEL(gp->kf.type, hipLaunchKernelGGL((predict_mean_tile_kernel<D, GENERIC>), dim3((unsigned)tiles), dim3(256), 0, s,
                                                  gp->Xt, gp->alpha, gp->Npad, Xs, gp->d, M, gp->inv_len, amp,
                                                  gp->mean, gp->kf, mu)));
    }
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_predict_var(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var, hipStream_t s) {
    if (M <= 0) return ALABI_OK;
    const int db = dim_bucket(gp->d);
    const long long tiles = (M + 63) / 64;
    const int grid = (int)(tiles < 512 ? tiles : 512);
    const size_t need = (size_t)grid * gp->Npad * 64 * sizeof(double);
    if (need > gp->ws_bytes) {
        if (gp->ws) {
            ALABI_HIP_CHECK(hipStreamSynchronize(s));
            ALABI_HIP_CHECK(hipFree(gp->ws));
            gp->ws = nullptr; gp->ws_bytes = 0;
        }
        ALABI_HIP_CHECK(hipMalloc(&gp->ws, need));
        gp->ws_bytes = need;
    }
    const double amp = exp(gp->log_amp);
    int split = 0;   // ablation knob for tools/prof_predict.py only (results are wrong unless 0)
    if (const char* env = getenv("ALABI_PV_ABLATE")) split = atoi(env);
    ALABI_DISPATCH_DIM(db, hipLaunchKernelGGL(predict_var_kernel<D>, dim3(grid), dim3(256), 0, s, gp->L, gp->dinv, gp->Xt,
                                              gp->alpha, gp->N, gp->Npad, Xs, gp->d, M, gp->inv_len, amp,
                                              gp->mean, gp->ws, mu, var, split, gp->kf));
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

// L^-1 into the workspace, tile-major: ws[t] = L^-1[:, 64t:64t+64] as [Npad][64] (rows above block t unspecified).
int launch_factor_inverse(alabi_gp* gp, hipStream_t s) {
    const int nb = gp->Npad / 64;
    const size_t need = (size_t)gp->Npad * gp->Npad * sizeof(double);
    if (need > gp->ws_bytes) {
        if (gp->ws) {
            ALABI_HIP_CHECK(hipStreamSynchronize(s));
            ALABI_HIP_CHECK(hipFree(gp->ws));
            gp->ws = nullptr; gp->ws_bytes = 0;
        }
        ALABI_HIP_CHECK(hipMalloc(&gp->ws, need));
        gp->ws_bytes = need;
    }
    hipLaunchKernelGGL((predict_var_kernel<1, true>), dim3(nb), dim3(256), 0, s, gp->L, gp->dinv, gp->Xt, gp->alpha, gp->N,
                       gp->Npad, (const double*)nullptr, gp->d, (long long)gp->Npad, gp->inv_len, 1.0, 0.0, gp->ws,
                       (double*)nullptr, (double*)nullptr, 0, gp->kf);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
