// Autocorrelation functions of an ensemble chain on the device (gfx950): the FFT part of emcee's autocorr.integrated_time, reached
// from the reference at alabi/mcmc_utils.py:45 (sampler.get_autocorr_time(tol=0)) and alabi/core.py:2387 after every run_emcee.
//
//   per series (walker w, dimension k):  f = FFT(x - mean(x), 2 n),  acf = IFFT(|f|^2)[:n_t] / acf[0],   n = next power of two >= n_t
//   out[k][lag] = mean over the walkers of acf_{w,k}[lag]           (the Sokal window and tau = 2 cumsum - 1 follow on the host)
//
// Why not a library FFT: rocFFT compiles its kernels at run time for every NEW transform length -- 1.0-1.9 s per length on MI355X
// (tools/prof_rocfft_first_call.py) -- and the length follows the number of steps of the run, so a default run_emcee (5e4 steps,
// 0.2 s of sampling) spent ten times its sampling time in that compilation.  The chain is 1 GB and the transforms are 3e10 flops:
// nothing here needs a tuned FFT, it needs one that is simply there.
//
// Layout.  The chain is [n_t][S] (S = walkers x dimensions, contiguous): a series is strided.  A batch of series is first
// transposed to rows xt[s][0 .. n_t) (acf_transpose_kernel, LDS tiles), its means taken from the rows.  A series of length
// M = 2 n = M1 M2 is then transformed by the four-step scheme with every one-dimensional transform inside LDS (radix-2 Stockham):
//   A  acf_cols_fwd_kernel   for J columns j2 at a time: length-M1 transforms over j1 of x[j1 M2 + j2] (the zero padding and the mean
//                            are applied on load), times the twiddle w_M^(j2 k1) -> Z[s][k1][j2]              (complex scratch)
//   B  acf_rows_kernel       per row k1: length-M2 transform over j2 -> the spectrum F[k1 + M1 k2]; |F|^2; and at once the FIRST
//                            step of the inverse, which is again a transform over the contiguous index (k2) of the same row;
//                            times w_M^(-k1 j2) -> Z[s][k1][j2] in place
//   C  acf_cols_inv_kernel   for J columns: length-M1 inverse transforms over k1 -> acf[j1 M2 + j2] in natural order; lags < n_t
//                            are written to R[s][lag]
//   D  acf_reduce_kernel     out[k][lag] += sum over the batch's walkers of R[(w,k)][lag] / R[(w,k)][0]
// The spectrum never needs to be in natural order: |F|^2 is taken element by element and the inverse starts from the same layout.
// HBM traffic per batch of 256 series of 2^17 points: ~2.5 GB; a 1 GB chain takes ~15 ms.
#include <cmath>
#include <cstdlib>

#include "gp_device.hpp"

namespace alabi {

struct cplx { double re, im; };
__device__ inline cplx cmul(cplx a, cplx b) { return {fma(a.re, b.re, -a.im * b.im), fma(a.re, b.im, a.im * b.re)}; }

// exp(sign 2 pi i num / den) for 0 <= num < den (den a power of two): sincospi on the exactly representable fraction
__device__ inline cplx twiddle(long long num, long long den, double sign) {
    double s, c;
    sincospi(2.0 * (double)num / (double)den, &s, &c);
    return {c, sign * s};
}

// In-LDS radix-2 Stockham transforms of `ncol` interleaved columns of length L (element i of column c at buf[i * ncol + c]):
// log2(L) stages between two buffers, natural order in and out.  sign = -1 forward, +1 inverse (unnormalised).  Returns the buffer
// that holds the result.  Every thread of the workgroup must call it.
__device__ inline cplx* lds_fft(cplx* a, cplx* b, int L, int ncol, double sign) {
    const int work = (L >> 1) * ncol;
    for (int Ns = 1; Ns < L; Ns <<= 1) {
        for (int e = threadIdx.x; e < work; e += blockDim.x) {
            const int j = e / ncol, c = e - j * ncol;
            const int k = j & (Ns - 1);
            const cplx w = twiddle(k, 2LL * Ns, sign);
            const cplx u = a[(size_t)j * ncol + c];
            const cplx v = cmul(a[(size_t)(j + (L >> 1)) * ncol + c], w);
            const int j0 = ((j - k) << 1) + k;                       // (j / Ns) * 2 Ns + k
            b[(size_t)j0 * ncol + c] = {u.re + v.re, u.im + v.im};
            b[(size_t)(j0 + Ns) * ncol + c] = {u.re - v.re, u.im - v.im};
        }
        __syncthreads();
        cplx* t = a; a = b; b = t;
    }
    return a;
}

// chain[t][S] -> xt[s - s0][t] for the series s0 .. s0 + ns (64 x 64 tiles through LDS: coalesced on both sides)
__global__ void __launch_bounds__(256)
acf_transpose_kernel(const double* __restrict__ chain, long long n_t, int S, int s0, int ns, double* __restrict__ xt, long long ldx) {
    __shared__ double tile[64][65];
    const long long t0 = (long long)blockIdx.x * 64;
    const int sb = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;                            // r: time, c: series
        tile[r][c] = (t0 + r < n_t && sb + c < ns) ? chain[(size_t)(t0 + r) * S + s0 + sb + c] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;                            // r: series, c: time
        if (sb + r < ns && t0 + c < n_t) xt[(size_t)(sb + r) * ldx + t0 + c] = tile[c][r];
    }
}

__global__ void __launch_bounds__(256)
acf_mean_kernel(const double* __restrict__ xt, long long n_t, long long ldx, double* __restrict__ mean) {
    __shared__ double scratch[16];
    const double* x = xt + (size_t)blockIdx.x * ldx;
    double s = 0.0;
    for (long long t = threadIdx.x; t < n_t; t += 256) s += x[t];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) mean[blockIdx.x] = s / (double)n_t;
}

// A: blockIdx.x = block of J columns j2, blockIdx.y = series.  Dynamic LDS: 2 x M1 x J complex.
__global__ void __launch_bounds__(256)
acf_cols_fwd_kernel(const double* __restrict__ xt, const double* __restrict__ mean, long long n_t, long long ldx, int M1, int M2, int J,
                    cplx* __restrict__ Z) {
    extern __shared__ __attribute__((aligned(16))) unsigned char acf_smem[];
    cplx* a = reinterpret_cast<cplx*>(acf_smem);
    cplx* b = a + (size_t)M1 * J;
    const int s = blockIdx.y, jb = blockIdx.x * J;
    const double* x = xt + (size_t)s * ldx;
    const double mu = mean[s];
    for (int e = threadIdx.x; e < M1 * J; e += 256) {
        const int j1 = e / J, c = e - j1 * J;
        const long long t = (long long)j1 * M2 + jb + c;
        a[e] = {t < n_t ? x[t] - mu : 0.0, 0.0};
    }
    __syncthreads();
    cplx* r = lds_fft(a, b, M1, J, -1.0);
    cplx* Zs = Z + (size_t)s * M1 * M2;
    const long long M = (long long)M1 * M2;
    for (int e = threadIdx.x; e < M1 * J; e += 256) {
        const int k1 = e / J, c = e - k1 * J;
        const int j2 = jb + c;
        const cplx w = twiddle(((long long)j2 * k1) & (M - 1), M, -1.0);
        Zs[(size_t)k1 * M2 + j2] = cmul(r[e], w);
    }
}

// B: blockIdx.x = row k1, blockIdx.y = series.  Dynamic LDS: 2 x M2 complex.
__global__ void __launch_bounds__(256)
acf_rows_kernel(cplx* __restrict__ Z, int M1, int M2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char acf_smem[];
    cplx* a = reinterpret_cast<cplx*>(acf_smem);
    cplx* b = a + M2;
    const int k1 = blockIdx.x;
    cplx* row = Z + ((size_t)blockIdx.y * M1 + k1) * M2;
    for (int e = threadIdx.x; e < M2; e += 256) a[e] = row[e];
    __syncthreads();
    cplx* r = lds_fft(a, b, M2, 1, -1.0);                             // r[k2] = F[k1 + M1 k2]
    for (int e = threadIdx.x; e < M2; e += 256) r[e] = {fma(r[e].re, r[e].re, r[e].im * r[e].im), 0.0};
    __syncthreads();
    cplx* q = lds_fft(r, r == a ? b : a, M2, 1, +1.0);                 // q[j2] = sum_k2 |F|^2 w_M2^(+k2 j2)
    const long long M = (long long)M1 * M2;
    for (int e = threadIdx.x; e < M2; e += 256) {
        const cplx w = twiddle(((long long)k1 * e) & (M - 1), M, +1.0);
        row[e] = cmul(q[e], w);
    }
}

// C: blockIdx.x = block of J columns j2, blockIdx.y = series: acf[j1 M2 + j2] (unnormalised: the reduction divides by lag 0 anyway)
__global__ void __launch_bounds__(256)
acf_cols_inv_kernel(const cplx* __restrict__ Z, long long n_t, int M1, int M2, int J, double* __restrict__ R, long long ldr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char acf_smem[];
    cplx* a = reinterpret_cast<cplx*>(acf_smem);
    cplx* b = a + (size_t)M1 * J;
    const int s = blockIdx.y, jb = blockIdx.x * J;
    const cplx* Zs = Z + (size_t)s * M1 * M2;
    for (int e = threadIdx.x; e < M1 * J; e += 256) {
        const int k1 = e / J, c = e - k1 * J;
        a[e] = Zs[(size_t)k1 * M2 + jb + c];
    }
    __syncthreads();
    cplx* r = lds_fft(a, b, M1, J, +1.0);
    double* Rs = R + (size_t)s * ldr;
    for (int e = threadIdx.x; e < M1 * J; e += 256) {
        const int j1 = e / J, c = e - j1 * J;
        const long long lag = (long long)j1 * M2 + jb + c;
        if (lag < n_t) Rs[lag] = r[e].re;
    }
}

// D: out[k][lag] (+)= scale * sum over the batch's walkers of R[(w, k)][lag] / R[(w, k)][0]; series s of the batch = walker w0 + s / n_d,
// dimension s % n_d (the batch holds whole walkers)
__global__ void __launch_bounds__(256)
acf_reduce_kernel(const double* __restrict__ R, long long ldr, long long n_t, int nw_batch, int n_d, double scale, int first,
                  double* __restrict__ out) {
    const long long lag = (long long)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (lag >= n_t) return;
    double acc = 0.0;
    for (int w = 0; w < nw_batch; ++w) {
        const double* r = R + (size_t)(w * n_d + k) * ldr;
        acc += r[lag] / r[0];
    }
    double* o = out + (size_t)k * n_t + lag;
    *o = first ? scale * acc : fma(scale, acc, *o);
}

}  // namespace alabi

using namespace alabi;

extern "C" int alabi_chain_autocorr(const double* chain, long long n_t, int n_w, int n_d, double* acf_mean, void* stream) {
    if (!chain || !acf_mean || n_t < 1 || n_w < 1 || n_d < 1) return ALABI_BAD_ARGUMENT;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    long long n = 1;
    int p = 0;
    while (n < n_t) { n <<= 1; ++p; }
    p += 1;                                                           // M = 2 n = 2^p
    if (p > 22) return ALABI_BAD_ARGUMENT;                            // n_t <= 2^21 steps (the caller falls back beyond)
    if (p < 4) p = 4;                                                 // tiny chains: pad further (the extra zeros change nothing)
    const int M1 = 1 << (p / 2), M2 = 1 << (p - p / 2);
    const long long M = (long long)M1 * M2;
    const int S = n_w * n_d;
    // columns per workgroup of the column transforms: 2 buffers x M1 x J x 16 bytes <= 128 KB
    int J = 4096 / M1;
    if (J > 16) J = 16;
    if (J > M2) J = M2;
    // walkers per batch: <= 1 GB of complex scratch
    long long wb = ((1LL << 30) / (M * 16)) / n_d;
    if (wb < 1) wb = 1;
    if (wb > n_w) wb = n_w;
    const long long sb_max = wb * n_d;
    const long long ldx = n_t, ldr = n_t;
    const size_t bytes_Z = (size_t)sb_max * M * sizeof(cplx), bytes_x = (size_t)sb_max * ldx * sizeof(double);
    const size_t need = bytes_Z + 2 * bytes_x + (size_t)sb_max * sizeof(double) + 1024;
    void* ws = nullptr;
    size_t got = 0;
    if (dev_alloc_cached(&ws, need, &got) != (int)hipSuccess) return hip_fail(hipErrorOutOfMemory, "hipMalloc(autocorrelation workspace)", __FILE__, __LINE__);
    cplx* Z = reinterpret_cast<cplx*>(ws);
    double* xt = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + bytes_Z);
    double* R = xt + (size_t)sb_max * ldx;
    double* mean = R + (size_t)sb_max * ldr;
    const size_t lds_cols = (size_t)2 * M1 * J * sizeof(cplx), lds_rows = (size_t)2 * M2 * sizeof(cplx);
    int rc = ALABI_OK;
    auto fail = [&](hipError_t e, const char* what) { rc = hip_fail(e, what, __FILE__, __LINE__); };
    hipError_t he;
    if ((he = hipFuncSetAttribute(reinterpret_cast<const void*>(acf_cols_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072)) != hipSuccess) fail(he, "hipFuncSetAttribute");
    if ((he = hipFuncSetAttribute(reinterpret_cast<const void*>(acf_cols_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072)) != hipSuccess) fail(he, "hipFuncSetAttribute");
    if ((he = hipFuncSetAttribute(reinterpret_cast<const void*>(acf_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072)) != hipSuccess) fail(he, "hipFuncSetAttribute");
    for (int w0 = 0; w0 < n_w && rc == ALABI_OK; w0 += (int)wb) {
        const int nwb = (int)((n_w - w0) < wb ? (n_w - w0) : wb), ns = nwb * n_d, s0 = w0 * n_d;
        hipLaunchKernelGGL(acf_transpose_kernel, dim3((unsigned)((n_t + 63) / 64), (ns + 63) / 64), dim3(256), 0, s, chain, n_t, S, s0, ns, xt, ldx);
        hipLaunchKernelGGL(acf_mean_kernel, dim3(ns), dim3(256), 0, s, xt, n_t, ldx, mean);
        hipLaunchKernelGGL(acf_cols_fwd_kernel, dim3(M2 / J, ns), dim3(256), lds_cols, s, xt, mean, n_t, ldx, M1, M2, J, Z);
        hipLaunchKernelGGL(acf_rows_kernel, dim3(M1, ns), dim3(256), lds_rows, s, Z, M1, M2);
        hipLaunchKernelGGL(acf_cols_inv_kernel, dim3(M2 / J, ns), dim3(256), lds_cols, s, Z, n_t, M1, M2, J, R, ldr);
        hipLaunchKernelGGL(acf_reduce_kernel, dim3((unsigned)((n_t + 255) / 256), n_d), dim3(256), 0, s, R, ldr, n_t, nwb, n_d, 1.0 / n_w,
                           w0 == 0 ? 1 : 0, acf_mean);
        if ((he = hipGetLastError()) != hipSuccess) fail(he, "autocorrelation kernels");
    }
    // the workspace goes back to the cache only when nothing in flight uses it any more
    if ((he = hipStreamSynchronize(s)) != hipSuccess && rc == ALABI_OK) fail(he, "hipStreamSynchronize");
    dev_cache_give(ws, got);
    return rc;
}
