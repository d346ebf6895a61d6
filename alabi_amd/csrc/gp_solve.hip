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
#include <cstdlib>

#include <atomic>
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
    #pragma unroll
    for (int e_ = 0; e_ < 16; ++e_) {
        const int e = tid + 256 * e_;
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
    #pragma unroll
    for (int e_ = 0; e_ < 16; ++e_) {
        const int e = tid + 256 * e_;
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

// ---------------------------------------------------------------------------------------------------
// Single-launch dataflow solve.  Workgroup i owns block row i.  Forward: it folds L[i,k] z_k into its right-hand side
// as soon as z_k appears (k ascending), solves its diagonal block and publishes z_i; backward: the same with
// L[k,i]^T alpha_k for k descending.  z and alpha are handed over through buffers pre-filled with a sentinel NaN:
// every word is one aligned 8-byte write-through (sc1) store and the consumer polls the words themselves with sc1
// loads -- the data is the flag (cdna_hip_programming.md Guideline 16, R2).  The dependency graph is acyclic
// (fwd(i) <- fwd(k<i); bwd(i) <- fwd(i), bwd(k>i)) and nb <= 157 workgroups are all resident, so there is no deadlock;
// every spin is bounded and a time-out makes all workgroups leave (the host then runs the per-step kernels).
// Critical path: nb x (64-step diagonal solve + one hand-off) per sweep instead of nb kernel launches.
#define ALABI_TRSV_EMPTY 0x7FF8A1AB1D15EA5Eull

__device__ inline bool trsv_wait_block(const unsigned long long* src, double* dst_lds, int* err, int spin_limit, int lane) {
    unsigned long long v = ALABI_TRSV_EMPTY;
    int spins = 0;
    bool ok = true;
    while (true) {
        if (v == ALABI_TRSV_EMPTY) v = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v != ALABI_TRSV_EMPTY)) break;
        if (++spins > spin_limit ||
            ((spins & 63) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (ok) dst_lds[lane] = __longlong_as_double((long long)v);
    else if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
}

__global__ void __launch_bounds__(256)
trsv_stream_kernel(const double* __restrict__ L, int ld, int nb, const double* __restrict__ dinv,
                   const double* __restrict__ r, unsigned long long* __restrict__ zbuf,
                   unsigned long long* __restrict__ abuf, double* __restrict__ alpha, int* __restrict__ err, int spin_limit) {
    __shared__ double lkk[64][65];
    __shared__ double xs[64];        // the z_k / alpha_k block being folded in
    __shared__ double rhs[64];
    __shared__ double part[4][64];
    __shared__ int abort_s;
    const int tid = threadIdx.x, i = blockIdx.x;
    const double* Lb = L + (size_t)(i * 64) * ld + i * 64;
    #pragma unroll
    for (int e_ = 0; e_ < 16; ++e_) {
        const int e = tid + 256 * e_;
        int rr = e >> 6, c = e & 63;
        lkk[rr][c] = (c <= rr) ? Lb[(size_t)rr * ld + c] : 0.0;
    }
    if (tid == 0) abort_s = 0;
    const double di = (tid < 64) ? dinv[i * 64 + tid] : 0.0;
    __syncthreads();
    // The 64-step substitutions below are chains of dependent operations in ONE wave: an LDS operand inside the chain costs a
    // full LDS round trip per step (the compiler places each read next to its use).  The lane's row of L_ii (forward) and its
    // column (backward) are therefore read into registers beforehand, while the hand-offs of the earlier blocks are awaited.
    double lrow[64], lcol[64];
    if (tid < 64) {
#pragma unroll
        for (int j = 0; j < 64; ++j) { lrow[j] = lkk[tid][j]; lcol[j] = lkk[j][tid]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- forward: rhs_i = r_i - sum_{k<i} L[i,k] z_k ; thread = (row, quarter of the 64 columns)
    {
        const int row = tid >> 2, q = tid & 3;
        double acc = 0.0;
        for (int k = 0; k < i; ++k) {
            const double* Lrow = L + (size_t)(i * 64 + row) * ld + k * 64 + q * 16;
            double lv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) lv[c] = Lrow[c];            // issued before the wait: latency hidden
            if (tid < 64 && !trsv_wait_block(zbuf + k * 64, xs, err, spin_limit, tid)) abort_s = 1;
            __syncthreads();
            if (abort_s) return;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc = fma(lv[c], xs[q * 16 + c], acc);
            __syncthreads();                                         // xs is overwritten by the next block
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (q == 0) rhs[row] = r[i * 64 + row] - acc;
    }
    __syncthreads();
    double zi = 0.0;   // lane j of wave 0 keeps z_i[j]
    if (tid < 64) {
        double v = rhs[tid];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const double zj = lane_bcast(v * di, j);
            if (tid == j) v = zj;
            if (tid > j) v = fma(-lrow[j], zj, v);
        }
        zi = v;
        __hip_atomic_store(zbuf + i * 64 + tid, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---------------- backward: alpha_i = L_ii^-T (z_i - sum_{k>i} L[k,i]^T alpha_k); lanes run along the row index
    {
        const int rr = tid & 63, q = tid >> 6;
        double acc = 0.0;
        for (int k = nb - 1; k > i; --k) {
            const double* Lc = L + (size_t)(k * 64 + q * 16) * ld + i * 64 + rr;
            double lv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) lv[c] = Lc[(size_t)c * ld];
            if (tid < 64 && !trsv_wait_block(abuf + k * 64, xs, err, spin_limit, tid)) abort_s = 1;
            __syncthreads();
            if (abort_s) return;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc = fma(lv[c], xs[q * 16 + c], acc);
            __syncthreads();
        }
        part[q][rr] = acc;
    }
    __syncthreads();
    if (tid < 64) {
        double v = zi - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
#pragma unroll
        for (int j = 63; j >= 0; --j) {
            const double aj = lane_bcast(v * di, j);
            if (tid == j) v = aj;
            if (tid < j) v = fma(-lcol[j], aj, v);
        }
        __hip_atomic_store(abuf + i * 64 + tid, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        alpha[i * 64 + tid] = v;
    }
}

__global__ void __launch_bounds__(256)
trsv_fill_kernel(unsigned long long* __restrict__ a, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = ALABI_TRSV_EMPTY;
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
    const char* env = getenv("ALABI_TRSV_STREAM");
    // A time-out is remembered process-wide for a while (concurrent solves from the CV threads can starve each other of
    // CUs: every later set_y would otherwise burn the whole spin budget again before falling back).
    static std::atomic<int> stream_penalty{0};
    const bool penalised = stream_penalty.load(std::memory_order_relaxed) > 0;
    if (penalised) stream_penalty.fetch_sub(1, std::memory_order_relaxed);
    if (nb <= 200 && gp->flags && !penalised && !(env && env[0] == '0')) {
        // single-launch dataflow solve (all nb workgroups resident); z / alpha hand-off buffers live in `work`
        unsigned long long* zbuf = reinterpret_cast<unsigned long long*>(z);
        unsigned long long* abuf = reinterpret_cast<unsigned long long*>(gp->work2);
        ALABI_HIP_CHECK(hipMemsetAsync(gp->flags, 0, sizeof(int), s));
        hipLaunchKernelGGL(trsv_fill_kernel, dim3((gp->Npad + 255) / 256), dim3(256), 0, s, zbuf, gp->Npad);
        hipLaunchKernelGGL(trsv_fill_kernel, dim3((gp->Npad + 255) / 256), dim3(256), 0, s, abuf, gp->Npad);
        hipLaunchKernelGGL(trsv_stream_kernel, dim3(nb), dim3(256), 0, s, gp->L, ld, nb, gp->dinv, r, zbuf, abuf, gp->alpha,
                           gp->flags, 1 << 20);
        ALABI_LAUNCH_CHECK();
        int flag = 0;
        ALABI_HIP_CHECK(hipMemcpyAsync(&flag, gp->flags, sizeof(int), hipMemcpyDeviceToHost, s));
        ALABI_HIP_CHECK(hipStreamSynchronize(s));
        if (!flag) return ALABI_OK;
        // timed out (workgroups not co-resident?): fall through to the launch-per-step kernels, and skip the dataflow
        // kernel for the next 256 solves of this process
        stream_penalty.store(256, std::memory_order_relaxed);
        hipLaunchKernelGGL(residual_kernel, dim3((gp->Npad + 255) / 256), dim3(256), 0, s, gp->y, gp->N, gp->Npad, gp->mean, r);
    }
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
