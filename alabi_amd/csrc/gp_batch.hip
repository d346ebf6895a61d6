// Batched GP fit + held-out mean for the hyper-parameter search (gfx950).
//
// Reference: alabi/gp_utils.py:511-637 (_evaluate_candidate_worker: per fold gp.compute(train) -> log_likelihood -> predict(val))
// mapped over candidates by a process pool at gp_utils.py:640-700 (100 + 50 + 25 candidates x 5 folds = 875 fits of 0.8 N points per
// init_gp(hyperopt_method="cv"), alabi/core.py:1287-1305).  One fit of N = 1600 is a chain of 25 dependent block columns that
// leaves 92 % of the matrix cores idle (profiles/r03_*), and 875 of them one after the other cost as much as 1e5 MCMC steps.
// Here a whole stage of the search is ONE call: every job (hyper-parameter vector, training rows, validation rows) shares the
// resident inputs X / y and
//   batch_prepare_kernel   gathers and scales each job's training rows                      (grid: rows x jobs)
//   batch_assemble_kernel  writes each job's kernel matrix (assemble_tile: the bits of the single-matrix path)
//   chol_tasks8_batch_kernel  factorises ALL of them in one launch of the task queue (gp_cholesky.hip: interleaved task lists,
//                          one list and head counter per XCD, chains of different matrices side by side)
//   batch_solve_kernel     alpha = K^-1 (y - m), log-determinant, (y - m)^T alpha            (one workgroup per job)
//   batch_predict_kernel   mean at the validation rows                                       (32 queries per workgroup)
// followed by ONE read-back.  Jobs are processed in chunks that fit the workspace (ALABI_BATCH_BYTES, default 16 GiB).
#include <cmath>
#include <cstdlib>
#include <vector>

#include "gp_device.hpp"

namespace alabi {

struct BatchJob {
    double* A;            // [Npad, Npad] kernel matrix -> factor
    double* Xt;           // [dbucket, Npad] scaled training inputs
    double* alpha;        // [Npad]
    double* dinv;         // [Npad]
    int* info;            // [1]
    double* red;          // [2] log-determinant, (y - m)^T alpha
    double* mu;           // [nval] held-out mean (caller's buffer)
    const int* train;     // [N] rows of X
    const int* val;       // [nval]
    double amp, wn, mean, kalpha;
    int N, Npad, nval, pad_;
    double inv_len[ALABI_MAX_DIM];
};

// A row index outside [0, n_rows) must never reach an address: it reads as an all-zero row / a zero target and raises *bad, which
// the host turns into ALABI_BAD_ARGUMENT after the call (the row lists come from the caller; alabi_cv_fold_lists builds valid ones).
__device__ inline int batch_row(const int* __restrict__ idx, int i, int n_rows, int* __restrict__ bad) {
    const int r = idx[i];
    if ((unsigned)r < (unsigned)n_rows) return r;
    *bad = 1;
    return -1;
}

__global__ void __launch_bounds__(256)
batch_prepare_kernel(const BatchJob* __restrict__ jobs, const double* __restrict__ X, int d, int dbucket, int n_rows, int* __restrict__ bad) {
    const BatchJob& j = jobs[blockIdx.y];
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n == 0) *j.info = 0;
    if (n >= j.Npad) return;
    const int src = n < j.N ? batch_row(j.train, n, n_rows, bad) : -1;
    for (int k = 0; k < dbucket; ++k) {
        double v = 0.0;
        if (src >= 0 && k < d) v = X[(size_t)src * d + k] * j.inv_len[k];
        j.Xt[(size_t)k * j.Npad + n] = v;
    }
}

__global__ void __launch_bounds__(256)
batch_assemble_kernel(const BatchJob* __restrict__ jobs, int d, int kernel_type) {
    __shared__ double xi[ALABI_MAX_DIM][64];
    __shared__ double xj[ALABI_MAX_DIM][64];
    const BatchJob& j = jobs[blockIdx.y];
    const int nb = j.Npad / 64, t = blockIdx.x;
    if (t >= nb * (nb + 1) / 2) return;
    int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    assemble_tile(xi, xj, j.Xt, j.N, j.Npad, d, j.amp, j.wn, KernelFn{kernel_type, j.kalpha}, j.A, bi, bj);
}

// alpha = L^-T L^-1 (y - m) for one job per workgroup: both sweeps by 64-row blocks with the whole right-hand side in LDS.
// Forward (left-looking): the block's rows dotted with the solved part -- lanes along the columns, 512 contiguous bytes per row
// and instruction, one DPP wave sum per row -- then the 64-step substitution of the diagonal block in wave 0 (its row of L_ii in
// registers, solved entries broadcast with v_readlane).  Backward: the block's COLUMNS of the rows below, lanes along the columns
// again (coalesced), four partial sums per column, the transposed substitution.  2 N^2 flops reading the lower triangle twice:
// bound by what one CU can pull (~0.3 ms at N = 1600), which 256 jobs at a time hide.
__global__ void __launch_bounds__(256)
batch_solve_kernel(const BatchJob* __restrict__ jobs, const double* __restrict__ y, int n_rows, int* __restrict__ bad) {
    extern __shared__ double bs_z[];                     // [Npad]
    __shared__ double lkk[64][65];
    __shared__ double rhs[64];
    __shared__ double part[4][64];
    __shared__ double scratch[16];
    const BatchJob& j = jobs[blockIdx.x];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int N = j.N, Npad = j.Npad, nb = Npad / 64, ld = Npad;
    const double* __restrict__ L = j.A;
    if (*j.info != 0) {                                  // not positive definite: no solve (the host reports the pivot)
        if (tid < 2) j.red[tid] = __builtin_nan("");
        return;
    }
    auto target = [&](int n) { const int r = batch_row(j.train, n, n_rows, bad); return r >= 0 ? y[r] : j.mean; };
    for (int n = tid; n < Npad; n += 256) bs_z[n] = n < N ? target(n) - j.mean : 0.0;
    __syncthreads();
    for (int i = 0; i < nb; ++i) {
        const double* Lb = L + (size_t)(i * 64) * ld + i * 64;
#pragma unroll
        for (int e_ = 0; e_ < 16; ++e_) {
            const int e = tid + 256 * e_, r = e >> 6, c = e & 63;
            lkk[r][c] = (c <= r) ? Lb[(size_t)r * ld + c] : 0.0;
        }
        for (int rr = 0; rr < 16; rr += 4) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const double* p = L + (size_t)(i * 64 + w * 16 + rr) * ld + lane;
#pragma unroll 4
            for (int m = 0; m < i; ++m) {
                const double zv = bs_z[64 * m + lane];
                a0 = fma(p[64 * m], zv, a0);
                a1 = fma(p[(size_t)ld + 64 * m], zv, a1);
                a2 = fma(p[2 * (size_t)ld + 64 * m], zv, a2);
                a3 = fma(p[3 * (size_t)ld + 64 * m], zv, a3);
            }
            a0 = wave_sum_dpp(a0); a1 = wave_sum_dpp(a1); a2 = wave_sum_dpp(a2); a3 = wave_sum_dpp(a3);
            if (lane == 63) { rhs[w * 16 + rr] = a0; rhs[w * 16 + rr + 1] = a1; rhs[w * 16 + rr + 2] = a2; rhs[w * 16 + rr + 3] = a3; }
        }
        __syncthreads();
        if (w == 0) {
            double lrow[64];
#pragma unroll
            for (int c = 0; c < 64; ++c) lrow[c] = lkk[lane][c];
            const double di = j.dinv[i * 64 + lane];
            double v = bs_z[i * 64 + lane] - rhs[lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 64; ++c) {
                const double zc = lane_bcast(v * di, c);
                if (lane == c) v = zc;
                if (lane > c) v = fma(-lrow[c], zc, v);
            }
            bs_z[i * 64 + lane] = v;
        }
        __syncthreads();
    }
    for (int i = nb - 1; i >= 0; --i) {
        const double* Lb = L + (size_t)(i * 64) * ld + i * 64;
#pragma unroll
        for (int e_ = 0; e_ < 16; ++e_) {
            const int e = tid + 256 * e_, r = e >> 6, c = e & 63;
            lkk[r][c] = (c <= r) ? Lb[(size_t)r * ld + c] : 0.0;
        }
        double acc = 0.0;
        {
            const double* p = L + (size_t)i * 64 + lane;
#pragma unroll 8
            for (int r = (i + 1) * 64 + w; r < Npad; r += 4) acc = fma(p[(size_t)r * ld], bs_z[r], acc);
        }
        part[w][lane] = acc;
        __syncthreads();
        if (w == 0) {
            double lcol[64];
#pragma unroll
            for (int c = 0; c < 64; ++c) lcol[c] = lkk[c][lane];
            const double di = j.dinv[i * 64 + lane];
            double v = bs_z[i * 64 + lane] - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 63; c >= 0; --c) {
                const double ac = lane_bcast(v * di, c);
                if (lane == c) v = ac;
                if (lane < c) v = fma(-lcol[c], ac, v);
            }
            bs_z[i * 64 + lane] = v;
        }
        __syncthreads();
    }
    double ld_sum = 0.0, ra = 0.0;
    for (int n = tid; n < Npad; n += 256) {
        const double a = bs_z[n];
        j.alpha[n] = a;
        if (n < N) {
            ld_sum += log(L[(size_t)n * ld + n]);
            ra = fma(target(n) - j.mean, a, ra);
        }
    }
    ld_sum = block_sum(ld_sum, scratch);
    ra = block_sum(ra, scratch);
    if (tid == 0) { j.red[0] = 2.0 * ld_sum; j.red[1] = ra; }
}

// mu*[q] = amp sum_n alpha_n f(|x_q - x_n|^2) + mean for 32 validation rows per workgroup; a thread keeps 32 squared distances of
// its training point in registers (coordinates outermost: one coalesced load of the point, 32 LDS broadcasts of the queries).
#define ALABI_BATCH_QT 32
template <bool GENERIC>
__global__ void __launch_bounds__(256)
batch_predict_kernel(const BatchJob* __restrict__ jobs, const double* __restrict__ X, int d, int kernel_type, int n_rows, int* __restrict__ bad) {
    __shared__ double xq[ALABI_MAX_DIM][ALABI_BATCH_QT];
    __shared__ double part[4][ALABI_BATCH_QT];
    const BatchJob& j = jobs[blockIdx.y];
    const int q0 = blockIdx.x * ALABI_BATCH_QT;
    if (q0 >= j.nval) return;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int nq = j.nval - q0 < ALABI_BATCH_QT ? j.nval - q0 : ALABI_BATCH_QT;
    if (*j.info != 0) {
        if (tid < nq) j.mu[q0 + tid] = __builtin_nan("");
        return;
    }
    for (int e = tid; e < d * ALABI_BATCH_QT; e += 256) {
        const int q = e / d, k = e % d;
        const int r = q < nq ? batch_row(j.val, q0 + q, n_rows, bad) : -1;
        xq[k][q] = r >= 0 ? X[(size_t)r * d + k] * j.inv_len[k] : 0.0;
    }
    __syncthreads();
    const KernelFn kf{kernel_type, j.kalpha};
    double acc[ALABI_BATCH_QT];
#pragma unroll
    for (int q = 0; q < ALABI_BATCH_QT; ++q) acc[q] = 0.0;
    const int Npad = j.Npad;
    for (int n = tid; n < Npad; n += 256) {
        const double a = j.alpha[n];                     // 0 in the padding
        double r2[ALABI_BATCH_QT];
#pragma unroll
        for (int q = 0; q < ALABI_BATCH_QT; ++q) r2[q] = 0.0;
        for (int k = 0; k < d; ++k) {
            const double xv = j.Xt[(size_t)k * Npad + n];
#pragma unroll
            for (int q = 0; q < ALABI_BATCH_QT; ++q) {
                const double df = xv - xq[k][q];
                r2[q] = fma(df, df, r2[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < ALABI_BATCH_QT; ++q) acc[q] = fma(a, radial<GENERIC>(r2[q], kf), acc[q]);
    }
#pragma unroll
    for (int q = 0; q < ALABI_BATCH_QT; ++q) {
        const double s = wave_sum_dpp(acc[q]);
        if (lane == 63) part[w][q] = s;
    }
    __syncthreads();
    if (tid < nq) j.mu[q0 + tid] = fma(j.amp, (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]), j.mean);
}

// Row lists of a k-fold split (sklearn's KFold.split order: ascending rows on both sides) for C candidates at once: fold_of[c][r] = the
// fold of row r under candidate c's shuffle (-1: in no fold).  One workgroup per (candidate, fold) writes the fold's rows to
// val_idx[val_off[c k + f] ..) and every other used row to train_idx[train_off[c k + f] ..): an ordered compaction -- each thread
// owns a contiguous run of rows, a block scan of the two counts gives its write positions.
__global__ void __launch_bounds__(256)
batch_fold_lists_kernel(const signed char* __restrict__ fold_of, int n, int k, const long long* __restrict__ train_off,
                        const long long* __restrict__ val_off, int* __restrict__ train_idx, int* __restrict__ val_idx) {
    __shared__ int cv[256], ct[256];
    const int job = blockIdx.x, c = job / k, f = job % k, tid = threadIdx.x;
    const signed char* fo = fold_of + (size_t)c * n;
    const int per = (n + 255) / 256, r0 = tid * per, r1 = min(n, r0 + per);
    int nv = 0, nt = 0;
    for (int r = r0; r < r1; ++r) { const int v = fo[r]; nv += v == f; nt += (v >= 0) & (v != f); }
    cv[tid] = nv; ct[tid] = nt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {               // inclusive scan (Hillis-Steele: 8 rounds)
        const int av = tid >= off ? cv[tid - off] : 0, at = tid >= off ? ct[tid - off] : 0;
        __syncthreads();
        cv[tid] += av; ct[tid] += at;
        __syncthreads();
    }
    int* vo = val_idx + val_off[job] + (cv[tid] - nv);
    int* to = train_idx + train_off[job] + (ct[tid] - nt);
    for (int r = r0; r < r1; ++r) {
        const int v = fo[r];
        if (v == f) *vo++ = r;
        else if (v >= 0) *to++ = r;
    }
}

__global__ void __launch_bounds__(256)
batch_copy_factor_kernel(const double* __restrict__ L, int ld, int N, double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * N) return;
    const int r = (int)(e / N), c = (int)(e % N);
    out[e] = (c <= r) ? L[(size_t)r * ld + c] : 0.0;
}

}  // namespace alabi

using namespace alabi;

struct alabi_gp_batch {
    int d = 0, kernel_type = 0;
    size_t budget = 0;                 // bytes of matrix workspace
    double* mats = nullptr; size_t mats_bytes = 0;
    double* aux = nullptr; size_t aux_bytes = 0;         // per job: Xt, alpha, dinv, red
    int* info = nullptr; size_t info_cap = 0;
    int* bad = nullptr;                // [1] a row index out of range was seen
    BatchJob* jobs = nullptr; size_t jobs_cap = 0;
    double* host_red = nullptr; int* host_info = nullptr; size_t host_cap = 0;   // pinned read-back
    CholBatchQueue queue;
    // the last chunk (alabi_gp_batch_get_factor)
    int last_first = 0, last_count = 0;
    std::vector<BatchJob> last_jobs;
    int timeouts = 0;                  // chunks that fell back to the launch-per-step factorisation
};

static inline hipStream_t bstream(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

int alabi_gp_batch_create(int d, int kernel_type, long long workspace_bytes, alabi_gp_batch** out) {
    if (!out || d <= 0 || d > ALABI_MAX_DIM || kernel_type < 0 || kernel_type > 3 || workspace_bytes < 0) return ALABI_BAD_ARGUMENT;
    alabi_gp_batch* b = new (std::nothrow) alabi_gp_batch();
    if (!b) return ALABI_BAD_ARGUMENT;
    b->d = d; b->kernel_type = kernel_type;
    // matrices in flight per launch: 16 GiB hold every fold of a C3-sized search stage (500 x 20 MB) and 32 matrices of N = 8000
    // (measured at N = 8000: 12 per launch 5.7 ms per fit, 32 per launch 4.6 ms); never more than half of what is free
    size_t budget = (size_t)16 << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 2 < budget) budget = free_b / 2;
    if (const char* e = getenv("ALABI_BATCH_BYTES")) { const long long v = atoll(e); if (v > 0) budget = (size_t)v; }
    if (workspace_bytes > 0) budget = (size_t)workspace_bytes;
    b->budget = budget;
    *out = b;
    return ALABI_OK;
}

int alabi_gp_batch_destroy(alabi_gp_batch* b) {
    if (!b) return ALABI_OK;
    (void)hipDeviceSynchronize();
    if (b->mats) (void)hipFree(b->mats);
    if (b->aux) (void)hipFree(b->aux);
    if (b->info) (void)hipFree(b->info);
    if (b->bad) (void)hipFree(b->bad);
    if (b->jobs) (void)hipFree(b->jobs);
    if (b->host_red) (void)hipHostFree(b->host_red);
    if (b->host_info) (void)hipHostFree(b->host_info);
    chol_batch_free(b->queue);
    delete b;
    return ALABI_OK;
}

int alabi_gp_batch_timeouts(alabi_gp_batch* b, int* count) {
    if (!b || !count) return ALABI_BAD_ARGUMENT;
    *count = b->timeouts;
    return ALABI_OK;
}

static int grow(void** p, size_t* have, size_t need, hipStream_t s) {
    if (*have >= need) return ALABI_OK;
    if (*p) { ALABI_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(*p); *p = nullptr; *have = 0; }
    ALABI_HIP_CHECK(hipMalloc(p, need));
    *have = need;
    return ALABI_OK;
}

// one chunk of jobs [c0, c1) whose matrices fit the workspace
static int run_chunk(alabi_gp_batch* b, const double* X, const double* y, int n_rows, int c0, int c1, const double* hyper, const int* train_idx,
                     const long long* train_off, const int* val_idx, const long long* val_off, double* mu_val, double* nll, int* status,
                     hipStream_t s) {
    const int B = c1 - c0, d = b->d, db = dim_bucket(d), hs = 4 + d;
    std::vector<BatchJob> hj(B);
    size_t mat_doubles = 0, aux_doubles = 0;
    int max_npad = 0, max_nval = 0;
    for (int q = 0; q < B; ++q) {
        const int job = c0 + q;
        const int N = (int)(train_off[job + 1] - train_off[job]), nval = (int)(val_off[job + 1] - val_off[job]);
        const int Npad = round_up(N, ALABI_BLK);
        mat_doubles += (size_t)Npad * Npad;
        aux_doubles += (size_t)(db + 2) * Npad + 2;
        if (Npad > max_npad) max_npad = Npad;
        if (nval > max_nval) max_nval = nval;
    }
    int st;
    { void* p = b->mats; if ((st = grow(&p, &b->mats_bytes, mat_doubles * sizeof(double), s)) != ALABI_OK) return st; b->mats = (double*)p; }
    { void* p = b->aux; if ((st = grow(&p, &b->aux_bytes, aux_doubles * sizeof(double), s)) != ALABI_OK) return st; b->aux = (double*)p; }
    { void* p = b->info; size_t have = b->info_cap * sizeof(int); if ((st = grow(&p, &have, (size_t)B * sizeof(int), s)) != ALABI_OK) return st; b->info = (int*)p; b->info_cap = have / sizeof(int); }
    { void* p = b->jobs; size_t have = b->jobs_cap * sizeof(BatchJob); if ((st = grow(&p, &have, (size_t)B * sizeof(BatchJob), s)) != ALABI_OK) return st; b->jobs = (BatchJob*)p; b->jobs_cap = have / sizeof(BatchJob); }
    if (b->host_cap < (size_t)B) {
        ALABI_HIP_CHECK(hipStreamSynchronize(s));
        if (b->host_red) (void)hipHostFree(b->host_red);
        if (b->host_info) (void)hipHostFree(b->host_info);
        b->host_red = nullptr; b->host_info = nullptr; b->host_cap = 0;
        ALABI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&b->host_red), (size_t)B * 2 * sizeof(double), hipHostMallocDefault));
        ALABI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&b->host_info), ((size_t)B + 1) * sizeof(int), hipHostMallocDefault));
        b->host_cap = B;
    }
    // the reductions of all jobs are contiguous at the end of `aux` so that one copy brings them back
    double* red0 = b->aux + (aux_doubles - 2 * (size_t)B);
    size_t mo = 0, ao = 0;
    std::vector<int> ld(B);
    std::vector<double*> pa(B), pd(B);
    std::vector<int*> pi(B);
    for (int q = 0; q < B; ++q) {
        const int job = c0 + q;
        BatchJob& j = hj[q];
        const double* h = hyper + (size_t)job * hs;
        j.N = (int)(train_off[job + 1] - train_off[job]);
        j.nval = (int)(val_off[job + 1] - val_off[job]);
        j.Npad = round_up(j.N, ALABI_BLK);
        j.A = b->mats + mo; mo += (size_t)j.Npad * j.Npad;
        j.Xt = b->aux + ao; ao += (size_t)db * j.Npad;
        j.alpha = b->aux + ao; ao += j.Npad;
        j.dinv = b->aux + ao; ao += j.Npad;
        j.red = red0 + 2 * (size_t)q;
        j.info = b->info + q;
        j.train = train_idx + train_off[job];
        j.val = val_idx ? val_idx + val_off[job] : nullptr;
        j.mu = mu_val ? mu_val + val_off[job] : nullptr;
        j.mean = h[0]; j.wn = std::exp(h[1]); j.amp = std::exp(h[2]); j.kalpha = std::exp(h[3]);
        for (int k = 0; k < ALABI_MAX_DIM; ++k) j.inv_len[k] = k < d ? std::exp(-0.5 * h[4 + k]) : 0.0;
        ld[q] = j.Npad; pa[q] = j.A; pd[q] = j.dinv; pi[q] = j.info;
    }
    ALABI_HIP_CHECK(hipMemcpyAsync(b->jobs, hj.data(), (size_t)B * sizeof(BatchJob), hipMemcpyHostToDevice, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));                                    // `hj` is pageable: staged before the call returns, but keep it simple
    if ((st = chol_batch_prepare(b->queue, B, ld.data(), pa.data(), pd.data(), pi.data(), s)) != ALABI_OK) return st;
    const int max_nb = max_npad / 64, max_tiles = max_nb * (max_nb + 1) / 2;
    auto assemble = [&]() {
        hipLaunchKernelGGL(batch_prepare_kernel, dim3((max_npad + 255) / 256, B), dim3(256), 0, s, b->jobs, X, d, db, n_rows, b->bad);
        hipLaunchKernelGGL(batch_assemble_kernel, dim3(max_tiles, B), dim3(256), 0, s, b->jobs, d, b->kernel_type);
    };
    if (!b->bad) ALABI_HIP_CHECK(hipMalloc(&b->bad, sizeof(int)));
    ALABI_HIP_CHECK(hipMemsetAsync(b->bad, 0, sizeof(int), s));
    assemble();
    ALABI_LAUNCH_CHECK();
    const char* forced = getenv("ALABI_BATCH_QUEUE");                            // tests: 0 = the launch-per-step factorisation per matrix
    bool queued = !(forced && forced[0] == '0');
    if (queued && (st = chol_batch_launch(b->queue, s)) != ALABI_OK) return st;
    if (queued) {
        ALABI_HIP_CHECK(hipMemcpyAsync(&b->host_info[B], b->queue.ctl + 1, sizeof(int), hipMemcpyDeviceToHost, s));
        ALABI_HIP_CHECK(hipStreamSynchronize(s));
        if (b->host_info[B] != 0) {                                              // a wait ran out: the matrices are undefined
            b->timeouts++;
            queued = false;
            assemble();
        }
    }
    if (!queued)
        for (int q = 0; q < B; ++q)
            if ((st = launch_cholesky_steps(hj[q].A, hj[q].Npad, hj[q].info, hj[q].dinv, s)) != ALABI_OK) return st;
    if ((size_t)max_npad * sizeof(double) > 24 * 1024)                           // beyond the default dynamic LDS limit together with the static arrays
        ALABI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(batch_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 12288 * 8));
    hipLaunchKernelGGL(batch_solve_kernel, dim3(B), dim3(256), (size_t)max_npad * sizeof(double), s, b->jobs, y, n_rows, b->bad);
    if (max_nval > 0 && mu_val) {
        dim3 grid((max_nval + ALABI_BATCH_QT - 1) / ALABI_BATCH_QT, B);
        if (b->kernel_type == 0) hipLaunchKernelGGL(batch_predict_kernel<false>, grid, dim3(256), 0, s, b->jobs, X, d, b->kernel_type, n_rows, b->bad);
        else hipLaunchKernelGGL(batch_predict_kernel<true>, grid, dim3(256), 0, s, b->jobs, X, d, b->kernel_type, n_rows, b->bad);
    }
    ALABI_LAUNCH_CHECK();
    ALABI_HIP_CHECK(hipMemcpyAsync(b->host_red, red0, (size_t)B * 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipMemcpyAsync(b->host_info, b->info, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipMemcpyAsync(&b->host_info[B], b->bad, sizeof(int), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));
    if (b->host_info[B] != 0) { g_last_error = "alabi_gp_batch_fit_predict: a row index lies outside [0, n)"; return ALABI_BAD_ARGUMENT; }
    for (int q = 0; q < B; ++q) {
        status[c0 + q] = b->host_info[q];
        nll[c0 + q] = b->host_info[q] != 0 ? INFINITY
                                           : 0.5 * b->host_red[2 * q + 1] + 0.5 * b->host_red[2 * q] + 0.5 * hj[q].N * std::log(2.0 * 3.141592653589793);
    }
    b->last_first = c0; b->last_count = B; b->last_jobs.swap(hj);
    return ALABI_OK;
}

int alabi_gp_batch_fit_predict(alabi_gp_batch* b, const double* X, const double* y, int n, int njobs, const double* hyper,
                               const int* train_idx, const long long* train_off, const int* val_idx, const long long* val_off,
                               double* mu_val, double* nll, int* status, void* stream) {
    if (!b || !X || !y || n <= 0 || njobs <= 0 || !hyper || !train_idx || !train_off || !val_off || !nll || !status) return ALABI_BAD_ARGUMENT;
    const int hs = 4 + b->d;
    for (int job = 0; job < njobs; ++job) {
        const long long N = train_off[job + 1] - train_off[job], nv = val_off[job + 1] - val_off[job];
        if (N <= 0 || N > 12288 || nv < 0 || (nv > 0 && (!val_idx || !mu_val))) return ALABI_BAD_ARGUMENT;
        for (int k = 0; k < hs; ++k)
            if (!std::isfinite(hyper[(size_t)job * hs + k])) return ALABI_BAD_ARGUMENT;
    }
    hipStream_t s = bstream(stream);
    // chunks of about equal size (48 jobs with room for 32: 24 + 24, not 32 + 16: every launch ends with a tail in which the last
    // matrices' chains run alone)
    size_t total = 0;
    for (int job = 0; job < njobs; ++job) {
        const size_t Npad = (size_t)round_up((int)(train_off[job + 1] - train_off[job]), ALABI_BLK);
        total += Npad * Npad * sizeof(double);
    }
    const size_t nchunks = (total + b->budget - 1) / b->budget;
    const size_t target = nchunks > 0 ? (total + nchunks - 1) / nchunks : total;
    int c0 = 0;
    while (c0 < njobs) {
        size_t bytes = 0;
        int c1 = c0;
        while (c1 < njobs && c1 - c0 < 8192) {
            const size_t Npad = (size_t)round_up((int)(train_off[c1 + 1] - train_off[c1]), ALABI_BLK);
            if (c1 > c0 && (bytes + Npad * Npad * sizeof(double) > b->budget || bytes >= target)) break;
            bytes += Npad * Npad * sizeof(double);
            ++c1;
        }
        const int st = run_chunk(b, X, y, n, c0, c1, hyper, train_idx, train_off, val_idx, val_off, mu_val, nll, status, s);
        if (st != ALABI_OK) return st;
        c0 = c1;
    }
    return ALABI_OK;
}

int alabi_cv_fold_lists(const signed char* fold_of, int ncand, int n, int k, const long long* train_off, const long long* val_off,
                        int* train_idx, int* val_idx, void* stream) {
    if (!fold_of || ncand <= 0 || n <= 0 || k < 2 || k > 127 || !train_off || !val_off || !train_idx || !val_idx) return ALABI_BAD_ARGUMENT;
    hipLaunchKernelGGL(batch_fold_lists_kernel, dim3((unsigned)ncand * k), dim3(256), 0, bstream(stream), fold_of, n, k, train_off, val_off,
                       train_idx, val_idx);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int alabi_gp_batch_get_factor(alabi_gp_batch* b, int job, double* L_out, void* stream) {
    if (!b || !L_out || job < b->last_first || job >= b->last_first + b->last_count) return ALABI_BAD_ARGUMENT;
    const BatchJob& j = b->last_jobs[job - b->last_first];
    const size_t n2 = (size_t)j.N * j.N;
    hipLaunchKernelGGL(batch_copy_factor_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, bstream(stream), j.A, j.Npad, j.N, L_out);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int alabi_gp_batch_get_alpha(alabi_gp_batch* b, int job, double* alpha_out, void* stream) {
    if (!b || !alpha_out || job < b->last_first || job >= b->last_first + b->last_count) return ALABI_BAD_ARGUMENT;
    const BatchJob& j = b->last_jobs[job - b->last_first];
    ALABI_HIP_CHECK(hipMemcpyAsync(alpha_out, j.alpha, (size_t)j.N * sizeof(double), hipMemcpyDeviceToDevice, bstream(stream)));
    return ALABI_OK;
}

}  // extern "C"
