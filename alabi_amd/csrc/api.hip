// extern "C" entry points of libalabi_hip.so (declared in include/alabi_hip.h).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#include "gp_device.hpp"

namespace alabi {
thread_local std::string g_last_error;

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__global__ void __launch_bounds__(256)
copy_factor_kernel(const double* __restrict__ L, int ld, int N, double* __restrict__ out) {
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * N) return;
    int r = (int)(e / N), c = (int)(e % N);
    out[e] = (c <= r) ? L[(size_t)r * ld + c] : 0.0;
}

namespace {
struct CachedBuf { void* p; size_t bytes; };
std::mutex g_cache_mu;
std::vector<CachedBuf> g_cache;        // at most 4 buffers, the smallest is dropped first
}

void* dev_cache_take(size_t need, size_t* bytes) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    int best = -1;
    for (int i = 0; i < (int)g_cache.size(); ++i)
        if (g_cache[i].bytes >= need && (best < 0 || g_cache[i].bytes < g_cache[best].bytes)) best = i;
    if (best < 0) return nullptr;
    void* p = g_cache[best].p;
    *bytes = g_cache[best].bytes;
    g_cache.erase(g_cache.begin() + best);
    return p;
}

void dev_cache_give(void* p, size_t bytes) {
    if (!p) return;
    void* drop = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_cache.push_back({p, bytes});
        if (g_cache.size() > 4) {
            int sm = 0;
            for (int i = 1; i < (int)g_cache.size(); ++i) if (g_cache[i].bytes < g_cache[sm].bytes) sm = i;
            drop = g_cache[sm].p;
            g_cache.erase(g_cache.begin() + sm);
        }
    }
    if (drop) (void)hipFree(drop);
}

int dev_alloc_cached(void** p, size_t need, size_t* bytes) {
    *p = dev_cache_take(need, bytes);
    if (*p) return (int)hipSuccess;
    const hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; *bytes = 0; return (int)e; }
    *bytes = need;
    return (int)hipSuccess;
}

static int fill_dimvec(DimVec& v, const double* src, int d, int stride, int off, double pad) {
    for (int k = 0; k < ALABI_MAX_DIM; ++k) v.v[k] = (k < d) ? src[k * stride + off] : pad;
    return 0;
}
}  // namespace alabi

using namespace alabi;

extern "C" {

int alabi_abi_version(void) { return 1; }

const char* alabi_status_string(int status) {
    switch (status) {
        case ALABI_OK: return "ok";
        case ALABI_NOT_POSITIVE_DEFINITE: return "matrix is not positive definite";
        case ALABI_BAD_ARGUMENT: return "bad argument";
        case ALABI_HIP_ERROR: return "HIP runtime error";
        case ALABI_NOT_COMPUTED: return "GP not computed / y not set";
        case ALABI_TIMEOUT: return "persistent ensemble kernel timed out waiting for a hand-off";
        default: return "unknown status";
    }
}

const char* alabi_last_error(void) { return g_last_error.c_str(); }

int alabi_device_info(int* n_cu, int* lds_bytes, char* arch) {
    int dev = 0;
    ALABI_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    ALABI_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (arch) { strncpy(arch, prop.gcnArchName, 63); arch[63] = 0; }
    return ALABI_OK;
}

// ---------------------------------------------------------------------------------- GP
int alabi_gp_create(int n_cap, int d, alabi_gp** out) {
    if (!out || n_cap <= 0 || d <= 0 || d > ALABI_MAX_DIM) return ALABI_BAD_ARGUMENT;
    alabi_gp* gp = new (std::nothrow) alabi_gp();
    if (!gp) return ALABI_BAD_ARGUMENT;
    gp->n_cap = round_up(n_cap, ALABI_BLK);
    gp->d = d;
    for (int k = 0; k < ALABI_MAX_DIM; ++k) { gp->log_M[k] = 0.0; gp->inv_len.v[k] = (k < d) ? 1.0 : 0.0; }
    const size_t nc = gp->n_cap;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(&gp->L, nc * nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&gp->Xt, (size_t)dim_bucket(d) * nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&gp->y, nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&gp->alpha, nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&gp->dinv, nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&gp->work, 2 * nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&gp->work2, nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&gp->flags, sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&gp->red, 4 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&gp->info, sizeof(int));
    if (e != hipSuccess) {
        alabi_gp_destroy(gp);
        return hip_fail(e, "hipMalloc(gp buffers)", __FILE__, __LINE__);
    }
    *out = gp;
    return ALABI_OK;
}

int alabi_gp_destroy(alabi_gp* gp) {
    if (!gp) return ALABI_OK;
    if (gp->L) (void)hipFree(gp->L);
    if (gp->Xt) (void)hipFree(gp->Xt);
    if (gp->Xa) (void)hipFree(gp->Xa);
    if (gp->xa_centre) (void)hipFree(gp->xa_centre);
    if (gp->host_status) (void)hipHostFree(gp->host_status);
    if (gp->point_host) (void)hipHostFree(gp->point_host);
    if (gp->point_dev) (void)hipFree(gp->point_dev);
    if (gp->y) (void)hipFree(gp->y);
    if (gp->alpha) (void)hipFree(gp->alpha);
    if (gp->dinv) (void)hipFree(gp->dinv);
    if (gp->work) (void)hipFree(gp->work);
    if (gp->work2) (void)hipFree(gp->work2);
    if (gp->flags) (void)hipFree(gp->flags);
    if (gp->red) (void)hipFree(gp->red);
    if (gp->info) (void)hipFree(gp->info);
    if (gp->chol_ctl) (void)hipFree(gp->chol_ctl);
    if (gp->ws || gp->winv) (void)hipDeviceSynchronize();   // nothing in flight may still use the buffers handed to the cache
    alabi::dev_cache_give(gp->ws, gp->ws_bytes);
    if (gp->scan) (void)hipFree(gp->scan);
    alabi::dev_cache_give(gp->winv, gp->winv_bytes);
    if (gp->small) (void)hipFree(gp->small);
    if (gp->pgrad) (void)hipFree(gp->pgrad);
    if (gp->Xc) (void)hipFree(gp->Xc);
    if (gp->ens_h) (void)hipFree(gp->ens_h);
    if (gp->mupart) (void)hipFree(gp->mupart);
    delete gp;
    return ALABI_OK;
}

int alabi_gp_set_hyper(alabi_gp* gp, double mean, double log_white_noise, double log_amp, const double* log_M) {
    if (!gp || !log_M) return ALABI_BAD_ARGUMENT;
    if (!std::isfinite(mean) || !std::isfinite(log_white_noise) || !std::isfinite(log_amp)) return ALABI_BAD_ARGUMENT;
    for (int k = 0; k < gp->d; ++k)
        if (!std::isfinite(log_M[k])) return ALABI_BAD_ARGUMENT;
    gp->mean = mean; gp->log_wn = log_white_noise; gp->log_amp = log_amp;
    for (int k = 0; k < gp->d; ++k) { gp->log_M[k] = log_M[k]; gp->inv_len.v[k] = std::exp(-0.5 * log_M[k]); }
    gp->computed = false; gp->has_alpha = false; gp->gen++;
    return ALABI_OK;
}

int alabi_gp_set_kernel(alabi_gp* gp, int kernel_type, double log_alpha) {
    if (!gp || kernel_type < 0 || kernel_type > 3 || !std::isfinite(log_alpha)) return ALABI_BAD_ARGUMENT;
    gp->kf.type = kernel_type;
    gp->kf.alpha = std::exp(log_alpha);
    gp->computed = false; gp->has_alpha = false; gp->gen++;
    return ALABI_OK;
}

int alabi_gp_compute(alabi_gp* gp, const double* X, int N, void* stream) {
    if (!gp || !X || N <= 0 || N > gp->n_cap) return ALABI_BAD_ARGUMENT;
    hipStream_t s = as_stream(stream);
    gp->N = N; gp->Npad = round_up(N, ALABI_BLK);
    gp->computed = false; gp->has_alpha = false; gp->gen++;
    int st;
    if ((st = launch_prepare_inputs(gp, X, N, s)) != ALABI_OK) return st;
    // 16..160 block columns: the task-queue factorisation (one launch); a wait that runs out there is remembered for a while
    static std::atomic<int> tasks_penalty{0};
    int queued = 0, ctl_ints = 0;
    const char* tq = getenv("ALABI_CHOL_TASKS");              // "1" forces the queue: then the penalty does not apply either
    if (!(tq && tq[0] == '1') && tasks_penalty.load(std::memory_order_relaxed) > 0) tasks_penalty.fetch_sub(1, std::memory_order_relaxed);
    else if ((st = cholesky_tasks_prepare(gp, s, &ctl_ints)) != ALABI_OK) return st;
    if ((st = launch_assemble(gp, s, ctl_ints)) != ALABI_OK) return st;   // also clears the status word and the queue's control words
    if (ctl_ints > 0 && (st = launch_cholesky_tasks(gp, s, &queued)) != ALABI_OK) return st;
    if (!queued && (st = launch_cholesky(gp, s)) != ALABI_OK) return st;
    // one read-back of (pivot status, queue time-out) into pinned memory, one synchronisation
    if (!gp->host_status) ALABI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&gp->host_status), 2 * sizeof(int), hipHostMallocDefault));
    gp->host_status[1] = 0;
    ALABI_HIP_CHECK(hipMemcpyAsync(&gp->host_status[0], gp->info, sizeof(int), hipMemcpyDeviceToHost, s));
    if (queued) ALABI_HIP_CHECK(hipMemcpyAsync(&gp->host_status[1], gp->chol_ctl + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));
    gp->factor_path = queued ? 2 : 1;
    if (queued && gp->host_status[1]) {                     // undefined matrix state: assemble and factorise again, step by step
        gp->factor_path = 3;
        tasks_penalty.store(64, std::memory_order_relaxed);
        if ((st = launch_assemble(gp, s)) != ALABI_OK) return st;
        if ((st = launch_cholesky(gp, s)) != ALABI_OK) return st;
        ALABI_HIP_CHECK(hipMemcpyAsync(&gp->host_status[0], gp->info, sizeof(int), hipMemcpyDeviceToHost, s));
        ALABI_HIP_CHECK(hipStreamSynchronize(s));
    }
    const int info = gp->host_status[0];
    gp->last_pivot = info;
    if (info != 0) return ALABI_NOT_POSITIVE_DEFINITE;
    gp->computed = true;
    gp->factor_gen++;
    return ALABI_OK;
}

int alabi_gp_fit_predict(alabi_gp* gp, const double* X, int N, const double* y, const double* Xs, long long M, double* mu,
                         double* nll_out, void* stream) {
    if (!gp || !y || !nll_out || M < 0 || (M > 0 && (!Xs || !mu))) return ALABI_BAD_ARGUMENT;
    int st = alabi_gp_compute(gp, X, N, stream);
    if (st != ALABI_OK) return st;
    if ((st = alabi_gp_set_y(gp, y, stream)) != ALABI_OK) return st;
    if ((st = alabi_gp_nll(gp, nll_out, stream)) != ALABI_OK) return st;
    if (M > 0) st = alabi_gp_predict(gp, Xs, M, mu, nullptr, stream);
    return st;
}

int alabi_gp_append(alabi_gp* gp, const double* x_new, void* stream) {
    if (!gp || !x_new) return ALABI_BAD_ARGUMENT;
    if (!gp->computed) return ALABI_NOT_COMPUTED;
    if (gp->N >= gp->Npad || gp->N + 1 > gp->n_cap) return ALABI_BAD_ARGUMENT;      // no padding row left: refit
    hipStream_t s = as_stream(stream);
    int st = ensure_winv(gp, s);
    if (st == ALABI_NOT_COMPUTED) return ALABI_BAD_ARGUMENT;                          // no room for the cached L^-1: refit
    if (st != ALABI_OK) return st;
    if ((st = launch_append(gp, x_new, s)) != ALABI_OK) return st;
    int info = 0;
    ALABI_HIP_CHECK(hipMemcpyAsync(&info, gp->info, sizeof(int), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));
    gp->last_pivot = info;
    if (info != 0) return ALABI_NOT_POSITIVE_DEFINITE;                                // factor and cache untouched
    gp->N += 1;
    gp->has_alpha = false;
    gp->gen++; gp->factor_gen++;
    gp->winv_gen = gp->factor_gen;                                                    // the cache was extended with the factor
    return ALABI_OK;
}

int alabi_gp_set_mean(alabi_gp* gp, double mean) {
    if (!gp || !std::isfinite(mean)) return ALABI_BAD_ARGUMENT;
    if (mean != gp->mean) { gp->mean = mean; gp->has_alpha = false; gp->gen++; }
    return ALABI_OK;
}

int alabi_gp_last_pivot(alabi_gp* gp, int* pivot) {
    if (!gp || !pivot) return ALABI_BAD_ARGUMENT;
    *pivot = gp->last_pivot;
    return ALABI_OK;
}

int alabi_gp_n(alabi_gp* gp, int* n) {
    if (!gp || !n) return ALABI_BAD_ARGUMENT;
    *n = gp->N;
    return ALABI_OK;
}

int alabi_gp_set_y(alabi_gp* gp, const double* y, void* stream) {
    if (!gp || !y) return ALABI_BAD_ARGUMENT;
    if (!gp->computed) return ALABI_NOT_COMPUTED;
    hipStream_t s = as_stream(stream);
    ALABI_HIP_CHECK(hipMemcpyAsync(gp->y, y, (size_t)gp->N * sizeof(double), hipMemcpyDeviceToDevice, s));
    int st = launch_alpha(gp, s);
    if (st != ALABI_OK) return st;
    gp->has_alpha = true; gp->gen++;
    return ALABI_OK;
}

int alabi_gp_predict(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var, void* stream) {
    if (!gp || M < 0 || (M > 0 && (!Xs || !mu))) return ALABI_BAD_ARGUMENT;
    if (!gp->computed || !gp->has_alpha) return ALABI_NOT_COMPUTED;
    if (M == 0) return ALABI_OK;
    hipStream_t s = as_stream(stream);
    if (var) {
        // up to 128 queries (64 beyond Npad = 2048): groups of 16 multiply with the cached L^-1, one workgroup per (block row,
        // group), K* evaluated in place -- no pre-pass, no substitution chain (the path of per-point objective calls:
        // utility.py:1030-1163, core.py:1441, and of small batches); measured cross-over against the tile kernels with the
        // split K* pre-pass: tools/prof_medium_batch.py
        const char* env = getenv("ALABI_PV_SMALL");
        const long long small_max = gp->Npad <= 2048 ? 128 : 64;
        if (M <= small_max && gp->Npad >= 256 && gp->d <= 32 && !(env && env[0] == '0')) {
            (void)want_winv(gp, M);            // count the request
            return launch_predict_var_small(gp, Xs, (int)M, mu, var, s);
        }
        return launch_predict_var(gp, Xs, M, mu, var, s);
    }
    return launch_predict_mean(gp, Xs, M, mu, s);
}

int alabi_gp_predict_grad(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var, double* dmu, double* dvar,
                          void* stream) {
    if (!gp || M < 0 || (M > 0 && (!Xs || !dmu || !dvar))) return ALABI_BAD_ARGUMENT;
    if (!gp->computed || !gp->has_alpha) return ALABI_NOT_COMPUTED;
    if (M == 0) return ALABI_OK;
    return launch_predict_grad(gp, Xs, M, mu, var, dmu, dvar, as_stream(stream));
}

// One point in, its value and gradient out, everything on the host side: the evaluation an optimiser makes ~30 times per
// active-learning iteration (the polish step of find_next_point).  Pinned staging buffers owned by the handle: one small copy in,
// the three kernels, one small copy out, one synchronisation -- no tensor is created on the way.
int alabi_gp_predict_grad_point(alabi_gp* gp, const double* x, double* out, void* stream) {
    if (!gp || !x || !out) return ALABI_BAD_ARGUMENT;
    if (!gp->computed || !gp->has_alpha) return ALABI_NOT_COMPUTED;
    hipStream_t s = as_stream(stream);
    const int d = gp->d, nres = 2 + 2 * d;
    if (!gp->point_host) {
        ALABI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&gp->point_host), (size_t)(3 + 3 * ALABI_MAX_DIM) * sizeof(double), hipHostMallocDefault));
        ALABI_HIP_CHECK(hipMalloc(&gp->point_dev, (size_t)(3 + 3 * ALABI_MAX_DIM) * sizeof(double)));
    }
    // Zero copy: the kernels read the point from and write the 2 + 2 d results to the pinned (device-visible, coherent) staging buffer
    // themselves -- two copy operations fewer on the stream per evaluation (ALABI_POINT_COPY=1: explicit copies, the earlier form).
    static const bool copies = [] { const char* e = getenv("ALABI_POINT_COPY"); return e && e[0] == '1'; }();
    for (int k = 0; k < d; ++k) gp->point_host[k] = x[k];
    int st;
    if (copies) {
        double* xd = gp->point_dev; double* res = gp->point_dev + d;      // res: mu, var, dmu[d], dvar[d]
        ALABI_HIP_CHECK(hipMemcpyAsync(xd, gp->point_host, (size_t)d * sizeof(double), hipMemcpyHostToDevice, s));
        if ((st = launch_predict_grad(gp, xd, 1, res, res + 1, res + 2, res + 2 + d, s)) != ALABI_OK) return st;
        ALABI_HIP_CHECK(hipMemcpyAsync(gp->point_host + d, res, (size_t)nres * sizeof(double), hipMemcpyDeviceToHost, s));
    } else {
        double* res = gp->point_host + d;
        if ((st = launch_predict_grad(gp, gp->point_host, 1, res, res + 1, res + 2, res + 2 + d, s)) != ALABI_OK) return st;
    }
    ALABI_HIP_CHECK(hipStreamSynchronize(s));
    for (int k = 0; k < nres; ++k) out[k] = gp->point_host[d + k];
    return ALABI_OK;
}

static int gp_reductions_to_host(alabi_gp* gp, double out[2], hipStream_t s) {
    int st = launch_reductions(gp, s);
    if (st != ALABI_OK) return st;
    ALABI_HIP_CHECK(hipMemcpyAsync(out, gp->red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));
    return ALABI_OK;
}

int alabi_gp_logdet(alabi_gp* gp, double* out, void* stream) {
    if (!gp || !out) return ALABI_BAD_ARGUMENT;
    if (!gp->computed) return ALABI_NOT_COMPUTED;
    double r[2];
    int st = gp_reductions_to_host(gp, r, as_stream(stream));
    if (st != ALABI_OK) return st;
    *out = r[0];
    return ALABI_OK;
}

int alabi_gp_nll(alabi_gp* gp, double* out, void* stream) {
    if (!gp || !out) return ALABI_BAD_ARGUMENT;
    if (!gp->computed || !gp->has_alpha) return ALABI_NOT_COMPUTED;
    double r[2];
    int st = gp_reductions_to_host(gp, r, as_stream(stream));
    if (st != ALABI_OK) return st;
    *out = 0.5 * r[1] + 0.5 * r[0] + 0.5 * gp->N * std::log(2.0 * 3.141592653589793);
    return ALABI_OK;
}

int alabi_gp_grad_log_likelihood(alabi_gp* gp, double* grad_out, void* stream) {
    if (!gp || !grad_out) return ALABI_BAD_ARGUMENT;
    if (!gp->computed || !gp->has_alpha) return ALABI_NOT_COMPUTED;
    hipStream_t s = as_stream(stream);
    double* grad_dev = gp->work;                       // solve scratch, free once alpha exists (2 n_cap >= d + 4 doubles)
    if (2 * (size_t)gp->n_cap < (size_t)gp->d + 4) return ALABI_BAD_ARGUMENT;
    int st = launch_grad_log_likelihood(gp, grad_dev, s);
    if (st != ALABI_OK) return st;
    ALABI_HIP_CHECK(hipMemcpyAsync(grad_out, grad_dev, (size_t)(gp->d + 4) * sizeof(double), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));
    return ALABI_OK;
}

int alabi_gp_get_alpha(alabi_gp* gp, double* alpha_out, void* stream) {
    if (!gp || !alpha_out) return ALABI_BAD_ARGUMENT;
    if (!gp->computed || !gp->has_alpha) return ALABI_NOT_COMPUTED;
    ALABI_HIP_CHECK(hipMemcpyAsync(alpha_out, gp->alpha, (size_t)gp->N * sizeof(double), hipMemcpyDeviceToDevice,
                                   as_stream(stream)));
    return ALABI_OK;
}

int alabi_gp_get_factor(alabi_gp* gp, double* L_out, void* stream) {
    if (!gp || !L_out) return ALABI_BAD_ARGUMENT;
    if (!gp->computed) return ALABI_NOT_COMPUTED;
    size_t n2 = (size_t)gp->N * gp->N;
    hipLaunchKernelGGL(copy_factor_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       gp->L, gp->Npad, gp->N, L_out);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int alabi_gp_last_factor_path(alabi_gp* gp, int* path) {
    if (!gp || !path) return ALABI_BAD_ARGUMENT;
    *path = gp->factor_path;
    return ALABI_OK;
}

int alabi_gp_get_inverse(alabi_gp* gp, double* Kinv_out, void* stream) {
    if (!gp || !Kinv_out) return ALABI_BAD_ARGUMENT;
    if (!gp->computed) return ALABI_NOT_COMPUTED;
    return launch_get_inverse(gp, Kinv_out, as_stream(stream));
}

int alabi_kernel_matrix(const double* X1, int n1, const double* X2, int n2, int d, int kernel_type, double log_alpha,
                        double log_amp, const double* log_M, double* K_out, void* stream) {
    if (!X1 || !X2 || !K_out || !log_M || n1 <= 0 || n2 <= 0 || d <= 0 || d > ALABI_MAX_DIM || kernel_type < 0 ||
        kernel_type > 3)
        return ALABI_BAD_ARGUMENT;
    const KernelFn kf{kernel_type, std::exp(log_alpha)};
    DimVec inv;
    for (int k = 0; k < ALABI_MAX_DIM; ++k) inv.v[k] = (k < d) ? std::exp(-0.5 * log_M[k]) : 0.0;
    return launch_kernel_matrix(X1, n1, X2, n2, d, std::exp(log_amp), inv, kf, K_out, as_stream(stream));
}

// ----------------------------------------------------------------------------- utility
int alabi_utility_eval(int algo, const double* Xs, long long M, int d, const double* bounds, double y_best,
                       const double* mu, const double* var, double* u, void* stream) {
    if (algo < 0 || algo > 2 || M < 0 || d <= 0 || d > ALABI_MAX_DIM || !bounds) return ALABI_BAD_ARGUMENT;
    if (M == 0) return ALABI_OK;
    if (!Xs || !mu || !var || !u) return ALABI_BAD_ARGUMENT;
    DimVec lo, hi;
    fill_dimvec(lo, bounds, d, 2, 0, 0.0);
    fill_dimvec(hi, bounds, d, 2, 1, 0.0);
    return launch_utility_eval(algo, Xs, M, d, lo, hi, y_best, mu, var, u, as_stream(stream));
}

int alabi_utility_scan(alabi_gp* gp, int algo, const double* Xs, long long M, const double* bounds, double y_best,
                       double* u, double* mu_out, double* var_out, double* best_val, long long* best_idx,
                       void* stream) {
    if (!gp || algo < 0 || algo > 2 || M <= 0 || !Xs || !bounds || !best_val || !best_idx) return ALABI_BAD_ARGUMENT;
    if (!gp->computed || !gp->has_alpha) return ALABI_NOT_COMPUTED;
    hipStream_t s = as_stream(stream);
    const int nblocks = 1024;
    const size_t need = ((size_t)3 * M + 2 * nblocks) * sizeof(double);
    if (need > gp->scan_bytes) {
        if (gp->scan) {
            ALABI_HIP_CHECK(hipStreamSynchronize(s));
            ALABI_HIP_CHECK(hipFree(gp->scan));
            gp->scan = nullptr; gp->scan_bytes = 0;
        }
        ALABI_HIP_CHECK(hipMalloc(&gp->scan, need));
        gp->scan_bytes = need;
    }
    double* pv = gp->scan;
    long long* pi = reinterpret_cast<long long*>(gp->scan + nblocks);
    double* mu = mu_out ? mu_out : gp->scan + 2 * nblocks;
    double* var = var_out ? var_out : gp->scan + 2 * nblocks + M;
    double* uu = u ? u : gp->scan + 2 * nblocks + 2 * M;
    int st;
    if ((st = launch_predict_var(gp, Xs, M, mu, var, s)) != ALABI_OK) return st;
    DimVec lo, hi;
    fill_dimvec(lo, bounds, gp->d, 2, 0, 0.0);
    fill_dimvec(hi, bounds, gp->d, 2, 1, 0.0);
    if ((st = launch_utility_eval(algo, Xs, M, gp->d, lo, hi, y_best, mu, var, uu, s)) != ALABI_OK) return st;
    if ((st = launch_argmin(uu, M, pv, pi, nblocks, s)) != ALABI_OK) return st;
    double hv; long long hi_idx;
    ALABI_HIP_CHECK(hipMemcpyAsync(&hv, pv, sizeof(double), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipMemcpyAsync(&hi_idx, pi, sizeof(long long), hipMemcpyDeviceToHost, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));
    *best_idx = hi_idx;
    *best_val = (hi_idx >= 0) ? hv : NAN;
    return ALABI_OK;
}

// ---------------------------------------------------------------------------- ensemble
static void free_draws(DrawBuffers& b) {
    if (b.order) (void)hipFree(b.order);
    if (b.cw) (void)hipFree(b.cw);
    if (b.zz) (void)hipFree(b.zz);
    if (b.lnfac) (void)hipFree(b.lnfac);
    if (b.lnu) (void)hipFree(b.lnu);
    if (b.partner) (void)hipFree(b.partner);
    if (b.u_z) (void)hipFree(b.u_z);
    if (b.u_acc) (void)hipFree(b.u_acc);
    if (b.packed) (void)hipFree(b.packed);
    if (b.pos_of) (void)hipFree(b.pos_of);
    if (b.link) (void)hipFree(b.link);
    b = DrawBuffers{};
}

static DrawBuffers offset_draws(const DrawBuffers& b, size_t off) {
    DrawBuffers r = b;
    r.order += off; r.cw += off; r.zz += off; r.lnfac += off; r.lnu += off;
    r.partner += off; r.u_z += off; r.u_acc += off; r.packed += 4 * off; r.pos_of += off; r.link += 2 * off;
    return r;
}

int alabi_ens_create(alabi_gp* gp, int W, int d, int n_ensembles, const double* bounds, unsigned long long seed,
                     alabi_ens** out) {
    if (!gp || !out || !bounds || W < 2 || W > 8192 || d != gp->d || n_ensembles < 1 || n_ensembles > 4096)
        return ALABI_BAD_ARGUMENT;
    if ((long long)W * n_ensembles > (1LL << 22)) return ALABI_BAD_ARGUMENT;
    alabi_ens* e = new (std::nothrow) alabi_ens();
    if (!e) return ALABI_BAD_ARGUMENT;
    e->gp = gp; e->W = W; e->d = d; e->E = n_ensembles; e->seed = seed;
    { static std::atomic<long long> next_serial{0}; e->serial = ++next_serial; }
    for (int k = 0; k < ALABI_MAX_DIM; ++k) {
        e->lo[k] = (k < d) ? bounds[2 * k] : 0.0;
        e->hi[k] = (k < d) ? bounds[2 * k + 1] : 0.0;
    }
    // 256 compute lanes (up to 4 point pairs each) cover Npad <= 2048: few, fat waves keep the per-wave overhead of a
    // proposal (broadcast, DPP reduction) small -- measured 2.00 us per half step against 2.11 us with 512 lanes.
    // Larger training sets run on the launch-per-half-step path with 512-lane workgroups (room for several proposals per
    // workgroup in ens_half_multi_kernel).
    e->threads = (gp->Npad / 2 <= 1024) ? 256 : 512;
    if (const char* env = getenv("ALABI_ENS_THREADS")) {
        int v = atoi(env);
        if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) e->threads = v;
    }
    const long long WT = (long long)W * n_ensembles;
    long long cap = (4LL << 20) / WT;
    if (cap > 1024) cap = 1024;
    if (cap < 16) cap = 16;
    e->chunk_cap = (int)cap;
    const size_t n = (size_t)e->chunk_cap * WT;
    DrawBuffers& b = e->draws;
    hipError_t err = hipSuccess;
    if (err == hipSuccess) err = hipMalloc(&b.order, n * sizeof(int));
    if (err == hipSuccess) err = hipMalloc(&b.cw, n * sizeof(int));
    if (err == hipSuccess) err = hipMalloc(&b.zz, n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&b.lnfac, n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&b.lnu, n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&b.partner, n * sizeof(int));
    if (err == hipSuccess) err = hipMalloc(&b.u_z, n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&b.u_acc, n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&b.packed, 4 * n * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMalloc(&b.pos_of, n * sizeof(int));
    if (err == hipSuccess) err = hipMalloc(&b.link, 2 * n * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMalloc(&e->run_state, 4 * sizeof(long long));
    if (err == hipSuccess) err = hipMalloc(&e->consts, 5 * ALABI_MAX_DIM * sizeof(double));
    // persistent dataflow path: one workgroup per list position, all co-resident (at most one per CU)
    {
        int dev = 0, n_cu = 0;
        const char* env = getenv("ALABI_ENS_STREAM");
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            n_ensembles <= n_cu && !(env && env[0] == '0')) {
            // at most one workgroup per CU: G positions-in-flight per ensemble, each workgroup strides over the list
            int G = n_cu / n_ensembles;
            if (G > (W + 1) / 2) G = (W + 1) / 2;
            e->stream_grid = G;
            const size_t hist_words = ((size_t)e->chunk_cap + 1) * WT * (d + 2);   // row = coords, logp, accepted
            if (err == hipSuccess) err = hipMalloc(&e->hist, hist_words * sizeof(unsigned long long));
            if (err == hipSuccess) err = hipMalloc(&e->err, sizeof(int));
            e->stream_ok = (err == hipSuccess) ? 1 : 0;
        }
    }
    if (err != hipSuccess) {
        alabi_ens_destroy(e);
        return hip_fail(err, "hipMalloc(ensemble buffers)", __FILE__, __LINE__);
    }
    *out = e;
    return ALABI_OK;
}

int alabi_ens_destroy(alabi_ens* e) {
    if (!e) return ALABI_OK;
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
    free_draws(e->draws);
    if (e->run_state) (void)hipFree(e->run_state);
    if (e->consts) (void)hipFree(e->consts);
    if (e->hist) (void)hipFree(e->hist);
    if (e->part) (void)hipFree(e->part);
    if (e->cand) (void)hipFree(e->cand);
    if (e->err) (void)hipFree(e->err);
    delete e;
    return ALABI_OK;
}

int alabi_ens_set_normal_prior(alabi_ens* e, const double* mean, const double* std) {
    if (!e || !mean || !std) return ALABI_BAD_ARGUMENT;
    e->has_prior = 0; e->prior_const = 0.0;
    for (int k = 0; k < ALABI_MAX_DIM; ++k) { e->prior_mean[k] = 0.0; e->prior_istd[k] = 0.0; }
    for (int k = 0; k < e->d; ++k) {
        if (std::isfinite(mean[k]) && std::isfinite(std[k]) && std[k] > 0.0) {
            e->prior_mean[k] = mean[k];
            e->prior_istd[k] = 1.0 / std[k];
            e->prior_const += -std::log(std[k]) - 0.9189385332046727;   // -log(std) - log(2 pi) / 2, as norm.logpdf
            e->has_prior = 1;
        }
    }
    e->consts_gen = -1;                                                  // re-upload
    e->settings_gen++;
    if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
    return ALABI_OK;
}

int alabi_ens_set_logp_affine(alabi_ens* e, double scale, double shift) {
    if (!e || !(scale > 0.0) || !std::isfinite(scale) || !std::isfinite(shift)) return ALABI_BAD_ARGUMENT;
    e->lp_scale = scale; e->lp_shift = shift;
    e->settings_gen++;
    if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }   // captured launches carry the old values
    return ALABI_OK;
}

int alabi_ens_set_logp_map(alabi_ens* e, int kind) {
    if (!e || kind < 0 || kind > 2) return ALABI_BAD_ARGUMENT;
    e->ymap = kind;
    e->settings_gen++;
    if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }   // captured launches carry the old value
    return ALABI_OK;
}

int alabi_ens_set_stream(alabi_ens* e, int enabled) {
    if (!e) return ALABI_BAD_ARGUMENT;
    if (enabled && !(e->hist && e->err)) return ALABI_BAD_ARGUMENT;
    e->stream_ok = enabled ? 1 : 0;
    e->settings_gen++;
    return ALABI_OK;
}

int alabi_ens_last_path(alabi_ens* e, int* path) {
    if (!e || !path) return ALABI_BAD_ARGUMENT;
    *path = e->last_path;
    return ALABI_OK;
}

int alabi_ens_group_plan(alabi_ens* e, int* out) {
    if (!e || !out) return ALABI_BAD_ARGUMENT;
    for (int i = 0; i < 8; ++i) out[i] = e->group_plan[i];
    return ALABI_OK;
}

// (inv_len, lo, hi) live in device memory; refreshed whenever the GP's hyper-parameters changed.
static int sync_consts(alabi_ens* e, hipStream_t s) {
    // squared exponential: the half-step kernels read the centred inputs and h = |x - c|^2 / 2 - ln|alpha| (ens_se_prepare; per GP,
    // rebuilt when the inputs, y or the hyper-parameters moved) -- here, i.e. before any stream capture
    if (e->gp->kf.type == 0 && e->gp->has_alpha) { const int st = ens_se_prepare(e->gp, s); if (st != ALABI_OK) return st; }
    if (e->consts_gen == e->gp->gen) return ALABI_OK;
    double host[5 * ALABI_MAX_DIM];
    for (int k = 0; k < ALABI_MAX_DIM; ++k) {
        host[k] = e->gp->inv_len.v[k];
        host[ALABI_MAX_DIM + k] = e->lo[k];
        host[2 * ALABI_MAX_DIM + k] = e->hi[k];
        host[3 * ALABI_MAX_DIM + k] = e->prior_mean[k];
        host[4 * ALABI_MAX_DIM + k] = e->prior_istd[k];
    }
    ALABI_HIP_CHECK(hipMemcpyAsync(e->consts, host, sizeof(host), hipMemcpyHostToDevice, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));  // `host` is a stack buffer
    e->consts_gen = e->gp->gen;
    return ALABI_OK;
}

int alabi_ens_lnprob(alabi_ens* e, const double* coords, double* logp, void* stream) {
    if (!e || !coords || !logp) return ALABI_BAD_ARGUMENT;
    if (!e->gp->computed || !e->gp->has_alpha) return ALABI_NOT_COMPUTED;
    int st = sync_consts(e, as_stream(stream));
    if (st != ALABI_OK) return st;
    return launch_ens_lnprob(e, coords, e->W * e->E, logp, 1, as_stream(stream));
}

int alabi_ens_surrogate(alabi_ens* e, const double* points, int M, double* like, void* stream) {
    if (!e || !points || !like || M < 0) return ALABI_BAD_ARGUMENT;
    if (!e->gp->computed || !e->gp->has_alpha) return ALABI_NOT_COMPUTED;
    if (M == 0) return ALABI_OK;
    int st = sync_consts(e, as_stream(stream));
    if (st != ALABI_OK) return st;
    // the normal-prior rows are part of the fused log-probability, not of the surrogate: evaluate without them
    const int hp = e->has_prior; const double pc = e->prior_const;
    e->has_prior = 0; e->prior_const = 0.0;
    st = launch_ens_lnprob(e, points, M, like, 0, as_stream(stream));
    e->has_prior = hp; e->prior_const = pc;
    return st;
}

static HalfArgs base_args(alabi_ens* e, double* coords, double* logp) {
    HalfArgs h{};
    alabi_gp* gp = e->gp;
    h.coords = coords; h.logp = logp; h.consts = e->consts;
    h.Xt = gp->Xt; h.alpha = gp->alpha; h.Npad = gp->Npad;
    h.Xc = gp->Xc; h.ens_h = gp->ens_h; h.centre = gp->xa_centre;
    // the affine map of the log-probability folds into the amplitude and the mean: c (amp s + m) + e = (c amp) s + (c m + e)
    h.amp = e->lp_scale * std::exp(gp->log_amp); h.mean = std::fma(e->lp_scale, gp->mean, e->lp_shift); h.kf = gp->kf;
    h.W = e->W; h.d = e->d; h.n0 = (e->W + 1) / 2;
    h.thin_by = 1; h.run_state = e->run_state;
    h.has_prior = e->has_prior; h.prior_const = e->prior_const; h.ymap = e->ymap;
    return h;
}

static int set_run_state(alabi_ens* e, long long step0, long long done, hipStream_t s) {
    long long host[2] = {step0, done};
    ALABI_HIP_CHECK(hipMemcpyAsync(e->run_state, host, sizeof(host), hipMemcpyHostToDevice, s));
    ALABI_HIP_CHECK(hipStreamSynchronize(s));  // `host` is a stack buffer
    return ALABI_OK;
}

// enqueue `n` steps that consume the chunk buffers (draw + 2n half steps + advance)
static int enqueue_chunk(alabi_ens* e, HalfArgs h, int n, double a, hipStream_t s) {
    int st;
    if ((st = launch_ens_draw(e, n, a, s)) != ALABI_OK) return st;
    const int n0 = h.n0, n1 = e->W - n0;
    const size_t WT = (size_t)e->W * e->E;
    for (int t = 0; t < n; ++t) {
        h.rec = offset_draws(e->draws, (size_t)t * WT);
        h.local_t = t; h.part_begin = 0;
        h.split = 0;
        if ((st = launch_ens_half_args(e, h, n0, s)) != ALABI_OK) return st;
        h.split = 1;
        if ((st = launch_ens_half_args(e, h, n1, s)) != ALABI_OK) return st;
    }
    return launch_ens_advance(e, n, s);
}

int alabi_ens_run(alabi_ens* e, double* coords, double* logp, long long step0, long long nsteps, int thin_by,
                  double a, double* chain, double* chain_logp, long long* n_accept, void* stream) {
    if (!e || !coords || !logp || nsteps < 0 || thin_by < 1 || !(a > 1.0)) return ALABI_BAD_ARGUMENT;
    if (!e->gp->computed || !e->gp->has_alpha) return ALABI_NOT_COMPUTED;
    if (nsteps == 0) return ALABI_OK;
    hipStream_t s = as_stream(stream);
    int st;
    if ((st = sync_consts(e, s)) != ALABI_OK) return st;
    if ((st = set_run_state(e, step0, 0, s)) != ALABI_OK) return st;
    // Persistent dataflow path: training set pinned in registers (Npad <= 2048: 256 compute lanes x up to 4 point pairs), one
    // workgroup per list position.  It synchronises at the end to read the time-out flag.
    e->last_path = 0;
    // Which persistent kernel: ens_stream_kernel where the training set fits one workgroup's registers (N <= 2048, small d),
    // ens_group_kernel (training set partitioned over groups of workgroups, matrix-core kernel sums) beyond that;
    // ALABI_ENS_GROUP=0 disables the latter, =1 prefers it wherever its blocking is feasible.
    const char* genv = getenv("ALABI_ENS_GROUP");
    const bool group_off = genv && genv[0] == '0', group_pref = genv && genv[0] == '1';
    const bool can_stream = ens_stream_fits(e);
    // ens_stream_kernel takes ceil(W/2 / stream_grid) proposals per workgroup one after the other (2.0 / 3.7 / 7.1 us per half step at
    // 1 / 2 / 4 of them, N = 2000); from four on the group kernel is ahead (5.3 us at W = 2048: 1.9e8 against 1.4e8 samples/s)
    const bool crowded = can_stream && e->stream_grid > 0 && ((e->W + 1) / 2 + e->stream_grid - 1) / e->stream_grid >= 4;
    const bool use_group = e->stream_ok && s != nullptr && !group_off && (group_pref || !can_stream || crowded) && ens_group_fits(e) &&
                           ens_group_buffers(e, s);
    if (e->stream_ok && s != nullptr && (can_stream || use_group)) {
        e->last_path = use_group ? 3 : 1;
        ALABI_HIP_CHECK(hipMemsetAsync(e->err, 0, sizeof(int), s));
        long long remaining = nsteps;
        while (remaining > 0) {
            const int K = (int)(remaining < e->chunk_cap ? remaining : e->chunk_cap);
            if ((st = launch_ens_draw(e, K, a, s)) != ALABI_OK) return st;
            if (use_group) st = launch_ens_group(e, coords, logp, K, thin_by, chain, chain_logp, n_accept, s);
            else st = launch_ens_stream(e, coords, logp, K, thin_by, chain, chain_logp, n_accept, s);
            if (st != ALABI_OK) return st;
            if ((st = launch_ens_advance(e, K, s)) != ALABI_OK) return st;
            remaining -= K;
        }
        int flag = 0;
        ALABI_HIP_CHECK(hipMemcpyAsync(&flag, e->err, sizeof(int), hipMemcpyDeviceToHost, s));
        ALABI_HIP_CHECK(hipStreamSynchronize(s));
        return flag ? ALABI_TIMEOUT : ALABI_OK;
    }
    HalfArgs h = base_args(e, coords, logp);
    h.chain = chain; h.chain_logp = chain_logp; h.n_accept = n_accept; h.thin_by = thin_by;

    const char* env = getenv("ALABI_ENS_GRAPH");
    const bool want_graph = (s != nullptr) && !(env && env[0] == '0');
    int gsteps = e->chunk_cap < 256 ? e->chunk_cap : 256;
    if (const char* gs = getenv("ALABI_ENS_GRAPH_STEPS")) {
        int v = atoi(gs);
        if (v > 0 && v <= e->chunk_cap) gsteps = v;
    }
    long long remaining = nsteps;
    if (want_graph && remaining >= 2LL * gsteps) {
        alabi_ens::GraphKey key{coords, logp, chain, chain_logp, n_accept, thin_by, gsteps, a, e->gp->gen};
        const bool same = e->graph_exec && memcmp(&key, &e->graph_key, sizeof(key)) == 0;
        if (!same) {
            if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            ALABI_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            st = enqueue_chunk(e, h, gsteps, a, s);
            hipError_t ce = hipStreamEndCapture(s, &graph);
            if (st != ALABI_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
            if (ce != hipSuccess) return hip_fail(ce, "hipStreamEndCapture", __FILE__, __LINE__);
            hipError_t ie = hipGraphInstantiate(&e->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) { e->graph_exec = nullptr; return hip_fail(ie, "hipGraphInstantiate", __FILE__, __LINE__); }
            e->graph_key = key;
            e->graph_steps = gsteps;
        }
        while (remaining >= gsteps) {
            ALABI_HIP_CHECK(hipGraphLaunch(e->graph_exec, s));
            remaining -= gsteps;
        }
    }
    while (remaining > 0) {
        const int n = (int)(remaining < e->chunk_cap ? remaining : e->chunk_cap);
        if ((st = enqueue_chunk(e, h, n, a, s)) != ALABI_OK) return st;
        remaining -= n;
    }
    return ALABI_OK;
}

int alabi_ens_draw(alabi_ens* e, long long step0, int nsteps, double a, void* stream) {
    if (!e || nsteps <= 0 || nsteps > e->chunk_cap || !(a > 1.0)) return ALABI_BAD_ARGUMENT;
    hipStream_t s = as_stream(stream);
    int st;
    if ((st = set_run_state(e, step0, 0, s)) != ALABI_OK) return st;
    e->drawn_n = nsteps;
    return launch_ens_draw(e, nsteps, a, s);
}

int alabi_ens_half_step(alabi_ens* e, double* coords, double* logp, int t, int split, int part_begin, int part_end,
                        long long* n_accept, void* stream) {
    if (!e || !coords || !logp || t < 0 || t >= e->drawn_n || (split != 0 && split != 1) || e->E != 1)
        return ALABI_BAD_ARGUMENT;
    if (!e->gp->computed || !e->gp->has_alpha) return ALABI_NOT_COMPUTED;
    int st = sync_consts(e, as_stream(stream));
    if (st != ALABI_OK) return st;
    HalfArgs h = base_args(e, coords, logp);
    const int nS = split == 0 ? h.n0 : e->W - h.n0;
    if (part_begin < 0 || part_end > nS || part_begin > part_end) return ALABI_BAD_ARGUMENT;
    h.rec = offset_draws(e->draws, (size_t)t * e->W);
    h.local_t = t; h.split = split; h.part_begin = part_begin; h.n_accept = n_accept;
    return launch_ens_half_args(e, h, part_end - part_begin, as_stream(stream));
}

}  // extern "C" (reopened below)
namespace alabi {
int ens_sync_consts(alabi_ens* e, hipStream_t s) { return sync_consts(e, s); }
int alabi_ens_half_step_hist(alabi_ens* e, const double* coords, const double* logp, int t, int split, int part_begin, int part_end,
                             const double* shist, double* out, hipStream_t s) {
    if (!e || t < 0 || t >= e->drawn_n || e->E != 1 || !shist || !out) return ALABI_BAD_ARGUMENT;
    int st = sync_consts(e, s);
    if (st != ALABI_OK) return st;
    HalfArgs h = base_args(e, const_cast<double*>(coords), const_cast<double*>(logp));
    h.rec = offset_draws(e->draws, (size_t)t * e->W);
    h.local_t = t; h.split = split; h.part_begin = part_begin;
    h.shist = shist; h.sout = out;
    return launch_ens_half_args(e, h, part_end - part_begin, s);
}
}  // namespace alabi
extern "C" {

int alabi_ens_propose(alabi_ens* e, const double* coords, int t, int split, int gate_box, double* q, double* like,
                      void* stream) {
    if (!e || !coords || !q || t < 0 || t >= e->drawn_n || (split != 0 && split != 1) || e->E != 1) return ALABI_BAD_ARGUMENT;
    if (like && (!e->gp->computed || !e->gp->has_alpha)) return ALABI_NOT_COMPUTED;
    int st = sync_consts(e, as_stream(stream));
    if (st != ALABI_OK) return st;
    HalfArgs h = base_args(e, const_cast<double*>(coords), nullptr);
    const int nS = split == 0 ? h.n0 : e->W - h.n0;
    h.rec = offset_draws(e->draws, (size_t)t * e->W);
    h.local_t = t; h.split = split; h.part_begin = 0;
    return launch_ens_propose(e, h, nS, gate_box, q, like, as_stream(stream));
}

int alabi_ens_accept(alabi_ens* e, double* coords, double* logp, int t, int split, const double* q, const double* lp_new,
                     long long* n_accept, void* stream) {
    if (!e || !coords || !logp || !q || !lp_new || t < 0 || t >= e->drawn_n || (split != 0 && split != 1) || e->E != 1)
        return ALABI_BAD_ARGUMENT;
    HalfArgs h = base_args(e, coords, logp);
    const int nS = split == 0 ? h.n0 : e->W - h.n0;
    h.rec = offset_draws(e->draws, (size_t)t * e->W);
    h.local_t = t; h.split = split; h.part_begin = 0; h.n_accept = n_accept;
    return launch_ens_accept(e, h, nS, q, lp_new, as_stream(stream));
}

int alabi_ens_step_lists(alabi_ens* e, int t, int* order_out, int* n0, void* stream) {
    if (!e || t < 0 || t >= e->drawn_n || !order_out || !n0 || e->E != 1) return ALABI_BAD_ARGUMENT;
    ALABI_HIP_CHECK(hipMemcpyAsync(order_out, e->draws.order + (size_t)t * e->W, (size_t)e->W * sizeof(int),
                                   hipMemcpyDeviceToDevice, as_stream(stream)));
    *n0 = (e->W + 1) / 2;
    return ALABI_OK;
}

int alabi_ens_step_with_randoms(alabi_ens* e, double* coords, double* logp, const int* order, int n0,
                                const double* u_z, const int* partner, const double* u_acc, double a,
                                long long* n_accept, void* stream) {
    if (!e || !coords || !logp || !order || !u_z || !partner || !u_acc || n0 < 0 || n0 > e->W || !(a > 1.0) || e->E != 1)
        return ALABI_BAD_ARGUMENT;
    if (!e->gp->computed || !e->gp->has_alpha) return ALABI_NOT_COMPUTED;
    hipStream_t s = as_stream(stream);
    int st;
    if ((st = sync_consts(e, s)) != ALABI_OK) return st;
    if ((st = launch_ens_prep(e, order, n0, u_z, partner, u_acc, a, s)) != ALABI_OK) return st;
    e->drawn_n = 0;  // row 0 of the draw buffers now holds caller data
    HalfArgs h = base_args(e, coords, logp);
    h.rec = e->draws; h.n0 = n0; h.n_accept = n_accept; h.local_t = 0; h.part_begin = 0;
    h.split = 0;
    if ((st = launch_ens_half_args(e, h, n0, s)) != ALABI_OK) return st;
    h.split = 1;
    return launch_ens_half_args(e, h, e->W - n0, s);
}

int alabi_ens_export_draws(alabi_ens* e, long long step, double a, int* order, int* n0, double* u_z, int* partner,
                           double* u_acc, int* cw, double* zz, void* stream) {
    if (!e || !order || !n0 || !u_z || !partner || !u_acc) return ALABI_BAD_ARGUMENT;
    hipStream_t s = as_stream(stream);
    int st;
    if ((st = alabi_ens_draw(e, step, 1, a, stream)) != ALABI_OK) return st;
    const size_t WT = (size_t)e->W * e->E;
    ALABI_HIP_CHECK(hipMemcpyAsync(order, e->draws.order, WT * sizeof(int), hipMemcpyDeviceToDevice, s));
    ALABI_HIP_CHECK(hipMemcpyAsync(partner, e->draws.partner, WT * sizeof(int), hipMemcpyDeviceToDevice, s));
    ALABI_HIP_CHECK(hipMemcpyAsync(u_z, e->draws.u_z, WT * sizeof(double), hipMemcpyDeviceToDevice, s));
    ALABI_HIP_CHECK(hipMemcpyAsync(u_acc, e->draws.u_acc, WT * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (cw) ALABI_HIP_CHECK(hipMemcpyAsync(cw, e->draws.cw, WT * sizeof(int), hipMemcpyDeviceToDevice, s));
    if (zz) ALABI_HIP_CHECK(hipMemcpyAsync(zz, e->draws.zz, WT * sizeof(double), hipMemcpyDeviceToDevice, s));
    *n0 = (e->W + 1) / 2;
    return ALABI_OK;
}

}  // extern "C"
