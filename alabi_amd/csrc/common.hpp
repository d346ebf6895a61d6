// Shared declarations for libalabi_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/alabi_hip.h"

#define ALABI_BLK 64  // Cholesky / TRSM block edge; matrices are padded to a multiple of it

namespace alabi {

extern thread_local std::string g_last_error;

inline int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed at %s:%d: %s", what, file, line, hipGetErrorString(e));
    g_last_error = buf;
    return ALABI_HIP_ERROR;
}

#define ALABI_HIP_CHECK(expr)                                                    \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) return ::alabi::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define ALABI_LAUNCH_CHECK() ALABI_HIP_CHECK(hipGetLastError())

// Per-dimension scale passed by value to kernels (kernel-argument space, graph safe).
struct DimVec {
    double v[ALABI_MAX_DIM];
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Radial part of the stationary kernel: k(x,x') = amp * f(r2), r2 = sum_k (x_k - x'_k)^2 / M_k.
//   0 ExpSquaredKernel exp(-r2/2)              1 Matern32Kernel (1 + s) exp(-s), s = sqrt(3 r2)
//   2 Matern52Kernel (1 + s + s^2/3) exp(-s), s = sqrt(5 r2)     3 RationalQuadraticKernel (1 + r2/(2a))^-a
// (george kernel definitions; reference: alabi/core.py:1000-1014)
struct KernelFn {
    int type;
    double alpha;
};

}  // namespace alabi

// ---- handle layouts ------------------------------------------------------------------
struct alabi_gp {
    int n_cap = 0;     // capacity in training points (multiple of ALABI_BLK)
    int d = 0;
    int N = 0;         // current number of training points
    int Npad = 0;      // N rounded up to ALABI_BLK; leading dimension of L and Xt rows
    bool computed = false;
    bool has_alpha = false;
    long long gen = 0;  // bumped by every compute / set_y / set_hyper
    int last_pivot = 0;
    int factor_path = 0;      // alabi_gp_last_factor_path
    double* point_host = nullptr;   // alabi_gp_predict_grad_point: pinned [3 + 3 d] (the point, then mu, var, dmu, dvar)
    double* point_dev = nullptr;    // the same on the device
    // hyper-parameters (host copies)
    double mean = 0.0, log_wn = -12.0, log_amp = 0.0;
    alabi::KernelFn kf{0, 1.0};   // kernel family (+ alpha of the rational quadratic)
    double log_M[ALABI_MAX_DIM];
    alabi::DimVec inv_len;  // exp(-0.5 log_M): coordinates are pre-multiplied by it
    // device buffers
    double* L = nullptr;      // [n_cap, n_cap] row-major, lower triangle holds chol(K)
    double* Xt = nullptr;     // [d, n_cap]  scaled, transposed training inputs (SoA)
    double* Xa = nullptr;     // [round_up(d + 2, 4), n_cap] augmented rows (Xt, -|x|^2 / 2, 1, 0..) for the matrix-core predict-mean kernel
    long long xa_gen = -1;    // factor_gen the rows belong to (built lazily)
    double* xa_centre = nullptr;   // [ALABI_MAX_DIM] mean of the scaled training inputs: Xa rows and the query operands are taken relative to it
    double* y = nullptr;      // [n_cap]
    double* alpha = nullptr;  // [n_cap]
    double* dinv = nullptr;   // [n_cap] 1 / L_ii (every triangular solve multiplies by it)
    double* work = nullptr;   // [2 * n_cap] solve scratch
    double* work2 = nullptr;  // [n_cap] hand-off buffer of the dataflow solve
    int* flags = nullptr;     // [1] time-out flag of the dataflow solve
    double* red = nullptr;    // [4] reductions (logdet, r.alpha)
    int* info = nullptr;      // [1] Cholesky info (0 ok, else 1-based pivot)
    int* host_status = nullptr;   // pinned [2]: read-back of (info, task-queue time-out) after a factorisation
    int* chol_ctl = nullptr;  // task-queue factorisation: [0] queue head, [1] time-out flag, [2 + i * nb + j] tile versions
    size_t chol_ctl_ints = 0;
    double* ws = nullptr;     // predict-variance workspace
    size_t ws_bytes = 0;
    double* scan = nullptr;   // utility-scan scratch (partials)
    size_t scan_bytes = 0;
    double* winv = nullptr;   // L^-1, tile-major, for variance requests of at most 16 queries (built lazily per factor)
    size_t winv_bytes = 0;
    long long factor_gen = 0; // bumped by every successful compute
    long long winv_gen = -1;  // factor_gen the cached L^-1 belongs to
    long long req_gen = -1;   // factor_gen the counter below belongs to
    int var_requests = 0;     // variance requests seen for the current factor (decides when the cache pays)
    double* small = nullptr;  // [Npad / 64][16] partial sums of the small-batch variance kernel
    size_t small_bytes = 0;
    double* mupart = nullptr; // partial mean sums of the K* pre-pass when the training points are split over workgroups
    size_t mupart_bytes = 0;
    double* pgrad = nullptr;  // scratch of the query-gradient path (v, |v|^2 partials, z parts)
    size_t pgrad_bytes = 0;
    // ensemble kernels, squared exponential (ensemble.hip: se_pair_terms): the scaled training inputs relative to their mean
    // (xa_centre) and h_n = |x_n - c|^2 / 2 - ln|alpha_n| with sign(alpha_n) in the lowest mantissa bit, rebuilt whenever `gen` moves
    double* Xc = nullptr;     // [dim_bucket(d), Npad]
    double* ens_h = nullptr;  // [Npad]
    long long ens_h_gen = -1;
};

namespace alabi {
// Per-chunk draw buffers, all [chunk_cap, E, W]; the first five are the proposal records in LIST order
// (per ensemble: set 0 then set 1), the last three keep the raw draws for export / tests.
struct DrawBuffers {
    int* order;      // global walker id at each list position
    int* cw;         // global id of the partner walker drawn for that position
    double* zz;      // stretch factor z
    double* lnfac;   // (d-1) ln z
    double* lnu;     // ln u'
    int* partner;    // raw partner index into the complementary list (by list position)
    double* u_z;     // raw uniforms (by list position)
    double* u_acc;
    unsigned long long* packed;   // 4 words per list position: walker | partner << 32, zz, lnfac, lnu (persistent kernel)
    int* pos_of;     // list position (0..W-1 within the ensemble) of every walker, keyed by global walker id (inverse of `order`)
    unsigned long long* link;     // group kernel: 2 words per list position, where the two rows the proposal reads were produced (ens_link_kernel)
};
}  // namespace alabi

struct alabi_ens {
    alabi_gp* gp = nullptr;
    int W = 0, d = 0, E = 1;
    int threads = 1024;       // workgroup size of the half-step kernel
    double lp_scale = 1.0, lp_shift = 0.0;   // log-probability = lp_scale * GP mean + lp_shift inside the box (y scaler)
    // independent normal priors on selected coordinates (lnprior_normal): mean, 1 / std (0 = none), sum of the constants
    double prior_mean[ALABI_MAX_DIM] = {0}, prior_istd[ALABI_MAX_DIM] = {0};
    double prior_const = 0.0;
    int has_prior = 0;
    int ymap = 0;             // inverse y scaler applied to the GP mean inside the kernels (0 identity, 1 -10^x, 2 10^x)
    unsigned long long seed = 0;
    double lo[ALABI_MAX_DIM], hi[ALABI_MAX_DIM];
    double* consts = nullptr; // device [3][ALABI_MAX_DIM]: inv_len, lo, hi
    long long consts_gen = -1;
    int chunk_cap = 0;        // steps the draw buffers hold
    int drawn_n = 0;
    alabi::DrawBuffers draws{};
    // graph cache for the single-GPU run loop
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey {
        void *coords, *logp, *chain, *chain_logp, *n_accept;
        int thin_by, nsteps;
        double a;
        long long gp_gen;  // generation counter of the GP (re-capture after a refit)
    } graph_key{};
    int graph_steps = 0;
    long long* run_state = nullptr;  // device [4]: [0] first global step of the chunk, [1] steps done before it
    // persistent dataflow path (ens_stream_kernel)
    unsigned long long* hist = nullptr;  // [(chunk_cap+1)][E*W][d+1] version history of every walker
    int* err = nullptr;                  // [1] spin time-out flag
    int stream_grid = 0;                 // workgroups per ensemble of the persistent kernel
    int last_path = 0;                   // 1 persistent kernel (ens_stream_kernel), 3 group kernel (ens_group_kernel), 0 one launch per half step
    int stream_ok = 0;                   // eligible: training set fits the lanes' registers, one workgroup per CU
    // group kernel (ens_group_kernel: training set partitioned over the members of a group, proposals streamed through)
    unsigned long long* part = nullptr;  // [2 chunk_cap][E][NG][G][16 Q] partial kernel sums (allocated on first use)
    size_t part_words = 0;
    unsigned long long* cand = nullptr;  // [2 chunk_cap][E][n0][2 d + 4] candidate rows (proposal, old coordinates, logp, ln factors, prior)
    size_t cand_words = 0;
    long long settings_gen = 0;          // bumped by every setter that changes what captured launches carry (graph keys)
    long long serial = 0;                // unique per handle for the life of the process (a new handle at an old address is not the old one)
    int group_plan[8] = {0};  // blocking of the last group-kernel launch: Q, G, NG, RT, tpm, ltw, KS, LDS bytes (alabi_ens_group_plan)
};

namespace alabi {
// gp_assemble.hip
int launch_prepare_inputs(alabi_gp* gp, const double* X, int N, hipStream_t s);
int launch_assemble(alabi_gp* gp, hipStream_t s, int zero_ctl_ints = 0);   // also clears gp->info and the first zero_ctl_ints words of gp->chol_ctl
int launch_kernel_matrix(const double* X1, int n1, const double* X2, int n2, int d, double amp,
                         const DimVec& inv_len, KernelFn kf, double* K, hipStream_t s);
// gp_cholesky.hip
int launch_cholesky(alabi_gp* gp, hipStream_t s);
int cholesky_tasks_prepare(alabi_gp* gp, hipStream_t s, int* ctl_ints);   // > 0: the task queue will run, its control words (to be zeroed by the assembly)
int launch_cholesky_tasks(alabi_gp* gp, hipStream_t s, int* launched);
int launch_cholesky_steps(double* L, int Npad, int* info, double* dinv, hipStream_t s);   // launch-per-step path on a bare matrix
// batched task queue (gp_batch.hip): many independent matrices in one launch of chol_tasks8_batch_kernel
struct CholTask;
struct CholMat;
struct CholBatchQueue {
    std::vector<int> nbs;                       // key of the cached task list: block columns per matrix, lists, window
    int nlists = 0, window = 0, B = 0, shape_sig = 0;
    CholTask* tasks = nullptr; size_t tasks_cap = 0; int ntasks = 0;
    int* list_off = nullptr;                    // device [nlists + 1]
    CholMat* mats = nullptr; size_t mats_cap = 0;
    int* ctl = nullptr; size_t ctl_ints = 0, ctl_cap = 0;   // [32 q] list heads, [1] time-out flag, from [256] on: per matrix tile versions + slab counters
    double* linv = nullptr; size_t linv_cap = 0;            // per matrix the slab buffers [nb][4][64][16] (gp_cholesky.hip: what the panel solves read of a diagonal tile)
};
int chol_batch_prepare(CholBatchQueue& q, int B, const int* ld, double* const* A, double* const* dinv, int* const* info, hipStream_t s);
int chol_batch_launch(CholBatchQueue& q, hipStream_t s);
void chol_batch_free(CholBatchQueue& q);
// gp_solve.hip
int launch_alpha(alabi_gp* gp, hipStream_t s);
int launch_reductions(alabi_gp* gp, hipStream_t s);
// gp_predict.hip
int launch_predict_mean(alabi_gp* gp, const double* Xs, long long M, double* mu, hipStream_t s);
int ens_se_prepare(alabi_gp* gp, hipStream_t s);     // Xc and ens_h of the current (inputs, alpha) for the squared-exponential half-step kernels
int ensure_xa(alabi_gp* gp, hipStream_t s);           // augmented, centred rows Xa of the current factor (matrix-core predict-mean, group ensemble kernel)
// Process-wide cache of the large scratch buffers (variance workspace, cached L^-1): the reference creates a new GP object
// for every refit (gp_utils.py:233), and a 1-2 GiB hipMalloc per new handle costs tens of milliseconds.
void* dev_cache_take(size_t need, size_t* bytes);   // a cached buffer of at least `need` bytes, or nullptr
void dev_cache_give(void* p, size_t bytes);         // hand a buffer back (may free it or another one)
int dev_alloc_cached(void** p, size_t need, size_t* bytes);   // cache first, then hipMalloc; hipError_t as int
int launch_factor_inverse(alabi_gp* gp, hipStream_t s);
int launch_append(alabi_gp* gp, const double* x_new, hipStream_t s);   // gp_append.hip
int launch_factor_inverse_dnc(alabi_gp* gp, double* R, double* dst, hipStream_t s);   // gp_inverse.hip
int launch_predict_grad(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var, double* dmu, double* dvar,
                        hipStream_t s);                                  // gp_predict_grad.hip
int want_winv(alabi_gp* gp, long long M);            // counts the request; 1 when the cached L^-1 should serve it
int ensure_winv(alabi_gp* gp, hipStream_t s);        // the cached L^-1 of the current factor in gp->winv (ALABI_NOT_COMPUTED: no room)
int launch_factor_inverse_into(alabi_gp* gp, double* dst, hipStream_t s);
int launch_predict_var_small(alabi_gp* gp, const double* Xs, int M, double* mu, double* var, hipStream_t s);
int launch_grad_log_likelihood(alabi_gp* gp, double* grad_dev, hipStream_t s);
int launch_get_inverse(alabi_gp* gp, double* Kinv_out, hipStream_t s);   // gp_grad.hip: K^-1 = W^T W, [N,N] row-major
int launch_predict_var(alabi_gp* gp, const double* Xs, long long M, double* mu, double* var,
                       hipStream_t s);
// utility.hip
int launch_utility_eval(int algo, const double* Xs, long long M, int d, const DimVec& lo,
                        const DimVec& hi, double y_best, const double* mu, const double* var,
                        double* u, hipStream_t s);
int launch_argmin(const double* u, long long M, double* partial_val, long long* partial_idx,
                  int nblocks, hipStream_t s);
// ensemble.hip
struct HalfArgs {
    double* coords;              // [E*W,d] in/out
    double* logp;                // [E*W] in/out
    DrawBuffers rec;             // already offset to the step
    const double* consts;        // device [5][ALABI_MAX_DIM]: inv_len, lo, hi, prior mean, prior 1 / std
    const double* Xt;            // [D,Npad]
    const double* alpha;         // [Npad]
    const double* Xc;            // [D,Npad] the scaled inputs relative to their mean   } squared exponential, half-step kernels
    const double* ens_h;         // [Npad] |x - c|^2 / 2 - ln|alpha|, sign(alpha) in bit 0  } (se_pair_terms); see alabi_gp
    const double* centre;        // [d] mean of the scaled training inputs
    double* chain;               // [nstore,E*W,d] or null
    double* chain_logp;          // [nstore,E*W] or null
    long long* n_accept;         // [E*W] or null
    const long long* run_state;  // [0] chunk's first global step, [1] steps done before the chunk
    int n0, W, d, Npad, split, part_begin, local_t, thin_by;
    int count;                   // proposals in this launch (several per workgroup in ens_half_multi_kernel)
    int has_prior;               // consts rows 3 / 4 hold prior mean and 1 / std
    int ymap;                    // inverse y scaler on the GP mean: 0 identity, 1 -10^x (nlog_scaler), 2 10^x (log_scaler)
    double prior_const;
    double amp, mean;
    KernelFn kf;
    // sharded ensemble (ens_sharded.hip): the walker rows live in a history indexed by (half step, rank, slot); rec.link[2 pos] holds
    // the word offsets (own | partner << 32, -1 = the state in coords / logp) of the two rows a proposal reads, the result row
    // (coords, logp, accepted) of proposal blockIdx.x goes to sout + blockIdx.x (d + 2) and coords / logp are left alone
    const double* shist = nullptr;
    double* sout = nullptr;
};
int launch_ens_draw(alabi_ens* e, int nsteps, double a, hipStream_t s);
int launch_ens_prep(alabi_ens* e, const int* order, int n0, const double* u_z, const int* partner,
                    const double* u_acc, double a, hipStream_t s);
int launch_ens_half_args(alabi_ens* e, const HalfArgs& args, int nblocks, hipStream_t s);
int launch_ens_lnprob(alabi_ens* e, const double* coords, int nwalkers, double* logp, int gate_box, hipStream_t s);
int launch_ens_propose(alabi_ens* e, const HalfArgs& args, int nblocks, int gate_box, double* q, double* like, hipStream_t s);
int launch_ens_accept(alabi_ens* e, const HalfArgs& args, int count, const double* q, const double* lp_new, hipStream_t s);
int launch_ens_advance(alabi_ens* e, long long n, hipStream_t s);
// half step of drawn local step t on list slice [begin, end) of `split` in history mode (ens_sharded.hip): rows read from shist / the
// start state, new rows written to `out` (one row of d + 2 doubles per proposal); coords / logp are not modified
int ens_sync_consts(alabi_ens* e, hipStream_t s);   // (inv_len, bounds, prior) on the device match the GP's hyper-parameters
int alabi_ens_half_step_hist(alabi_ens* e, const double* coords, const double* logp, int t, int split, int part_begin, int part_end,
                             const double* shist, double* out, hipStream_t s);
bool ens_stream_fits(const alabi_ens* e);
int launch_ens_stream(alabi_ens* e, double* coords, double* logp, int K, int thin_by, double* chain, double* chain_logp,
                      long long* n_accept, hipStream_t s);
// ens_group.hip
bool ens_group_fits(const alabi_ens* e);
bool ens_group_buffers(alabi_ens* e, hipStream_t s);   // the group kernel's hand-off buffers exist (allocates on first use); false: take another path
int launch_ens_group(alabi_ens* e, double* coords, double* logp, int K, int thin_by, double* chain, double* chain_logp,
                     long long* n_accept, hipStream_t s);
int launch_ens_hist_prologue(alabi_ens* e, double* coords, double* logp, int K, bool fill, hipStream_t s);
int launch_ens_hist_epilogue(alabi_ens* e, double* coords, double* logp, int K, int thin_by, double* chain, double* chain_logp,
                             long long* n_accept, hipStream_t s);
}  // namespace alabi
