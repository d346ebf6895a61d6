// Shared declarations for libalabi_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

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
    // hyper-parameters (host copies)
    double mean = 0.0, log_wn = -12.0, log_amp = 0.0;
    double log_M[ALABI_MAX_DIM];
    alabi::DimVec inv_len;  // exp(-0.5 log_M): coordinates are pre-multiplied by it
    // device buffers
    double* L = nullptr;      // [n_cap, n_cap] row-major, lower triangle holds chol(K)
    double* Xt = nullptr;     // [d, n_cap]  scaled, transposed training inputs (SoA)
    double* y = nullptr;      // [n_cap]
    double* alpha = nullptr;  // [n_cap]
    double* work = nullptr;   // [2 * n_cap] solve scratch
    double* red = nullptr;    // [4] reductions (logdet, r.alpha)
    int* info = nullptr;      // [1] Cholesky info (0 ok, else 1-based pivot)
    double* ws = nullptr;     // predict-variance workspace
    size_t ws_bytes = 0;
    double* scan = nullptr;   // utility-scan scratch (partials)
    size_t scan_bytes = 0;
};

struct alabi_ens {
    alabi_gp* gp = nullptr;
    int W = 0, d = 0;
    unsigned long long seed = 0;
    alabi::DimVec lo, hi;
    // drawn randoms for a chunk of steps
    int chunk_cap = 0;        // steps the buffers hold
    long long drawn_step0 = -1;
    int drawn_n = 0;
    int* order = nullptr;     // [chunk_cap, W]
    int* partner = nullptr;   // [chunk_cap, W]
    double* u_z = nullptr;    // [chunk_cap, W]
    double* u_acc = nullptr;  // [chunk_cap, W]
    // graph cache for the single-GPU run loop
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey {
        void *coords, *logp, *chain, *chain_logp, *n_accept;
        int thin_by, nsteps;
        double a;
        long long gp_gen;  // generation counter of the GP (re-capture after a refit)
    } graph_key{};
    int graph_steps = 0;
    long long* run_state = nullptr;  // device [4]: step_base, stored_base
};

namespace alabi {
// gp_assemble.hip
int launch_prepare_inputs(alabi_gp* gp, const double* X, int N, hipStream_t s);
int launch_assemble(alabi_gp* gp, hipStream_t s);
int launch_kernel_matrix(const double* X1, int n1, const double* X2, int n2, int d, double amp,
                         const DimVec& inv_len, double* K, hipStream_t s);
// gp_cholesky.hip
int launch_cholesky(alabi_gp* gp, hipStream_t s);
// gp_solve.hip
int launch_alpha(alabi_gp* gp, hipStream_t s);
int launch_reductions(alabi_gp* gp, hipStream_t s);
// gp_predict.hip
int launch_predict_mean(alabi_gp* gp, const double* Xs, long long M, double* mu, hipStream_t s);
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
    double* coords;              // [W,d] in/out
    double* logp;                // [W] in/out
    const int* order;            // [W] set 0 then set 1
    const double* u_z;           // [W] by walker id
    const int* partner;          // [W] by walker id
    const double* u_acc;         // [W] by walker id
    const double* Xt;            // [D,Npad]
    const double* alpha;         // [Npad]
    double* chain;               // [nstore,W,d] or null
    double* chain_logp;          // [nstore,W] or null
    long long* n_accept;         // [W] or null
    const long long* run_state;  // [0] chunk's first global step, [1] steps done before the chunk
    int n0, W, d, Npad, split, part_begin, local_t, thin_by;
    double a, amp, mean;
};
int launch_ens_draw(alabi_ens* e, int nsteps, hipStream_t s);
int launch_ens_half_args(alabi_ens* e, const HalfArgs& args, int nblocks, hipStream_t s);
int launch_ens_lnprob(alabi_ens* e, const double* coords, int W, double* logp, hipStream_t s);
int launch_ens_advance(alabi_ens* e, long long n, hipStream_t s);
}  // namespace alabi
