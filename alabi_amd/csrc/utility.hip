// Acquisition-function epilogue and arg-min reduction for gfx950.
//
// Replaces the per-point Python functions utility.bape_utility (alabi/utility.py:729-810),
// utility.agp_utility (:629-701) and utility.jones_utility (:853-946), their bounds gate
// lnprior_uniform (:218-275) and logsubexp (:489-504), evaluated over a candidate batch, and
// the arg-min that utility.minimize_objective takes over its valid restarts (:1149-1163).
// Elementwise on (mu, var): HBM traffic 8(d+3) bytes per candidate, negligible next to the
// predict that produced (mu, var).
#include "common.hpp"

namespace alabi {

__device__ inline double utility_value(int algo, double mu, double var, double y_best) {
    if (algo == ALABI_UTILITY_BAPE) {
        // -((2 mu + var) + logsubexp(var, 0)); logsubexp = -inf when var <= 0  -> +inf
        // The reference evaluates log(1 - exp(-var)) literally (utility.py:504), which cancels for small
        // var.  To land on the same side of that cancellation, exp(-var) is formed as RN(1 + expm1(-var))
        // for small var: that is the correctly rounded exponential whenever |expm1| << 1 (exactly 1.0 below
        // 2^-54, 1 - 2^-53 just above it); the absolute error of the utility is then eps / (1 - e^-var).
        double em = (var < 0.0009765625) ? 1.0 + expm1(0.0 - var) : exp(0.0 - var);
        double lse = (var <= 0.0) ? -INFINITY : var + log(1.0 - em);
        return -((2.0 * mu + var) + lse);
    } else if (algo == ALABI_UTILITY_AGP) {
        // -(mu + 0.5 log(2 pi e var)); log of a negative variance is NaN, of zero -inf
        const double two_pi_e = 2.0 * 3.141592653589793 * 2.718281828459045;
        return -(mu + 0.5 * log(two_pi_e * var));
    } else {
        const double zeta = 0.01;
        double sd = sqrt(var);
        if (!(sd > 0.0)) return 0.0;  // reference: `if std > 0 ... else return 0.0` (NaN std -> 0.0)
        double imp = mu - y_best - zeta;
        double z = imp / sd;
        double cdf = 0.5 * erfc(-z * 0.7071067811865476);
        double pdf = exp(-0.5 * z * z) * 0.3989422804014327;
        return -(imp * cdf + sd * pdf);
    }
}

__global__ void __launch_bounds__(256)
utility_eval_kernel(int algo, const double* __restrict__ Xs, long long M, int d, DimVec lo, DimVec hi,
                    double y_best, const double* __restrict__ mu, const double* __restrict__ var,
                    double* __restrict__ u) {
    long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    bool inb = true;
    for (int k = 0; k < d; ++k) {
        double x = Xs[m * d + k];
        inb = inb && (x > lo.v[k]) && (x < hi.v[k]);
    }
    u[m] = inb ? utility_value(algo, mu[m], var[m], y_best) : INFINITY;
}

// arg-min over FINITE values (the reference drops non-finite restarts); ties -> lowest index.
__device__ inline void argmin_combine(double& v, long long& i, double v2, long long i2) {
    if (i2 >= 0 && (i < 0 || v2 < v || (v2 == v && i2 < i))) { v = v2; i = i2; }
}

__global__ void __launch_bounds__(256)
argmin_partial_kernel(const double* __restrict__ u, long long M, double* __restrict__ pv,
                      long long* __restrict__ pi) {
    __shared__ double sv[256];
    __shared__ long long si[256];
    double bv = 0.0; long long bi = -1;
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
        double x = u[m];
        if (isfinite(x)) argmin_combine(bv, bi, x, m);
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            double v = sv[threadIdx.x]; long long i = si[threadIdx.x];
            argmin_combine(v, i, sv[threadIdx.x + w], si[threadIdx.x + w]);
            sv[threadIdx.x] = v; si[threadIdx.x] = i;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { pv[blockIdx.x] = sv[0]; pi[blockIdx.x] = si[0]; }
}

__global__ void __launch_bounds__(256)
argmin_final_kernel(double* __restrict__ pv, long long* __restrict__ pi, int n) {
    __shared__ double sv[256];
    __shared__ long long si[256];
    double bv = 0.0; long long bi = -1;
    for (int k = threadIdx.x; k < n; k += 256) argmin_combine(bv, bi, pv[k], pi[k]);
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            double v = sv[threadIdx.x]; long long i = si[threadIdx.x];
            argmin_combine(v, i, sv[threadIdx.x + w], si[threadIdx.x + w]);
            sv[threadIdx.x] = v; si[threadIdx.x] = i;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { pv[0] = sv[0]; pi[0] = si[0]; }
}

int launch_utility_eval(int algo, const double* Xs, long long M, int d, const DimVec& lo, const DimVec& hi,
                        double y_best, const double* mu, const double* var, double* u, hipStream_t s) {
    if (M <= 0) return ALABI_OK;
    long long blocks = (M + 255) / 256;
    if (blocks > 0x7fffffffLL) return ALABI_BAD_ARGUMENT;
    hipLaunchKernelGGL(utility_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, s, algo, Xs, M, d, lo, hi,
                       y_best, mu, var, u);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

int launch_argmin(const double* u, long long M, double* pv, long long* pi, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(argmin_partial_kernel, dim3(nblocks), dim3(256), 0, s, u, M, pv, pi);
    hipLaunchKernelGGL(argmin_final_kernel, dim3(1), dim3(256), 0, s, pv, pi, nblocks);
    ALABI_LAUNCH_CHECK();
    return ALABI_OK;
}

}  // namespace alabi
