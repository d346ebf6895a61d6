// Accuracy of the library's exp variants (gp_device.hpp) against long-double exp on the host.
// Build: hipcc -O3 --offload-arch=gfx950 -Ialabi_amd/csrc -Iinclude tools/micro/exp_accuracy.hip -o tools/micro/exp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "gp_device.hpp"
using namespace alabi;
__global__ void k(const double* x, double* a, double* b, double* c, int n) {
    __shared__ double t32[32], t64[64];
    if (threadIdx.x < 32) t32[threadIdx.x] = exp2((double)threadIdx.x / 32.0);
    if (threadIdx.x < 64) t64[threadIdx.x] = exp2((double)threadIdx.x / 64.0);
    __syncthreads();
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    a[i] = exp_direct(x[i]);
    b[i] = exp_tab32(x[i], t32);
    c[i] = exp2s_tab64(x[i] * ALABI_EXP2S_SCALE, t64);   // the product is rounded here; the kernel gets it from the MFMA
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    unsigned s = 1u;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; const double u = (double)(s >> 8) / 16777216.0;
        h[i] = (i & 3) == 0 ? -u * 2.0 : (i & 3) == 1 ? -u * 40.0 : (i & 3) == 2 ? -u * 700.0 : -u * 1.0e7; }
    h[0] = 0.0; h[1] = -1e-300; h[2] = -745.0; h[4] = 1e-9;
    h[3] = -1e30;   // exp_direct / exp_tab32 hold up to |x| ~ 1e40 (beyond: the reduction's residual overflows the polynomial)
    double *x, *a, *b, *c;
    hipMalloc(&x, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, a, b, c, n);
    std::vector<double> ra(n), rb(n), rc(n);
    hipMemcpy(ra.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(rb.data(), b, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(rc.data(), c, n * 8, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0, ec = 0; int bad = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = expl((long double)h[i]);
        if (t < 1e-300L) { if (ra[i] > 1e-299 || rb[i] > 1e-299 || rc[i] > 1e-299 || !(rc[i] >= 0.0)) { ++bad; printf("tail: x=%.17g -> %.3g %.3g %.3g\n", h[i], ra[i], rb[i], rc[i]); } continue; }
        // exp2s sees x * scale rounded once: its argument error |x| 2^-53 is part of what the kernel has too
        const double ulp = (double)t * 0x1p-52;
        ea = fmax(ea, fabs((double)(ra[i] - t)) / ulp); eb = fmax(eb, fabs((double)(rb[i] - t)) / ulp);
        if (h[i] > -2.0) ec = fmax(ec, fabs((double)(rc[i] - t)) / ulp);
        else if (fabs((double)(rc[i] - t)) > (2.0 + fabs(h[i])) * ulp) { ++bad; printf("x=%.17g: %.17g, exact %.17g\n", h[i], rc[i], (double)t); }
    }
    printf("max error in ulp: exp_direct %.2f, exp_tab32 %.2f, exp2s_tab64 (|x| < 2) %.2f; out-of-bound results %d\n", ea, eb, ec, bad);
    printf("exp2s_tab64 at 0, -1e-300, -745, -1e30, 1e-9: %.17g %.17g %.3g %.3g %.17g\n", rc[0], rc[1], rc[2], rc[3], rc[4]);
    return bad != 0;
}
