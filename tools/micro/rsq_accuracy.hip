// Accuracy of the hardware estimates v_rsq_f64 / v_rcp_f64 on gfx950 (how many Newton steps pivot_factors needs).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/rsq_accuracy.hip -o tools/micro/rsq_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* c0, double* r3, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double p = x[i];
    double r = __builtin_amdgcn_rsq(p);
    r0[i] = r;
    r = r * fma(-0.5 * p * r, r, 1.5);
    r1[i] = r;
    r = r * fma(-0.5 * p * r, r, 1.5);
    r2[i] = r;
    c0[i] = __builtin_amdgcn_rcp(p);
    const double q0 = __builtin_amdgcn_rsq(p), e = fma(-(p * q0), q0, 1.0);
    r3[i] = fma(q0 * e, fma(0.375, e, 0.5), q0);   // one third-order step
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    unsigned s = 1u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = std::exp(((double)(s >> 8) / 16777216.0 - 0.5) * 60.0); }
    double *x, *r0, *r1, *r2, *c0, *r3;
    hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8); hipMalloc(&c0, n * 8); hipMalloc(&r3, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, r0, r1, r2, c0, r3, n);
    std::vector<double> a(n), b(n), c(n), d(n), f(n);
    hipMemcpy(f.data(), r3, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), c0, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, ec = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / sqrtl((long double)h[i]);
        e0 = fmax(e0, fabs((double)((a[i] - t) / t))); e1 = fmax(e1, fabs((double)((b[i] - t) / t))); e2 = fmax(e2, fabs((double)((c[i] - t) / t))); e3 = fmax(e3, fabs((double)((f[i] - t) / t)));
        ec = fmax(ec, fabs((double)((d[i] - 1.0L / (long double)h[i]) * (long double)h[i])));
    }
    printf("v_rsq_f64 max relative error %.3e; after one Newton step %.3e; after two %.3e; v_rcp_f64 %.3e; one third-order step %.3e\n", e0, e1, e2, ec, e3);
    return 0;
}
