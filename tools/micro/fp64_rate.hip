// fp64 VALU issue rate on one SIMD: cycles per wave64 instruction for v_fma_f64 / v_add_f64 / v_mul_f64,
// with 1, 2 and 4 waves per SIMD and 8 independent chains per wave.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/fp64_rate.hip -o tools/micro/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void rate(double* out, long long* ticks, int iters, double c) {
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = fma(a[i], c, c);
                else if (OP == 1) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                else asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
            }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

int main() {
    double* out; long long* ticks;
    CK(hipMalloc(&out, 1024 * 8)); CK(hipMalloc(&ticks, 8));
    const int iters = 2000;
    const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64"};
    for (int op = 0; op < 3; ++op)
        for (int threads = 256; threads <= 1024; threads *= 2) {   // 4 SIMDs per CU: 1, 2, 4 waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                if (op == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(threads), 0, 0, out, ticks, iters, 0.999);
                if (op == 1) hipLaunchKernelGGL(rate<1>, dim3(1), dim3(threads), 0, 0, out, ticks, iters, 0.999);
                if (op == 2) hipLaunchKernelGGL(rate<2>, dim3(1), dim3(threads), 0, 0, out, ticks, iters, 0.999);
                CK(hipDeviceSynchronize());
            }
            long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
            const double n = (double)iters * 64.0;
            printf("%s  waves/SIMD %d: %.2f ticks per instruction per wave, %.2f ticks per instruction per SIMD\n", names[op],
                   threads / 256, h / n, h / n / (threads / 256));
        }
    return 0;
}
