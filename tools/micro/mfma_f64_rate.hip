// fp64 MFMA issue rate: ticks (s_memtime) and wall time per v_mfma_f64_16x16x4_f64 per SIMD, one wave per SIMD, four
// independent accumulators; also with every CU busy (sustained clock under full matrix load).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_rate.hip -o tools/micro/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) rate(double* out, long long* ticks, int iters) {
    v4f64 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = v4f64{0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

int main() {
    double* out; long long* ticks;
    CK(hipMalloc(&out, 4096 * 256 * 8)); CK(hipMalloc(&ticks, 8));
    const int iters = 20000;
    for (int nacc : {2, 4, 8})
    for (int grid : {256, 512}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&](int it) {
            if (nacc == 2) hipLaunchKernelGGL(rate<2>, dim3(grid), dim3(256), 0, 0, out, ticks, it);
            else if (nacc == 4) hipLaunchKernelGGL(rate<4>, dim3(grid), dim3(256), 0, 0, out, ticks, it);
            else hipLaunchKernelGGL(rate<8>, dim3(grid), dim3(256), 0, 0, out, ticks, it);
        };
        launch(100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch(iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
        const double n = (double)iters * 16.0;
        const double waves_per_simd = grid <= 256 ? 1.0 : grid / 256.0;
        printf("%d accumulators, grid %4d: %.1f ticks per MFMA (wave 0), %.2f ns per MFMA per SIMD, chip rate %.1f TFLOP/s\n", nacc, grid, h / n,
               1e6 * ms / (n * waves_per_simd), grid * 4.0 * n * 2048.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
