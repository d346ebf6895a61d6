// How long do write-through (sc1) stores hold the wave that issues them?  ONE workgroup of W waves; every wave issues K stores of
// 16 or 8 bytes per lane (1 KB / 512 B per instruction, distinct lines), stamps s_memrealtime after the last issue and again
// after s_waitcnt vmcnt(0).  Prints, per (W, K, bytes): issue time and issue + drain time of wave 0, in us.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/sc1_store_rate.hip -o /tmp/sc1_store_rate && /tmp/sc1_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int K, int BYTES>
__global__ void __launch_bounds__(512) k(double* buf, long long* out, int rowstride) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1u << 30, 0x00020000);
    u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int q = 0; q < K; ++q) {
        // 8 lanes x 16 B = one 128-byte line per row, 8 rows per instruction (the slab store's pattern)
        const unsigned off = (unsigned)(((w * K + q) * 8 + (lane >> 3)) * rowstride + (lane & 7) * 16);
        if (BYTES == 16) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
        else { u32x2 h = {v.x, v.y}; __builtin_amdgcn_raw_buffer_store_b64(h, rs, (unsigned)(((w * K + q) * 4 + (lane >> 4)) * rowstride + (lane & 15) * 8), 0, 16); }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { out[2 * w] = t1 - t0; out[2 * w + 1] = t2 - t0; }
}
template <int K, int BYTES>
void run(int W, double* buf, long long* out) {
    long long h[16];
    double best_i = 1e9, best_d = 1e9, worst_i = 0, worst_d = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL((k<K, BYTES>), dim3(1), dim3(64 * W), 0, 0, buf, out, 16000);
        hipDeviceSynchronize();
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double mi = 0, md = 0;
        for (int w = 0; w < W; ++w) { if (h[2 * w] > mi) mi = h[2 * w]; if (h[2 * w + 1] > md) md = h[2 * w + 1]; }
        if (rep >= 2) { if (mi < best_i) best_i = mi; if (md < best_d) best_d = md; if (mi > worst_i) worst_i = mi; if (md > worst_d) worst_d = md; }
    }
    printf("W=%d waves x K=%2d stores of %2d B per lane (%5.1f KB per wave): slowest wave issue %.2f-%.2f us, issue + drain %.2f-%.2f us\n", W, K, BYTES,
           K * 64 * BYTES / 1024.0, 0.01 * best_i, 0.01 * worst_i, 0.01 * best_d, 0.01 * worst_d);
}
int main() {
    double* buf; long long* out;
    hipMalloc(&buf, 1u << 30); hipMalloc(&out, 256);
    hipMemset(buf, 0, 1u << 30);
    for (int W : {1, 2, 4, 8}) {
        run<1, 16>(W, buf, out); run<2, 16>(W, buf, out); run<4, 16>(W, buf, out); run<8, 16>(W, buf, out); run<16, 16>(W, buf, out);
        run<2, 8>(W, buf, out); run<8, 8>(W, buf, out); run<16, 8>(W, buf, out);
    }
    return 0;
}
