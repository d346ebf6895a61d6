// Hand-off latency microbenchmark: two single-wave workgroups bounce a counter through device memory.
// Variants: load/store scope (agent = sc1, system = sc0 sc1), memory type (hipMalloc, fine-grained, uncached),
// placement (same XCD / different XCD; workgroups are dealt to XCDs round-robin by linear id).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/pingpong.hip -o /tmp/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SCOPE, int MODE>
__global__ void pingpong(unsigned long long* a, unsigned long long* b, int wgA, int wgB, int rounds, int* err) {
    const int me = (blockIdx.x == wgA) ? 0 : (blockIdx.x == wgB ? 1 : -1);
    if (me < 0 || threadIdx.x != 0) return;
    unsigned long long* mine = me == 0 ? a : b;     // I wait on `mine`, I write `other`
    unsigned long long* other = me == 0 ? b : a;
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            if (MODE == 0 || MODE >= 3) __hip_atomic_store(other, (unsigned long long)r, __ATOMIC_RELAXED, SCOPE);
            else __hip_atomic_exchange(other, (unsigned long long)r, __ATOMIC_RELAXED, SCOPE);
        }
        long long spins = 0;
        while (true) {
            unsigned long long v;
            if (MODE == 3) {          // poll through the SCALAR memory path (s_load ... glc: the scalar cache is bypassed)
                asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(mine) : "memory");
            } else if (MODE == 4) {   // the same after invalidating the scalar cache
                asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(mine) : "memory");
            } else {
                v = (MODE == 2) ? __hip_atomic_fetch_add(mine, 0ull, __ATOMIC_RELAXED, SCOPE) : __hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE);
            }
            if (v >= (unsigned long long)r) break;
            if (++spins > (MODE >= 3 ? 2000000 : 20000000)) { *err = 1; return; }
        }
        if (me == 1) {
            if (MODE == 0 || MODE >= 3) __hip_atomic_store(other, (unsigned long long)r, __ATOMIC_RELAXED, SCOPE);
            else __hip_atomic_exchange(other, (unsigned long long)r, __ATOMIC_RELAXED, SCOPE);
        }
    }
}

template <int SCOPE, int MODE>
double run(unsigned long long* buf, int wgA, int wgB, int rounds, int* err) {
    CK(hipMemset(buf, 0, 4096));
    CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = (wgA > wgB ? wgA : wgB) + 1;
    hipLaunchKernelGGL((pingpong<SCOPE, MODE>), dim3(grid), dim3(64), 0, 0, buf, buf + 256, wgA, wgB, 10, err);
    CK(hipDeviceSynchronize());
    CK(hipMemset(buf, 0, 4096));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((pingpong<SCOPE, MODE>), dim3(grid), dim3(64), 0, 0, buf, buf + 256, wgA, wgB, rounds, err);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
    if (h) return -1.0;
    return 1e3 * ms / rounds / 2.0;    // one-way hop in us
}

int main() {
    const int rounds = 20000;
    int* err; CK(hipMalloc(&err, 4));
    unsigned long long *coarse, *fine = nullptr, *unc = nullptr;
    CK(hipMalloc(&coarse, 4096));
    if (hipExtMallocWithFlags((void**)&fine, 4096, hipDeviceMallocFinegrained) != hipSuccess) fine = nullptr;
    if (hipExtMallocWithFlags((void**)&unc, 4096, hipDeviceMallocUncached) != hipSuccess) unc = nullptr;
    struct { const char* name; unsigned long long* p; } mems[] = {{"hipMalloc", coarse}, {"finegrained", fine}, {"uncached", unc}};
    struct { const char* name; int a, b; } places[] = {{"same XCD (wg 0, 8)", 0, 8}, {"other XCD (wg 0, 1)", 0, 1}, {"other XCD (wg 0, 4)", 0, 4}};
    for (auto& m : mems) {
        if (!m.p) { printf("%s: allocation not supported\n", m.name); continue; }
        for (auto& pl : places) {
            printf("%-12s %-22s one-way hop us: agent ld/st %.3f | system ld/st %.3f | agent xchg+ld %.3f | agent xchg+rmw-poll %.3f | "
                   "scalar-load poll (glc) %.3f | with s_dcache_inv %.3f   (-1: never seen)\n",
                   m.name, pl.name,
                   run<__HIP_MEMORY_SCOPE_AGENT, 0>(m.p, pl.a, pl.b, rounds, err),
                   run<__HIP_MEMORY_SCOPE_SYSTEM, 0>(m.p, pl.a, pl.b, rounds, err),
                   run<__HIP_MEMORY_SCOPE_AGENT, 1>(m.p, pl.a, pl.b, rounds, err),
                   run<__HIP_MEMORY_SCOPE_AGENT, 2>(m.p, pl.a, pl.b, rounds, err),
                   run<__HIP_MEMORY_SCOPE_AGENT, 3>(m.p, pl.a, pl.b, rounds, err),
                   run<__HIP_MEMORY_SCOPE_AGENT, 4>(m.p, pl.a, pl.b, rounds, err));
            fflush(stdout);
        }
    }
    return 0;
}
