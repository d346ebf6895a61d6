// Hand-off latency under load: NP pairs of single-wave workgroups bounce counters through device memory at the same time
// (pair i = workgroups 2i and 2i+1, i.e. neighbouring XCDs; each pair has its own 256-byte-aligned words), optionally with
// ROWS words per message like the ensemble kernel's rows (lane k polls word k; the data is the flag).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/pingpong_many.hip -o tools/micro/pingpong_many
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int WORDS, int SHIFT>
__global__ void pingpong_many(unsigned long long* buf, int rounds, int* err) {
    const int me = blockIdx.x & 1, pair = blockIdx.x >> 1, lane = threadIdx.x;
    unsigned long long* a = buf + (size_t)pair * 128 + SHIFT;   // 1 KB per pair: a at +SHIFT words, b 64 words further (SHIFT = 8: a 12-word row straddles two 128-byte lines)
    unsigned long long* mine = (me == 0 ? a : a + 64) + lane;
    unsigned long long* other = (me == 0 ? a + 64 : a) + lane;
    const bool act = lane < WORDS;
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0 && act) __hip_atomic_store(other, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long spins = 0;
        while (true) {
            unsigned long long v = act ? __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)r;
            if (__all(v >= (unsigned long long)r)) break;
            if (++spins > 20000000) { *err = 1; return; }
        }
        if (me == 1 && act) __hip_atomic_store(other, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int WORDS, int SHIFT>
double run(unsigned long long* buf, int npairs, int rounds, int* err) {
    CK(hipMemset(buf, 0, (size_t)npairs * 1024));
    CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pingpong_many<WORDS, SHIFT>), dim3(2 * npairs), dim3(64), 0, 0, buf, 10, err);
    CK(hipDeviceSynchronize());
    CK(hipMemset(buf, 0, (size_t)npairs * 1024));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((pingpong_many<WORDS, SHIFT>), dim3(2 * npairs), dim3(64), 0, 0, buf, rounds, err);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
    return h ? -1.0 : 1e3 * ms / rounds / 2.0;
}

int main() {
    const int rounds = 20000;
    int* err; CK(hipMalloc(&err, 4));
    unsigned long long* buf; CK(hipMalloc(&buf, 256 * 1024));
    for (int np : {1, 8, 32, 64, 128})
        printf("%3d concurrent pairs: one-way hop %.3f us with 1-word messages, %.3f us with 12-word rows in one 128-byte line, %.3f us with rows "
               "straddling two lines\n", np, run<1, 0>(buf, np, rounds, err), run<12, 0>(buf, np, rounds, err), run<12, 8>(buf, np, rounds, err));
    return 0;
}
