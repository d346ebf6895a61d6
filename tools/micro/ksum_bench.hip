// Two questions behind ens_stream_kernel's kernel-sum phase, answered with WALL-CLOCK time:
//  (1) fp64 VALU throughput of one SIMD with 1 / 2 / 4 waves issuing independent v_fma_f64 chains, every CU busy
//      (hipEvent time, not s_memtime ticks: the round-1 figure "1.38 ticks per instruction per SIMD at 4 waves" would be
//      3x the 16 lanes/clk/SIMD the 78.6 TFLOP/s peak implies);
//  (2) the time of one "proposal" of the persistent ensemble kernel's compute phase -- q broadcast through LDS, barrier,
//      kernel sum over 1024 point pairs held in registers, DPP reduction, barrier -- for 4 / 8 / 12 compute waves per
//      workgroup (1 / 2 / 3 per SIMD), the difference form r2 = sum (x - q)^2 against the norm form
//      r2 = |x|^2 + |q|^2 - 2 q.x, and q kept in SGPRs (readfirstlane) against VGPRs.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/ksum_bench.hip -o tools/micro/ksum_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../alabi_amd/csrc/gp_device.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using namespace alabi;
typedef double f64x2 __attribute__((ext_vector_type(2)));

// exp(x) for x <= 0 given directly (no -0.5 multiply): Cody-Waite + degree-13 Horner, 19 instructions
__device__ inline double exp_neg(double x) {
    const double n = rint(x * 1.4426950408889634);
    double r = fma(n, -0x1.62e42fee00000p-1, x);
    r = fma(n, -0x1.a39ef35793c76p-33, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
// exp(2h) for h <= 0: reduce h = n ln2/2 + r, |r| <= ln2/4, degree-10 Taylor of exp(r), then square: 17 instructions
__device__ inline double exp_twice(double h) {
    const double n = rint(h * 2.8853900817779268);
    double r = fma(n, -0x1.62e42fee00000p-2, h);
    r = fma(n, -0x1.a39ef35793c76p-34, r);
    double p = 2.755731922398589e-07;                    // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p * p, (int)n);
}

// exp(x), x <= 0, with a 32-entry table of 2^(j/32) in LDS: x = (32 n + j) ln2/32 + r, |r| <= ln2/64, degree-6 Taylor.
// 13 instructions on the fp64 pipe (the polynomial of the table-free form alone has 13) + 3 integer + 1 LDS read.
__device__ inline double exp_tab32(double x, const double* tab) {
    const double k = rint(x * 46.16624130844683);                 // 32 / ln2
    double r = fma(k, -0x1.62e42fee00000p-6, x);                   // ln2/32 hi
    r = fma(k, -0x1.a39ef35793c76p-38, r);                         // ln2/32 lo
    const int ki = (int)k;
    const double t = tab[ki & 31];
    double p = 0.001388888888888889;                               // 1/6!
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p * t, ki >> 5);
}

__global__ void fma_rate(double* out, long long* ticks, int iters, double c) {
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// One workgroup = wave 0 ("hand-off" wave: produces q, consumes the partial sums) + NW compute waves.
// D = 10 coordinates, PPT point pairs per lane; NORM: 0 difference form, 1 norm form; QS: 1 q in SGPRs, 0 q in VGPRs.
template <int NW, int PPT, int NORM, int QS>
__global__ void __launch_bounds__(64 * (NW + 1))
ksum(const double* __restrict__ Xt, const double* __restrict__ alpha, int Npad, int K, double* out, long long* ticks) {
    constexpr int D = 10;
    __shared__ __attribute__((aligned(16))) double qs_s[2][16];
    __shared__ __attribute__((aligned(16))) double scratch[2][16];
    __shared__ double etab[32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 32) etab[tid] = exp2(tid / 32.0);
    const bool comm = wv == 0;
    const int ct = tid - 64, TC = 64 * NW, half = Npad >> 1;
    f64x2 xa[PPT][D], aa[PPT], xx[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int idx = ct + j * TC;
        const bool v = !comm && idx < half;
        xx[j] = f64x2{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < D; ++k) {
            xa[j][k] = v ? reinterpret_cast<const f64x2*>(Xt + (size_t)k * Npad)[idx] : f64x2{0.0, 0.0};
            xx[j].x = fma(xa[j][k].x, xa[j][k].x, xx[j].x);
            xx[j].y = fma(xa[j][k].y, xa[j][k].y, xx[j].y);
        }
        aa[j] = v ? reinterpret_cast<const f64x2*>(alpha)[idx] : f64x2{0.0, 0.0};
        if (NORM) {                                      // -|x|^2 / 2 (form 1) or -|x|^2 / 4 (form 2), opaque to the compiler
            const double c = NORM == 1 ? -0.5 : -0.25;
            xx[j].x *= c; xx[j].y *= c;
            asm volatile("" : "+v"(xx[j].x), "+v"(xx[j].y));
        }
    }
    if (tid < 32) scratch[tid >> 4][tid & 15] = 0.0;
    __syncthreads();
    double total = 0.0, qv = 0.01 * lane + 0.001 * blockIdx.x;
    const long long t0 = clock64();
    for (int it = 0; it < K; ++it) {
        const int par = it & 1;
        if (comm) {
            qv = qv * 0.999 + 1e-3;                      // a "new proposal" every iteration (depends on the last result)
            if (lane < D) qs_s[par][lane] = qv;
            if (NORM) {
                double n2 = 0.0;
                for (int k = 0; k < D; ++k) { const double t = lane_bcast(qv, k); n2 = fma(t, t, n2); }
                if (lane == 15) qs_s[par][15] = (NORM == 1 ? -0.5 : -0.25) * n2;
            }
        }
        __syncthreads();
        if (!comm) {
            double q[D], qq = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double r = qs_s[par][k];
                q[k] = QS ? __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(r)),
                                             __builtin_amdgcn_readfirstlane(__double2loint(r))) : r;
            }
            if (NORM) {
                const double r = qs_s[par][15];
                qq = QS ? __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(r)),
                                           __builtin_amdgcn_readfirstlane(__double2loint(r))) : r;
            }
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                double r2a, r2b;
                if (NORM) {                              // -r2/2 (or -r2/4 with coordinates pre-scaled by 1/sqrt 2) directly
                    r2a = xx[j].x + qq; r2b = xx[j].y + qq;
#pragma unroll
                    for (int k = 0; k < D; ++k) { r2a = fma(xa[j][k].x, q[k], r2a); r2b = fma(xa[j][k].y, q[k], r2b); }
                    acc = fma(aa[j].x, NORM == 1 ? exp_neg(r2a) : exp_twice(r2a), acc);
                    acc = fma(aa[j].y, NORM == 1 ? exp_neg(r2b) : exp_twice(r2b), acc);
                    if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                    continue;
                } else {
                    r2a = 0.0; r2b = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double da = xa[j][k].x - q[k], db = xa[j][k].y - q[k];
                        r2a = fma(da, da, r2a); r2b = fma(db, db, r2b);
                    }
                }
                if (NORM == 0 && QS == 2) {                  // difference form, table exp
                    acc = fma(aa[j].x, exp_tab32(-0.5 * r2a, etab), acc);
                    acc = fma(aa[j].y, exp_tab32(-0.5 * r2b, etab), acc);
                } else {
                    acc = fma(aa[j].x, exp_neg_half(r2a), acc);
                    acc = fma(aa[j].y, exp_neg_half(r2b), acc);
                }
                if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            const double wsum = wave_sum_dpp(acc);
            if (lane == 63) scratch[par][wv - 1] = wsum;
        }
        __syncthreads();
        if (comm) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += scratch[par][w];
            total += s;
            qv += 1e-9 * s;
        }
    }
    const long long t1 = clock64();
    if (tid == 0) { out[blockIdx.x] = total; ticks[blockIdx.x] = t1 - t0; }
}

template <int NW, int PPT, int NORM, int QS>
static int run_ksum(const double* Xt, const double* alpha, int Npad, double* out, long long* ticks) {
    const int K = 4000, G = 128;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((ksum<NW, PPT, NORM, QS>), dim3(G), dim3(64 * (NW + 1)), 0, 0, Xt, alpha, Npad, K, out, ticks);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
    double o; CK(hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost));
    printf("ksum  compute waves %2d (%d per SIMD) pairs/lane %d  %s  q in %s: %.3f us per proposal (wall), %.0f ticks  [sum %.6e]\n",
           NW, (NW + 3) / 4, PPT, NORM == 0 ? "diff form, exp_neg_half     " : NORM == 1 ? "norm form, exp poly13       " : "norm form, exp deg10 squared",
           QS == 2 ? "SGPR, table exp" : QS ? "SGPR" : "VGPR", 1e3 * best / K, (double)h / K, o);
    return 0;
}

int main() {
    double* out; long long* ticks;
    CK(hipMalloc(&out, 256 * 1024 * 8)); CK(hipMalloc(&ticks, 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int w = 1; w <= 8; w *= 2) {
        if (256 * w > 1024) break;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(fma_rate, dim3(256), dim3(256 * w), 0, 0, out, ticks, iters, 0.999);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
        const double n = (double)iters * 64.0;                     // instructions per wave
        const double ns_per_instr_simd = 1e6 * best / (n * w);
        printf("v_fma_f64, 256 CUs busy, %d wave(s) per SIMD: %.3f ms wall, %.2f ns per instruction per SIMD = %.1f TFLOP/s chip; "
               "%.2f ticks per instruction per wave (s_memtime), tick = %.3f ns\n", w, best, ns_per_instr_simd,
               128.0 / ns_per_instr_simd * 1024.0 / 1e3, (double)h / n, 1e6 * best / (double)h);
    }
    // training set: N = 2000 (Npad 2048), d = 10
    const int Npad = 2048, D = 10;
    std::vector<double> hx((size_t)D * Npad), ha(Npad);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0; };
    for (auto& v : hx) v = 6.0 * rnd() - 3.0;
    for (auto& v : ha) v = rnd() - 0.5;
    double *Xt, *alpha;
    CK(hipMalloc(&Xt, hx.size() * 8)); CK(hipMalloc(&alpha, ha.size() * 8));
    CK(hipMemcpy(Xt, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(alpha, ha.data(), ha.size() * 8, hipMemcpyHostToDevice));
    if (run_ksum<4, 4, 0, 1>(Xt, alpha, Npad, out, ticks)) return 1;     // the shipped configuration
    if (run_ksum<4, 4, 0, 2>(Xt, alpha, Npad, out, ticks)) return 1;     // QS = 2: the same with the table-based exp
    if (run_ksum<4, 4, 1, 1>(Xt, alpha, Npad, out, ticks)) return 1;
    if (run_ksum<4, 4, 2, 1>(Xt, alpha, Npad, out, ticks)) return 1;
    if (run_ksum<8, 2, 0, 0>(Xt, alpha, Npad, out, ticks)) return 1;
    if (run_ksum<8, 2, 1, 0>(Xt, alpha, Npad, out, ticks)) return 1;
    if (run_ksum<8, 2, 2, 0>(Xt, alpha, Npad, out, ticks)) return 1;
    if (run_ksum<8, 2, 2, 1>(Xt, alpha, Npad, out, ticks)) return 1;
    if (run_ksum<15, 1, 2, 0>(Xt, alpha, Npad, out, ticks)) return 1;
    return 0;
}
