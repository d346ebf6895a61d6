// Device helpers shared by the predict and ensemble kernels (gfx950, wave64).
#pragma once
#include "common.hpp"

namespace alabi {

// Compile-time dimension buckets: a GP of dimension d runs the instantiation for the
// smallest bucket >= d; Xt rows beyond d are zero and so contribute nothing to r^2.
__host__ __device__ inline int dim_bucket(int d) {
    const int b[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64};
    for (int i = 0; i < 15; ++i)
        if (d <= b[i]) return b[i];
    return -1;
}

#define ALABI_DISPATCH_DIM(DB, ...)                                   \
    switch (DB) {                                                     \
        case 1: { constexpr int D = 1; __VA_ARGS__; } break;          \
        case 2: { constexpr int D = 2; __VA_ARGS__; } break;          \
        case 3: { constexpr int D = 3; __VA_ARGS__; } break;          \
        case 4: { constexpr int D = 4; __VA_ARGS__; } break;          \
        case 5: { constexpr int D = 5; __VA_ARGS__; } break;          \
        case 6: { constexpr int D = 6; __VA_ARGS__; } break;          \
        case 8: { constexpr int D = 8; __VA_ARGS__; } break;          \
        case 10: { constexpr int D = 10; __VA_ARGS__; } break;        \
        case 12: { constexpr int D = 12; __VA_ARGS__; } break;        \
        case 16: { constexpr int D = 16; __VA_ARGS__; } break;        \
        case 20: { constexpr int D = 20; __VA_ARGS__; } break;        \
        case 24: { constexpr int D = 24; __VA_ARGS__; } break;        \
        case 32: { constexpr int D = 32; __VA_ARGS__; } break;        \
        case 48: { constexpr int D = 48; __VA_ARGS__; } break;        \
        case 64: { constexpr int D = 64; __VA_ARGS__; } break;        \
        default: return ALABI_BAD_ARGUMENT;                           \
    }

// exp(-r2 / 2) for r2 >= 0 in 20 fp64 instructions (the library exp costs 27: it also handles overflow, NaN payloads and
// positive arguments).  Cody-Waite reduction x = n ln2 + r with |r| <= ln2/2, degree-13 Taylor polynomial in Horner form
// (truncation 4e-18), scaling by ldexp; large r2 underflows to 0 through ldexp.  Maximum error 1.23 ulp against exp of the
// exact argument on 6e5 samples in [0, 1500] (NumPy's exp: 1.18 ulp on the same samples).  Every kernel of the library uses
// this one function for the squared-exponential, so their results stay mutually consistent.
// Range: r2 < ~1e40 (beyond that the reduction's residual overflows the polynomial; coordinates of 1e20 length scales).
__device__ inline double exp_neg_half(double r2) {
    const double x = -0.5 * r2;
    const double n = rint(x * 1.4426950408889634);
    double r = fma(n, -0x1.62e42fee00000p-1, x);
    r = fma(n, -0x1.a39ef35793c76p-33, r);
    double p = 1.6059043836821613e-10;                   // 1/13!
    p = fma(p, r, 2.08767569878681e-09);                 // 1/12!
    p = fma(p, r, 2.505210838544172e-08);                // 1/11!
    p = fma(p, r, 2.755731922398589e-07);                // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);               // 1/9!
    p = fma(p, r, 2.48015873015873e-05);                 // 1/8!
    p = fma(p, r, 0.0001984126984126984);                // 1/7!
    p = fma(p, r, 0.001388888888888889);                 // 1/6!
    p = fma(p, r, 0.008333333333333333);                 // 1/5!
    p = fma(p, r, 0.041666666666666664);                 // 1/4!
    p = fma(p, r, 0.16666666666666666);                  // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// exp(x) for an argument that is already -r2/2 (the matrix-core predict path forms it as one dot product): the same
// reduction and polynomial as exp_neg_half, without the multiply.  x may exceed 0 by rounding; large |x| underflows to 0.
__device__ inline double exp_direct(double x) {
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

// exp(x), x <= ~0, with a 32-entry table of 2^(j/32) in LDS (`tab`, filled by the caller: tab[j] = exp2(j / 32.0)):
// x = (32 n + j) ln2/32 + r, |r| <= ln2/64, degree-6 Taylor polynomial (truncation 3e-18), result 2^n tab[j] p(r).
// 13 instructions on the fp64 pipe instead of 19, plus 3 integer instructions and one LDS read -- pays where the pipe is the
// bound (several waves per SIMD), not in the one-wave-per-SIMD ensemble kernels (tools/micro/ksum_bench: -1 %).
__device__ inline double exp_tab32(double x, const double* tab) {
    const double k = rint(x * 46.16624130844683);                 // 32 / ln2
    double r = fma(k, -0x1.62e42fee00000p-6, x);                   // ln2 / 32, high part (exact product for |k| < 2^20)
    r = fma(k, -0x1.a39ef35793c76p-38, r);                         // low part
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

// 2^(s / 64) = exp(s ln2 / 64) for an argument ALREADY multiplied by 64 / ln2 (the matrix-core predict kernel folds the factor
// into its query operand, so the product arrives scaled), with a 64-entry table tab[j] = 2^(j/64) in LDS.  s = 64 n + j + r
// with r = s - rint(s) exact and |r| <= 1/2: rint comes from adding 1.5 * 2^52, whose low word then holds 64 n + j as an
// integer (no convert instruction); degree-5 Taylor polynomial in r (truncation 3.5e-17); result 2^n tab[j] p(r).  The clamp
// keeps the integer inside 32 bits; everything below 2^-1094 is 0 through ldexp anyway.  11 instructions on the fp64 pipe
// against 13 for exp_tab32 and 19 for exp_direct.
#define ALABI_EXP2S_SCALE 92.33248261689366                          /* 64 / ln2 */
__device__ inline double exp2s_tab64(double s, const double* tab) {
    s = fmax(s, -70016.0);
    const double t = s + 0x1.8p52;
    const double r = s - (t - 0x1.8p52);
    const int ki = __double2loint(t);
    const double tj = tab[ki & 63];
    double p = 0x1.5d87fe78a6731p-40;                               // (ln2/64)^5 / 5!
    p = fma(p, r, 0x1.3b2ab6fba4e77p-31);
    p = fma(p, r, 0x1.c6b08d704a0c0p-23);
    p = fma(p, r, 0x1.ebfbdff82c58fp-15);
    p = fma(p, r, 0x1.62e42fefa39efp-7);
    p = fma(p, r, 1.0);
    return ldexp(p * tj, ki >> 6);
}

// The same with a 256-entry table tab[j] = 2^(j/256) (2 KB of LDS) for an argument multiplied by 256 / ln2: |r| <= 1/2 is then
// |x| <= ln2/512 and a degree-4 polynomial suffices (truncation x^5/120 = 3.8e-17): 10 instructions on the fp64 pipe (the clamp
// costs two of them: the compiler quiets a possible signalling NaN before v_max_f64; an inline-asm v_max_f64 would save one but
// reads the matrix-core result without the wait states only the compiler inserts).  <= 2.2 ulp against exp of the exact argument (2e5 samples, exact-arithmetic emulation).  Used by the group
// ensemble kernel, whose time is the fp64 pipe's.
#define ALABI_EXP2S256_SCALE 369.3299304675746                       /* 256 / ln2 */
__device__ inline double exp2s_tab256(double s, const double* tab) {
    s = fmax(s, -280064.0);
    const double t = s + 0x1.8p52;
    const double r = s - (t - 0x1.8p52);
    const int ki = __double2loint(t);
    const double tj = tab[ki & 255];
    double p = 0x1.3b2ab6fba4e77p-39;                                /* (ln2/256)^4 / 4! */
    p = fma(p, r, 0x1.c6b08d704a0c0p-29);                            /* (ln2/256)^3 / 3! */
    p = fma(p, r, 0x1.ebfbdff82c58fp-19);                            /* (ln2/256)^2 / 2! */
    p = fma(p, r, 0x1.62e42fefa39efp-9);                             /* ln2/256 */
    p = fma(p, r, 1.0);
    return ldexp(p * tj, ki >> 8);
}

// GENERIC = false compiles the squared-exponential alone (no run-time switch in the hot loops).
template <bool GENERIC = true>
__device__ inline double radial(double r2, KernelFn kf) {
    if (!GENERIC || kf.type == 0) return exp_neg_half(r2);
    if (kf.type == 1) { const double r = sqrt(3.0 * r2); return (1.0 + r) * exp(-r); }
    if (kf.type == 2) { const double r = sqrt(5.0 * r2); return (1.0 + r + r * r / 3.0) * exp(-r); }
    return exp(-kf.alpha * log1p(0.5 * r2 / kf.alpha));
}

// One 64 x 64 tile (bi >= bj) of K = amp f(r2) + wn I from the scaled, transposed inputs Xt [d][Npad], identity in the padding;
// 256 threads, each a 4 x 4 micro-tile (gp_assemble.hip).  xi / xj: the caller's LDS staging arrays.
__device__ inline void assemble_tile(double (*xi)[64], double (*xj)[64], const double* __restrict__ Xt, int N, int Npad, int d,
                                     double amp, double wn, KernelFn kf, double* __restrict__ K, int bi, int bj) {
    const int tid = threadIdx.x;
    for (int e = tid; e < d * 64; e += 256) {
        int k = e >> 6, c = e & 63;
        xi[k][c] = Xt[(size_t)k * Npad + bi * 64 + c];
        xj[k][c] = Xt[(size_t)k * Npad + bj * 64 + c];
    }
    __syncthreads();
    const int r0 = 4 * (tid >> 4), c0 = tid & 15;
    double r2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
    for (int k = 0; k < d; ++k) {
        double xa[4], xb[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xa[a] = xi[k][r0 + a];
#pragma unroll
        for (int b = 0; b < 4; ++b) xb[b] = xj[k][c0 + 16 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double df = xa[a] - xb[b];
                r2[a][b] = fma(df, df, r2[a][b]);
            }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int gr = bi * 64 + r0 + a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gc = bj * 64 + c0 + 16 * b;
            double v = amp * radial(r2[a][b], kf);
            if (gr == gc) v += wn;
            if (gr >= N || gc >= N) v = (gr == gc) ? 1.0 : 0.0;
            K[(size_t)gr * Npad + gc] = v;
        }
    }
}

// Run `...` with `constexpr bool GENERIC` = (kernel family != squared exponential).
#define ALABI_DISPATCH_KERNEL(KTYPE, ...)                         \
    if ((KTYPE) == 0) { constexpr bool GENERIC = false; __VA_ARGS__; } \
    else { constexpr bool GENERIC = true; __VA_ARGS__; }

// Sum over the 64 lanes of a wavefront (result valid in every lane).
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One DPP data move of a double (two 32-bit halves); lanes without a source read 0.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Wave-wide sum with DPP (no LDS round trips): row_shr 1,2,4,8 build the 16-lane row totals in lanes
// 15/31/47/63, row_bcast:15 and row_bcast:31 fold the rows; the total ends in lane 63.  The order of
// the additions is fixed, so the result is reproducible.  ~6 x (2 v_mov_dpp + v_add_f64).
__device__ inline double wave_sum_dpp(double v) {
    v += dpp_move<0x111, 0xf>(v);   // row_shr:1
    v += dpp_move<0x112, 0xf>(v);   // row_shr:2
    v += dpp_move<0x114, 0xf>(v);   // row_shr:4
    v += dpp_move<0x118, 0xf>(v);   // row_shr:8
    v += dpp_move<0x142, 0xa>(v);   // row_bcast:15 -> rows 1 and 3
    v += dpp_move<0x143, 0xc>(v);   // row_bcast:31 -> rows 2 and 3
    return v;                       // valid in lane 63
}

__device__ inline double lane_bcast(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Sum over the workgroup (any multiple of 64 threads up to 1024) in a fixed order; `scratch` is >= 16
// doubles of LDS.  Every thread returns the total.
__device__ inline double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();  // scratch may still be read from a previous call
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    double t = scratch[0];
    for (int k = 1; k < nw; ++k) t += scratch[k];
    return t;
}

// sum_n alpha[n] * exp(-0.5 * |Xt[:,n] - q|^2) over all (padded) training points, one
// query per workgroup: lanes run along n (coalesced SoA loads), then a block reduction.
// q holds the query already multiplied by inv_len; padded points carry alpha = 0.
template <int D, bool GENERIC = true>
__device__ inline double gp_kernel_dot_block(const double* __restrict__ Xt, const double* __restrict__ alpha,
                                             int Npad, const double* q_lds, double* scratch, KernelFn kf) {
    double q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) q[k] = q_lds[k];
    double acc = 0.0;
    for (int n = threadIdx.x; n < Npad; n += blockDim.x) {
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double df = Xt[(size_t)k * Npad + n] - q[k];
            r2 = fma(df, df, r2);
        }
        acc = fma(alpha[n], radial<GENERIC>(r2, kf), acc);
    }
    return block_sum(acc, scratch);
}

// One ds_read_b64 that the compiler will not pair with a neighbour into a ds_read2_b64 (volatile, LDS address space).
// ds_read2_b64 is banked modulo 32 dwords and serviced in groups of 16 lanes (8 LDS cycles even without conflicts);
// ds_read_b64 is banked modulo 64 in groups of 32 lanes: 2 cycles when the row stride is = 2 (mod 32) doubles for MFMA A / B
// operands read as [row = lane & 15][k = lane >> 4] (MI355X_MICROARCH.md, LDS).
__device__ inline double lds_read_b64(const double* p) {
    return *(const volatile __attribute__((address_space(3))) double*)p;
}

}  // namespace alabi
