// vmp_gmm_dev.h -- device helpers shared by the mixture pass kernels (vmp_gmm.hip: D <= 16 with the
// coefficient fragments in LDS; vmp_gmm_wide.hip: 16 < D <= 32 with the fragments streamed from L2).
#pragma once
#include "vmp_common.h"
#include "vmp_exp2_table.h"

namespace {

constexpr int TNC = 16;          // columns (plate elements) per wave tile
constexpr int RS = 18;           // row stride of the k-major r tile

__device__ inline v4f64 mfma_f64(double a, double b, v4f64 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// exp(x) for x <= 0 (softmax arguments after subtracting the maximum; -inf allowed).  On this
// chip fp64 vector work is not hidden behind fp64 matrix work (both run on the same fp64 units),
// so the exponential is the second largest cost of the pass after the MFMAs.  Table form:
// x = (256 m + j) ln2/256 + r, |r| <= ln2/512, exp(x) = 2^m * T[j] * (1 + r + r^2/2 + r^3/6 +
// r^4/24) (truncation 4e-17 relative); T = 2^(j/256) correctly rounded, in LDS.  11 fp64
// instructions instead of 21 for the polynomial-only form (exp_nonpos, vmp_common.h); <= 2 ulp.
typedef __attribute__((address_space(3))) double lds_f64;

__device__ __forceinline__ double lds_read(uint32_t byte_addr)
{
    return *(const lds_f64 *)(uintptr_t)byte_addr;
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t byte_addr)
{
    return *(const lds_u32 *)(uintptr_t)byte_addr;
}

// v_max_f64 without the canonicalising v_max x, x the compiler puts in front of fmax() operands
// it cannot prove quiet (MFMA results, shuffled values): they never hold signalling NaNs.
// The compiler's hazard recogniser does not look inside inline assembly, and a vector
// instruction that reads a register too soon after the MFMA that writes it gets stale data:
// mfma_settle() below must separate the matrix instructions from the first max_raw().
__device__ __forceinline__ double max_raw(double a, double b)
{
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// 32 wait states after the MFMAs that produce `acc` (the longest MFMA-write -> VALU-read
// requirement on this target is below 20), tied to the accumulators so that neither the
// MFMAs nor their consumers can be scheduled across it.
template <int KT>
__device__ __forceinline__ void mfma_settle(v4f64 (&acc)[KT])
{
    if constexpr (KT == 1)
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]));
    else if constexpr (KT == 2)
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
    else
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}

// NV exponentials at once, staged so that the NV table reads are in flight together:
// v[i] <- exp(v[i] - mx)
template <int NV>
__device__ __forceinline__ void exp_tab_batch(double (&v)[NV], double mx, uint32_t tab_addr)
{
    int ki[NV];
    double t[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double x = max_raw(v[i] - mx, -800.0);
        const double kf = __builtin_rint(x * 0x1.71547652b82fep+8);            // 256 / ln 2
        double r = __builtin_fma(kf, -0x1.62e42fee00000p-9, x);                 // ln2_hi / 256
        r = __builtin_fma(kf, -0x1.a39ef35793c76p-41, r);                       // ln2_lo / 256
        ki[i] = (int)kf;
        t[i] = lds_read(tab_addr + 8u * __builtin_amdgcn_ubfe((uint32_t)ki[i], 0u, 8u));
        v[i] = r;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double r = v[i];
        double p = __builtin_fma(r, 1.0 / 24.0, 1.0 / 6.0);
        p = __builtin_fma(p, r, 0.5);
        p = __builtin_fma(p, r, 1.0);
        v[i] = __builtin_fma(p, r, 1.0);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_ldexp(t[i] * v[i], ki[i] >> 8);
}

// 1 / s for 1 <= s <= 2^10: hardware estimate + two Newton steps (no scaling, no special cases)
__device__ __forceinline__ double recip_small(double s)
{
    double y = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-s, y, 1.0);
    return __builtin_fma(y, e, y);
}

}  // namespace

// vmp_gmm_wide.hip
int32_t vmp_gmm_wide_workspace_doubles(vmp_ctx *ctx, int D, int K, int64_t F2P, int64_t KP,
                                       int64_t *partial_doubles, int64_t *frag_doubles);
int32_t vmp_gmm_wide_pass(vmp_ctx *ctx, const double *Y, int64_t N, int D, int K, int64_t F2P,
                          int64_t KP, const double *Cmat, const int64_t *labels, double *R,
                          double *P, double *Cfrag, int *nb_out);
