// vmp_sweep.h -- a 32 x 32 symmetric positive definite matrix held by ONE wavefront as 2 x 2
// accumulator tiles of v_mfma_f64_16x16x4_f64 and inverted in registers by the symmetric sweep
// operator with 4 x 4 pivot blocks (see vmp_spd_mfma.hip for the derivation).  Element
// (16 tr + (l>>4) + 4 r, 16 tc + (l&15)) of the matrix is register r of tile (tr, tc) in lane l.
#pragma once
#include "vmp_common.h"

namespace vmp_sweep {

__device__ inline v4f64 mfma(double a, double b, v4f64 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ inline double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/x to fp64 round-off: hardware estimate + one cubic step (e = 1 - x r; r <- r (1 + e + e^2))
__device__ inline double fast_recip3(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

// log-determinant from the running (mantissa, exponent) pair of sweep_block
__device__ inline double sweep_logdet(double prod, double ld)
{
    return log(prod) + ld * 0.69314718055994530942;
}

// One sweep over pivot block P (rows / columns 4P .. 4P+3).
template <int P>
__device__ __forceinline__ void sweep_block(v4f64 (&T)[2][2], int l15, int l4, double &prod,
                                            double &ld, int &bad)
{
    constexpr int TP = P / 4, RR = P % 4, C0 = 4 * (P % 4);
    // ---- the 4 x 4 pivot block, uniform in all lanes ----------------------------------------
    const double pan = T[TP][TP][RR];
    double d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) d[a][b] = readlane_f64(pan, a * 16 + C0 + b);
    // D = L diag(p) L^T (unit lower L): no square roots, and one multiplication per solve step
    // less than the Cholesky form (16 + 16 operations against 16 + 20, 4 reciprocals of 4
    // operations against 4 inverse square roots of 9)
    const double p0 = d[0][0];
    const double r0 = fast_recip3(p0);
    const double l10 = d[0][1] * r0, l20 = d[0][2] * r0, l30 = d[0][3] * r0;
    const double p1 = __builtin_fma(-l10, d[0][1], d[1][1]);
    const double r1 = fast_recip3(p1);
    const double u21 = __builtin_fma(-l20, d[0][1], d[1][2]);
    const double u31 = __builtin_fma(-l30, d[0][1], d[1][3]);
    const double l21 = u21 * r1, l31 = u31 * r1;
    const double p2 = __builtin_fma(-l21, u21, __builtin_fma(-l20, d[0][2], d[2][2]));
    const double r2 = fast_recip3(p2);
    const double u32 = __builtin_fma(-l31, u21, __builtin_fma(-l30, d[0][2], d[2][3]));
    const double l32 = u32 * r2;
    const double p3 = __builtin_fma(-l32, u32, __builtin_fma(-l31, u31,
                                                             __builtin_fma(-l30, d[0][3], d[3][3])));
    const double r3 = fast_recip3(p3);
    if (!(p0 > 0.0 && p1 > 0.0 && p2 > 0.0 && p3 > 0.0)) bad = 1;
    // running determinant as mantissa x 2^exponent (`ld` counts the exponent): no logarithm on
    // the serial path, one at the very end (sweep_logdet)
    const double q0 = prod * (p0 * p1);
    ld += (double)__builtin_amdgcn_frexp_exp(q0);
    const double q1 = __builtin_amdgcn_frexp_mant(q0) * (p2 * p3);
    ld += (double)__builtin_amdgcn_frexp_exp(q1);
    prod = __builtin_amdgcn_frexp_mant(q1);
    // column c = l15 & 3 of D^-1: L y = e_c, z = y / p, L^T x = z
    const int c = l15 & 3;
    const double e0 = (c == 0) ? 1.0 : 0.0, e1 = (c == 1) ? 1.0 : 0.0;
    const double e2 = (c == 2) ? 1.0 : 0.0, e3 = (c == 3) ? 1.0 : 0.0;
    const double y1 = __builtin_fma(-l10, e0, e1);
    const double y2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, e0, e2));
    const double y3 = __builtin_fma(-l32, y2, __builtin_fma(-l31, y1, __builtin_fma(-l30, e0, e3)));
    const double x3 = y3 * r3;
    const double x2 = __builtin_fma(-l32, x3, y2 * r2);
    const double x1 = __builtin_fma(-l31, x3, __builtin_fma(-l21, x2, y1 * r1));
    const double x0 = __builtin_fma(-l30, x3, __builtin_fma(-l20, x2, __builtin_fma(-l10, x1, e0 * r0)));
    // this lane's entry D^-1[l4][c]
    const double val = (l4 == 0) ? x0 : (l4 == 1) ? x1 : (l4 == 2) ? x2 : x3;
    const bool incol = (l15 >= C0) && (l15 < C0 + 4);
    const double aop = (l15 < 4) ? val : 0.0;           // A[i][k] = D^-1[i][k], rows i < 4
    const double bop = incol ? val : 0.0;               // B[k][j] = D^-1[k][j - C0] on the block columns
    // ---- panels --------------------------------------------------------------------------------
    const double R0 = T[TP][0][RR], R1 = T[TP][1][RR];  // row panel = column panel transposed
    const v4f64 zero = {0.0, 0.0, 0.0, 0.0};
    const double Pv0 = mfma(aop, R0, zero)[0];          // P = D^-1 R, rows k = l4 in register 0
    const double Pv1 = mfma(aop, R1, zero)[0];
    // ---- rank-4 update of all four tiles; in tile column TP the block columns receive
    //      P^T = R^T D^-1 instead (old values dropped, operand D^-1 in place of -P) --------------
    const double B0 = (TP == 0 && incol) ? bop : -Pv0;
    const double B1 = (TP == 1 && incol) ? bop : -Pv1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T[0][TP][r] = incol ? 0.0 : T[0][TP][r];
        T[1][TP][r] = incol ? 0.0 : T[1][TP][r];
    }
    T[0][0] = mfma(R0, B0, T[0][0]);
    T[0][1] = mfma(R0, B1, T[0][1]);
    T[1][0] = mfma(R1, B0, T[1][0]);
    T[1][1] = mfma(R1, B1, T[1][1]);
    // ---- the swept rows and the pivot block ------------------------------------------------------
    T[TP][0][RR] = (TP == 0 && incol) ? -val : Pv0;
    T[TP][1][RR] = (TP == 1 && incol) ? -val : Pv1;
}

// Sweep the first `nblocks` pivot blocks (rows / columns beyond 4 nblocks hold the identity and
// are left alone: sweeping them is a no-op).  Afterwards T = -A^-1 on the swept part.
template <int P>
__device__ __forceinline__ void sweep_upto(v4f64 (&T)[2][2], int nblocks, int l15, int l4,
                                           double &prod, double &ld, int &bad)
{
    if constexpr (P < 8) {
        if (P < nblocks) sweep_block<P>(T, l15, l4, prod, ld, bad);
        sweep_upto<P + 1>(T, nblocks, l15, l4, prod, ld, bad);
    }
}

// Two independent matrices swept block by block in one instruction stream: the serial pivot
// chain of one (v_readlane -> 4 x 4 Cholesky -> solve -> MFMA) fills the latency gaps of the other.
template <int P>
__device__ __forceinline__ void sweep_pair(v4f64 (&Ta)[2][2], v4f64 (&Tb)[2][2], int nblocks,
                                           int l15, int l4, double &pa, double &la, int &ba,
                                           double &pb, double &lb, int &bb)
{
    if constexpr (P < 8) {
        if (P < nblocks) {
            sweep_block<P>(Ta, l15, l4, pa, la, ba);
            sweep_block<P>(Tb, l15, l4, pb, lb, bb);
        }
        sweep_pair<P + 1>(Ta, Tb, nblocks, l15, l4, pa, la, ba, pb, lb, bb);
    }
}

// NM independent matrices swept block by block in one instruction stream.
template <int P, int NM>
__device__ __forceinline__ void sweep_multi(v4f64 (&T)[NM][2][2], int nblocks, int l15, int l4,
                                            double (&prod)[NM], double (&ld)[NM], int (&bad)[NM])
{
    if constexpr (P < 8) {
        if (P < nblocks) {
#pragma unroll
            for (int m = 0; m < NM; ++m) sweep_block<P>(T[m], l15, l4, prod[m], ld[m], bad[m]);
        }
        sweep_multi<P + 1, NM>(T, nblocks, l15, l4, prod, ld, bad);
    }
}

}  // namespace vmp_sweep
