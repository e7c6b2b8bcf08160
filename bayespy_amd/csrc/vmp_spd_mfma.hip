// vmp_spd_mfma.hip -- batched SPD inverse / Gaussian moments for 16 < n <= 32 on the fp64
// matrix cores: linalg.chol + chol_inv + chol_logdet (utils/linalg.py:31-223) and the fused
// GaussianARDDistribution.compute_moments_and_cgf (gaussian.py:680-706) for per-plate
// posteriors (N matrices of K x K, the masked-data path of SURVEY.md 8f.1).
//
// One wavefront owns one matrix, held as 2 x 2 accumulator tiles of v_mfma_f64_16x16x4_f64
// (element (16 tr + (l>>4) + 4 r, 16 tc + (l&15)) in register r of tile (tr, tc): 32 VGPRs).
// The matrix is inverted by the SYMMETRIC SWEEP operator with 4 x 4 pivot blocks:
//     D = M[B,B],  P = D^-1 M[B,:]          (4 x 32 row panel)
//     M <- M - M[:,B] P                      (rank-4 update = one MFMA per tile)
//     M[B,:] <- P,  M[:,B] <- P^T,  M[B,B] <- -D^-1
// after the eight blocks M = -A^-1.  Why this maps to the matrix core without data movement:
// rows 4p..4p+3 of the matrix are exactly register p%4 of tile row p/4 with l>>4 = row, i.e.
// already in the B-operand layout (k = l>>4, j = l&15); and because the swept matrix stays
// symmetric, the column panel M[:,B] = M[B,:]^T in the A-operand layout (i = l&15, k = l>>4)
// is THE SAME register of the same lane.  The only scalar work per block is the 4 x 4 pivot
// block: its 10 entries are broadcast through SGPRs (v_readlane), every lane factors it and
// solves for the one column of D^-1 it needs as an operand.  No LDS, no barriers; a Gauss-Jordan
// sweep with one row per lane (the n <= 16 kernel) needs 32 serial broadcast steps instead.
#include "vmp_common.h"

namespace {

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
    // Cholesky D = L L^T (lower), reciprocal pivots
    const double p0 = d[0][0];
    const double i0 = fast_recip(sqrt(p0));
    const double l10 = d[0][1] * i0, l20 = d[0][2] * i0, l30 = d[0][3] * i0;
    const double p1 = d[1][1] - l10 * l10;
    const double i1 = fast_recip(sqrt(p1));
    const double l21 = (d[1][2] - l20 * l10) * i1, l31 = (d[1][3] - l30 * l10) * i1;
    const double p2 = d[2][2] - l20 * l20 - l21 * l21;
    const double i2 = fast_recip(sqrt(p2));
    const double l32 = (d[2][3] - l30 * l20 - l31 * l21) * i2;
    const double p3 = d[3][3] - l30 * l30 - l31 * l31 - l32 * l32;
    const double i3 = fast_recip(sqrt(p3));
    if (!(p0 > 0.0 && p1 > 0.0 && p2 > 0.0 && p3 > 0.0)) bad = 1;
    logdet_accumulate(p0 * p1, prod, ld);
    logdet_accumulate(p2 * p3, prod, ld);
    // column c = l15 & 3 of D^-1: L y = e_c, L^T x = y
    const int c = l15 & 3;
    const double e0 = (c == 0) ? 1.0 : 0.0, e1 = (c == 1) ? 1.0 : 0.0;
    const double e2 = (c == 2) ? 1.0 : 0.0, e3 = (c == 3) ? 1.0 : 0.0;
    const double y0 = e0 * i0;
    const double y1 = (e1 - l10 * y0) * i1;
    const double y2 = (e2 - l20 * y0 - l21 * y1) * i2;
    const double y3 = (e3 - l30 * y0 - l31 * y1 - l32 * y2) * i3;
    const double x3 = y3 * i3;
    const double x2 = (y2 - l32 * x3) * i2;
    const double x1 = (y1 - l21 * x2 - l31 * x3) * i1;
    const double x0 = (y0 - l10 * x1 - l20 * x2 - l30 * x3) * i0;
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
    const v4f64 Cp0 = mfma(R0, bop, zero);              // P^T on the block columns of tile (0, TP)
    const v4f64 Cp1 = mfma(R1, bop, zero);
    // ---- rank-4 update of all four tiles ----------------------------------------------------------
    T[0][0] = mfma(-R0, Pv0, T[0][0]);
    T[0][1] = mfma(-R0, Pv1, T[0][1]);
    T[1][0] = mfma(-R1, Pv0, T[1][0]);
    T[1][1] = mfma(-R1, Pv1, T[1][1]);
    // ---- the swept rows and columns -----------------------------------------------------------------
    T[TP][0][RR] = Pv0;
    T[TP][1][RR] = Pv1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T[0][TP][r] = incol ? Cp0[r] : T[0][TP][r];
        T[1][TP][r] = incol ? Cp1[r] : T[1][TP][r];
    }
    T[TP][TP][RR] = incol ? -val : T[TP][TP][RR];
}

template <int P>
__device__ __forceinline__ void sweep_all(v4f64 (&T)[2][2], int l15, int l4, double &prod,
                                          double &ld, int &bad)
{
    if constexpr (P < 8) {
        sweep_block<P>(T, l15, l4, prod, ld, bad);
        sweep_all<P + 1>(T, l15, l4, prod, ld, bad);
    }
}

// MOMENTS = false: A -> A^-1, log|A|.
// MOMENTS = true : (phi0 = `rhs`, phi1 = `A`) -> u0 = Cov phi0 (`vec_out`), u1 = Cov + u0 u0^T
// (`Ainv`), g = -1/2 u0.phi0 + 1/2 log|-2 phi1| (`logdet`), Cov = (-2 phi1)^-1.
template <bool MOMENTS>
__global__ void __launch_bounds__(256, 2)
spd_batched_mfma_kernel(int n, int64_t batch, const double *__restrict__ A,
                        const double *__restrict__ rhs, double *__restrict__ Ainv,
                        double *__restrict__ vec_out, double *__restrict__ logdet,
                        int32_t *__restrict__ info)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = l & 15, l4 = l >> 4;
    const int64_t nn = (int64_t)n * n;
    constexpr double SC = MOMENTS ? -1.0 : 0.5;          // -2 phi1, symmetrised / symmetrise
    for (int64_t b = (int64_t)blockIdx.x * 4 + w; b < batch; b += (int64_t)gridDim.x * 4) {
        const double *Ab = A + b * nn;
        v4f64 T[2][2];
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * tr + l4 + 4 * r, col = 16 * tc + l15;
                    T[tr][tc][r] = (row < n && col < n)
                                       ? SC * (Ab[row * n + col] + Ab[col * n + row])
                                       : ((row == col) ? 1.0 : 0.0);
                }
        double prod = 1.0, ld = 0.0;
        int bad = 0;
        sweep_all<0>(T, l15, l4, prod, ld, bad);
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) T[tr][tc] = -T[tr][tc];          // -(-A^-1)
        double lg = logdet_finish(prod, ld);
        if constexpr (MOMENTS) {
            const double *pb = rhs + b * n;
            // x^T = phi0^T Cov: the tiles are B operands as they are (rows l4 + 4r), phi0 enters
            // through row 0 of the A operand
            v4f64 xa[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
            for (int tr = 0; tr < 2; ++tr)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * tr + 4 * r + l4;
                    const double a = (l15 == 0 && m < n) ? pb[m] : 0.0;
                    xa[0] = mfma(a, T[tr][0][r], xa[0]);
                    xa[1] = mfma(a, T[tr][1][r], xa[1]);
                }
            // lanes 0..15 hold x[16 t + l15] in register 0: both operand layouts of x x^T
            const double x0 = (l4 == 0) ? xa[0][0] : 0.0, x1 = (l4 == 0) ? xa[1][0] : 0.0;
            T[0][0] = mfma(x0, x0, T[0][0]);
            T[0][1] = mfma(x0, x1, T[0][1]);
            T[1][0] = mfma(x1, x0, T[1][0]);
            T[1][1] = mfma(x1, x1, T[1][1]);
            double s = 0.0;
            if (l4 == 0) {
                if (l15 < n) {
                    s += x0 * pb[l15];
                    vec_out[b * n + l15] = x0;
                }
                if (16 + l15 < n) {
                    s += x1 * pb[16 + l15];
                    vec_out[b * n + 16 + l15] = x1;
                }
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
            s = readlane_f64(s, 0);
            lg = -0.5 * s + 0.5 * lg;
        }
        if (Ainv) {
            double *Ob = Ainv + b * nn;
#pragma unroll
            for (int tr = 0; tr < 2; ++tr)
#pragma unroll
                for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * tr + l4 + 4 * r, col = 16 * tc + l15;
                        if (row < n && col < n) Ob[row * n + col] = T[tr][tc][r];
                    }
        }
        if (l == 0) {
            if (logdet) logdet[b] = lg;
            if (info) info[b] = bad;
        }
    }
}

}  // namespace

// Launchers used by vmp_spd_batched / vmp_gaussian_moments (vmp_generic.hip).
int32_t vmp_launch_spd_mfma(vmp_ctx *ctx, bool moments, int32_t n, int64_t batch, const double *A,
                            const double *rhs, double *Ainv, double *vec_out, double *logdet,
                            int32_t *info)
{
    int64_t blocks = (batch + 3) / 4;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    if (moments)
        hipLaunchKernelGGL(spd_batched_mfma_kernel<true>, dim3((unsigned)blocks), dim3(256), 0,
                           ctx->stream, n, batch, A, rhs, Ainv, vec_out, logdet, info);
    else
        hipLaunchKernelGGL(spd_batched_mfma_kernel<false>, dim3((unsigned)blocks), dim3(256), 0,
                           ctx->stream, n, batch, A, rhs, Ainv, vec_out, logdet, info);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}
