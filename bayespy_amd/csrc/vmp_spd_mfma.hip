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
#include "vmp_sweep.h"

namespace {

using namespace vmp_sweep;

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
    // the wavefront index as a scalar: the matrix base address then lives in SGPRs and the 64
    // element loads / stores share two per-lane offsets instead of 64 per-lane addresses
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
        sweep_upto<0>(T, (n + 3) / 4, l15, l4, prod, ld, bad);
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * tr + l4 + 4 * r, col = 16 * tc + l15;
                    // -(-A^-1) on the swept part; the padding stays the identity
                    T[tr][tc][r] = (row < n && col < n) ? -T[tr][tc][r] : T[tr][tc][r];
                }
        double lg = sweep_logdet(prod, ld);
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
