// vmp_gshared.hip -- ONE pass for the update of a Gaussian node whose posterior covariance is
// shared by its plates (generic engine; gfx950 only).
//
// Replaces, for a GaussianARD / Gaussian node with a plate-free precision (a scalar mask: every
// plate has the same Cov = (-2 phi1)^-1), the NumPy call sites of
//   gaussian.py:649-706  (compute_phi_from_parents + compute_moments_and_cgf: phi0 = prior + sum of
//                         messages, <x> = Cov phi0, the cumulant row -1/2 phi0 . <x>),
//   dot.py:581           (the Dot / SumMultiply message the node receives, m_n = B^T y_n), and the
//   plate sums the neighbours ask for afterwards (dot.py:581 again for the other parent:
//   sum_n y_n <x_n>^T and sum_n <x_n><x_n>^T; expfamily.py:449-468: sum_n <x_n>),
// which the generic engine used to run as ~12 launches over (N, K) / (D, N) arrays.
//
//     m_n = B^T y_n   (Y given: the D x N array the message contracts, B: D x K)
//           or the rows of a given N x K message array
//     x_n = Cov (p0 + m_n)                                         -> x (N x K, any strides)
//     stats = [ sum_n x_n (K) ; sum_n x_n x_n^T (K x K) ; sum_n y_n x_n^T (D x K, Y form only) ]
//
// Y form: the streaming-statistics tile kernel of the fused PCA block (vmp_pca.hip, role 1 /
// role 2 on v_mfma_f64_16x16x4_f64 over a (DP + KP) x 32 tile in LDS) with A = Cov B^T and the
// bias c = Cov p0 made by a one-workgroup launch first, a loader for either memory order of Y
// (plates contiguous / variable axis contiguous) and stores of <x> in the engine's (N, K) order.
// Algorithmic traffic per launch: 8 N (D + K) bytes; flops 4 N D K + 2 N K^2 (fp64 MFMA).
// Message form: rows through LDS, one thread per plate (fp64 vector units), 16 N K bytes.
// Partial sums per workgroup, combined in fixed order (bitwise reproducible).
#include "vmp_common.h"

namespace {

constexpr int TN = 32;        // plates per tile
constexpr int SZ = TN + 2;    // LDS row stride in doubles (== 2 mod 32)
constexpr int NT = 256;

inline int pow2_blocks(int x, int unit)
{
    int b = (x + unit - 1) / unit, p = 1;
    while (p < b) p <<= 1;
    return p;
}

__device__ inline v4f64 mfma_f64(double a, double b, v4f64 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Apad (KP x DP, zero padded) = Cov B^T; cpad (KP) = Cov p0.  Cov and B are staged in LDS first
// (the products then run without a dependent trip to memory per term).
__global__ void __launch_bounds__(NT)
gshared_prepare_kernel(int K, int D, int KP, int DP, const double *__restrict__ cov,
                       const double *__restrict__ B, int64_t b_sd, int64_t b_sk,
                       const double *__restrict__ p0, double *__restrict__ Apad,
                       double *__restrict__ cpad)
{
    extern __shared__ double lds[];
    double *cs = lds, *bs = cs + K * K;          // cs[k][j]; bs[d - d0][j], row stride K + 1
    constexpr int DC = 64;                       // rows of B per round
    for (int e = threadIdx.x; e < K * K; e += NT) cs[e] = cov[e];
    for (int d0 = 0; d0 < DP; d0 += DC) {
        __syncthreads();
        for (int e = threadIdx.x; e < DC * K; e += NT) {
            const int dd = e / K, j = e - dd * K;
            bs[dd * (K + 1) + j] = (d0 + dd < D) ? B[(int64_t)(d0 + dd) * b_sd + j * b_sk] : 0.0;
        }
        __syncthreads();
        const int dn = (DP - d0) < DC ? (DP - d0) : DC;
        for (int e = threadIdx.x; e < KP * dn; e += NT) {
            const int k = e / dn, dd = e - k * dn;
            double s = 0.0;
            if (k < K)
                for (int j = 0; j < K; ++j) s += cs[k * K + j] * bs[dd * (K + 1) + j];
            Apad[k * DP + d0 + dd] = s;
        }
    }
    for (int k = threadIdx.x; k < KP; k += NT) {
        double s = 0.0;
        if (k < K && p0)
            for (int j = 0; j < K; ++j) s += cs[k * K + j] * p0[j];
        cpad[k] = s;
    }
}

// partial block of one workgroup: [ (DP + KP) x KP role-2 sums ; KP sums of x ]
template <int DB, int KT>
__global__ void __launch_bounds__(NT, (DB >= 4 || DB * KT > 8) ? 1 : 2)
gshared_pass_kernel(const double *__restrict__ Y, int64_t y_sd, int64_t y_sn, int64_t N, int D,
                    int K, const double *__restrict__ Apad, const double *__restrict__ cpad,
                    double *__restrict__ X, int64_t x_sn, int64_t x_sk, double *__restrict__ P,
                    int64_t ntiles, int yvec, int xvec)
{
    constexpr int DP = 32 * DB, KP = 16 * KT, ZR = DP + KP;
    constexpr int YP = DP / 16;          // 16-byte loads per thread and tile
    constexpr int KS1 = DP / 4;          // role-1 MFMA k-steps
    constexpr int T1 = KT * (TN / 16);   // role-1 output tiles
    constexpr int R1 = (T1 + 3) / 4;
    constexpr int T2 = (2 * DB + KT) * KT;  // role-2 output tiles
    constexpr int R2 = (T2 + 3) / 4;

    __shared__ __attribute__((aligned(16))) double Z[ZR * SZ];
    __shared__ double SX[4 * 16];

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int l15 = l & 15, l4 = l >> 4;
    const int it1 = w % KT;
    // role-2 tiles are dealt starting behind the wavefronts that carry the role-1 tiles, so that
    // the matrix work of a tile is spread over the four SIMDs (D = 64, K = 16: 24 / 24 / 16 / 8
    // instructions per wavefront instead of 32 / 24 / 8 / 8)
    const int w2 = (w + 4 - (T1 % 4)) % 4;
    const int jt2 = w2 % KT;
    const bool nmajor = (y_sn == 1);     // plates contiguous (the reference's (D, N) data)

    double afrag[KS1];
    {
        const double *arow = Apad + (int64_t)(it1 * 16 + l15) * DP;
#pragma unroll
        for (int q = 0; q < KS1; ++q) afrag[q] = arow[32 * (q >> 3) + (q & 7) + 8 * l4];
    }
    double cv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cv[r] = cpad[it1 * 16 + l4 + 4 * r];
    double sx[4] = {0.0, 0.0, 0.0, 0.0};

    v4f64 acc2[R2];
#pragma unroll
    for (int m = 0; m < R2; ++m) acc2[m] = v4f64{0.0, 0.0, 0.0, 0.0};

    v2f64 yreg[YP];

    // plates contiguous: thread -> (row p*16 + tid/16, columns 2*(tid%16) ..+1)
    // variable axis contiguous: thread -> pair q = p*256 + tid of the tile's 32 x DP elements,
    //   plate q / (DP/2), rows 2*(q % (DP/2)) ..+1
    auto load_tile = [&](int64_t tile) {
        if (nmajor) {
            const int lrow = tid >> 4, lcol = (tid & 15) * 2;
            const int64_t n = tile * TN + lcol;
#pragma unroll
            for (int p = 0; p < YP; ++p) {
                const int row = p * 16 + lrow;
                v2f64 v = v2f64{0.0, 0.0};
                if (row < D) {
                    const double *src = Y + (int64_t)row * y_sd + n;
                    if (n + 1 < N) {
                        if (yvec) v = *reinterpret_cast<const v2f64 *>(src);
                        else { v.x = src[0]; v.y = src[1]; }
                    } else if (n < N) v.x = src[0];
                }
                yreg[p] = v;
            }
        } else {
#pragma unroll
            for (int p = 0; p < YP; ++p) {
                const int q = p * NT + tid;
                const int plate = q / (DP / 2), row = 2 * (q - plate * (DP / 2));
                const int64_t n = tile * TN + plate;
                v2f64 v = v2f64{0.0, 0.0};
                if (n < N && row < D) {
                    const double *src = Y + n * y_sn + (int64_t)row * y_sd;
                    if (row + 1 < D) {
                        if (yvec) v = *reinterpret_cast<const v2f64 *>(src);
                        else { v.x = src[0]; v.y = src[y_sd]; }
                    } else v.x = src[0];
                }
                yreg[p] = v;
            }
        }
    };

    int64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);

    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * TN;
        if (nmajor) {
            const int lrow = tid >> 4, lcol = (tid & 15) * 2;
#pragma unroll
            for (int p = 0; p < YP; ++p)
                *reinterpret_cast<v2f64 *>(&Z[(p * 16 + lrow) * SZ + lcol]) = yreg[p];
        } else {
#pragma unroll
            for (int p = 0; p < YP; ++p) {
                const int q = p * NT + tid;
                const int plate = q / (DP / 2), row = 2 * (q - plate * (DP / 2));
                Z[row * SZ + plate] = yreg[p].x;
                Z[(row + 1) * SZ + plate] = yreg[p].y;
            }
        }
        __syncthreads();

        const int64_t next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);

        // ---- role 1: X_tile = A * Y_tile + c --------------------------------------------
#pragma unroll
        for (int m = 0; m < R1; ++m) {
            const int t1 = w + 4 * m;
            if (t1 < T1) {
                const int jt = t1 / KT;
                const double *zb = Z + jt * 16 + l15 + 8 * l4 * SZ;
                v4f64 c0 = v4f64{0.0, 0.0, 0.0, 0.0};
                v4f64 c1 = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < KS1; q += 2) {
                    const double b0 = zb[(32 * (q >> 3) + (q & 7)) * SZ];
                    const double b1 = zb[(32 * ((q + 1) >> 3) + ((q + 1) & 7)) * SZ];
                    c0 = mfma_f64(afrag[q], b0, c0);
                    c1 = mfma_f64(afrag[q + 1], b1, c1);
                }
                c0 += c1;
                // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
                const int n = jt * 16 + l15;
                const bool nok = (n0 + n) < N;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = it1 * 16 + l4 + 4 * r;
                    const double v = nok ? c0[r] + cv[r] : 0.0;   // (rows k >= K: A and c are zero)
                    Z[(DP + k) * SZ + n] = v;
                    sx[r] += v;
                }
            }
        }
        __syncthreads();

        // ---- <x> of the tile to memory, in the caller's order ---------------------------
        if (x_sk == 1) {
            // rows of K contiguous doubles: pairs (k, k+1) of one plate per thread
#pragma unroll
            for (int p = 0; p < KT; ++p) {
                const int q = p * NT + tid;
                const int plate = q / (KP / 2), k = 2 * (q - plate * (KP / 2));
                const int64_t n = n0 + plate;
                if (n < N && k < K) {
                    double *dst = X + n * x_sn + k;
                    const double a = Z[(DP + k) * SZ + plate];
                    if (k + 1 < K) {
                        const double b = Z[(DP + k + 1) * SZ + plate];
                        if (xvec) *reinterpret_cast<v2f64 *>(dst) = v2f64{a, b};
                        else { dst[0] = a; dst[1] = b; }
                    } else dst[0] = a;
                }
            }
        } else {
            for (int q = tid; q < TN * KP; q += NT) {
                const int k = q / TN, plate = q - k * TN;
                const int64_t n = n0 + plate;
                if (n < N && k < K) X[n * x_sn + (int64_t)k * x_sk] = Z[(DP + k) * SZ + plate];
            }
        }

        // ---- role 2: S += Z_tile * X_tile^T ---------------------------------------------
        {
            const double *zbB = Z + (DP + jt2 * 16 + l15) * SZ + l4;
            const double *zbA = Z + l15 * SZ + l4;
#pragma unroll
            for (int q = 0; q < TN / 4; ++q) {
                const double b = zbB[4 * q];
#pragma unroll
                for (int m = 0; m < R2; ++m) {
                    const int t2 = w2 + 4 * m;
                    if (t2 < T2) {
                        const int it2 = t2 / KT;
                        const double a = zbA[it2 * 16 * SZ + 4 * q];
                        acc2[m] = mfma_f64(a, b, acc2[m]);
                    }
                }
            }
        }
        __syncthreads();
    }

    double *Pb = P + (int64_t)blockIdx.x * (ZR * KP + KP);
#pragma unroll
    for (int m = 0; m < R2; ++m) {
        const int t2 = w2 + 4 * m;
        if (t2 < T2) {
            const int it2 = t2 / KT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = it2 * 16 + l4 + 4 * r;
                Pb[row * KP + jt2 * 16 + l15] = acc2[m][r];
            }
        }
    }
    // sums of x: over the 16 columns a lane group holds, then over the wavefronts of a row tile
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = sx[r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (l15 == 0) SX[w * 16 + l4 + 4 * r] = v;
    }
    __syncthreads();
    if (tid < KP) {
        const int it = tid >> 4, kk = tid & 15;
        double s = 0.0;
        for (int ww = 0; ww < 4; ++ww)
            if (ww % KT == it) s += SX[ww * 16 + kk];
        Pb[ZR * KP + tid] = s;
    }
}

// stats (unpadded) = sum over the workgroups' partial blocks, in fixed order: 16 lanes per output
// element, lane j adds the blocks j, j + 16, ... (four running sums), the lanes are combined by
// xor-shuffles -- the result depends on the number of blocks only.
__global__ void __launch_bounds__(NT)
gshared_reduce_kernel(const double *__restrict__ P, int nb, int64_t blk, int K, int D, int KP,
                      int DP, int with_y, double *__restrict__ stats)
{
    const int total = K + K * K + (with_y ? D * K : 0);
    const int e = blockIdx.x * (NT / 16) + (threadIdx.x >> 4);
    const int j = threadIdx.x & 15;
    const bool ok = e < total;
    int64_t src = 0;
    if (ok) {
        if (e < K) src = (int64_t)(DP + KP) * KP + e;
        else if (e < K + K * K) {
            const int i = (e - K) / K, c = (e - K) - i * K;
            src = (int64_t)(DP + i) * KP + c;
        } else {
            const int d = (e - K - K * K) / K, k = (e - K - K * K) - d * K;
            src = (int64_t)d * KP + k;
        }
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (ok) {
        int b = j;
        for (; b + 48 < nb; b += 64) {
            s0 += P[(int64_t)b * blk + src];
            s1 += P[(int64_t)(b + 16) * blk + src];
            s2 += P[(int64_t)(b + 32) * blk + src];
            s3 += P[(int64_t)(b + 48) * blk + src];
        }
        for (; b < nb; b += 16) s0 += P[(int64_t)b * blk + src];
    }
    double v = (s0 + s1) + (s2 + s3);
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if (ok && j == 0) stats[e] = v;
}

// Message form: x_n = Cov (p0 + m_n) for given rows m_n, one thread per plate, rows staged in LDS.
// Partial block of one workgroup: [K sums of x ; K x K sums of x x^T] (unpadded).
// Dynamic LDS: cov (K x (K+1)) | p0 (K) | min (TR x (K+1)) | xout (TR x (K+1)).
__global__ void __launch_bounds__(NT)
gshared_rows_kernel(int64_t N, int K, int TR, const double *__restrict__ M, int64_t m_sn,
                    int64_t m_sk, const double *__restrict__ p0, const double *__restrict__ cov,
                    double *__restrict__ X, int64_t x_sn, int64_t x_sk, double *__restrict__ P,
                    int64_t ntiles)
{
    extern __shared__ double lds[];
    // (rows of Cov with stride K + 1: the 16 lanes that form <x>_k for k = 0..15 of one plate
    // read 16 different banks)
    double *covs = lds, *p0s = covs + K * (K + 1), *mt = p0s + K, *xt = mt + TR * (K + 1);
    const int tid = threadIdx.x;
    const int KK = K * K;
    for (int e = tid; e < KK; e += NT) covs[(e / K) * (K + 1) + (e % K)] = cov[e];
    for (int e = tid; e < K; e += NT) p0s[e] = p0 ? p0[e] : 0.0;
    double axx[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) axx[q] = 0.0;
    double ax = 0.0;
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * TR;
        for (int e = tid; e < TR * K; e += NT) {
            const int r = e / K, k = e - r * K;
            const int64_t n = n0 + r;
            mt[r * (K + 1) + k] = (n < N) ? M[n * m_sn + (int64_t)k * m_sk] + p0s[k] : 0.0;
        }
        __syncthreads();
        const int rows = (int)((N - n0) < TR ? (N - n0) : TR);
        // one thread per (plate, component): all lanes busy also when a tile holds few plates
        for (int e = tid; e < TR * K; e += NT) {
            const int r = e / K, k = e - r * K;
            double s0 = 0.0, s1 = 0.0;
            if (r < rows) {
                const double *row = mt + r * (K + 1), *cr = covs + k * (K + 1);
                int j = 0;
                for (; j + 1 < K; j += 2) { s0 += cr[j] * row[j]; s1 += cr[j + 1] * row[j + 1]; }
                if (j < K) s0 += cr[j] * row[j];
            }
            xt[r * (K + 1) + k] = s0 + s1;
        }
        __syncthreads();
        for (int e = tid; e < TR * K; e += NT) {
            const int r = e / K, k = e - r * K;
            const int64_t n = n0 + r;
            if (n < N) X[n * x_sn + (int64_t)k * x_sk] = xt[r * (K + 1) + k];
        }
        if (tid < K) {
            double s = 0.0;
            for (int r = 0; r < rows; ++r) s += xt[r * (K + 1) + tid];
            ax += s;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = q * NT + tid;
            if (e < KK) {
                const int i = e / K, j = e - i * K;
                double s0 = 0.0, s1 = 0.0;
                int r = 0;
                for (; r + 1 < rows; r += 2) {
                    s0 += xt[r * (K + 1) + i] * xt[r * (K + 1) + j];
                    s1 += xt[(r + 1) * (K + 1) + i] * xt[(r + 1) * (K + 1) + j];
                }
                if (r < rows) s0 += xt[r * (K + 1) + i] * xt[r * (K + 1) + j];
                axx[q] += s0 + s1;
            }
        }
        __syncthreads();
    }
    double *Pb = P + (int64_t)blockIdx.x * (K + KK);
    if (tid < K) Pb[tid] = ax;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = q * NT + tid;
        if (e < KK) Pb[K + e] = axx[q];
    }
}

__global__ void __launch_bounds__(NT)
gshared_rows_reduce_kernel(const double *__restrict__ P, int nb, int len, double *__restrict__ stats)
{
    const int e = blockIdx.x * NT + threadIdx.x;
    if (e >= len) return;
    double s0 = 0.0, s1 = 0.0;
    int b = 0;
    for (; b + 1 < nb; b += 2) {
        s0 += P[(int64_t)b * len + e];
        s1 += P[(int64_t)(b + 1) * len + e];
    }
    if (b < nb) s0 += P[(int64_t)b * len + e];
    stats[e] = s0 + s1;
}

}  // namespace

extern "C" {

size_t vmp_gaussian_shared_update_workspace_bytes(int32_t D, int32_t K)
{
    // partial blocks of <= 768 workgroups + A, c (Y form: D >= 1; message form: D = 0)
    if (K < 1 || K > 64 || D < 0 || D > 256) return 0;
    if (D == 0) return (size_t)(512 * (K + K * K)) * sizeof(double);
    const size_t DP = 32 * pow2_blocks(D, 32), KP = 16 * pow2_blocks(K, 16);
    return (768 * ((DP + KP) * KP + KP) + KP * DP + KP) * sizeof(double);
}

int32_t vmp_gaussian_shared_update(vmp_ctx *ctx, int64_t N, int32_t K, int32_t D,
                                   const double *Y, int64_t y_sd, int64_t y_sn, const double *B,
                                   int64_t b_sd, int64_t b_sk, const double *m0, int64_t m0_sn,
                                   int64_t m0_sk, const double *p0, const double *cov, double *x,
                                   int64_t x_sn, int64_t x_sk, double *stats, void *workspace,
                                   size_t workspace_bytes)
{
    VMP_REQUIRE(ctx, ctx != nullptr, VMP_ERR_INVALID, "null context");
    VMP_REQUIRE(ctx, N >= 0 && K >= 1 && K <= 64, VMP_ERR_UNSUPPORTED,
                "vmp_gaussian_shared_update: K = %d (built for 1..64)", K);
    VMP_REQUIRE(ctx, cov && x && stats && workspace, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, (Y != nullptr) != (m0 != nullptr), VMP_ERR_INVALID,
                "exactly one of Y (with B) and m0 must be given");
    VMP_REQUIRE(ctx, workspace_bytes >= vmp_gaussian_shared_update_workspace_bytes(Y ? D : 0, K),
                VMP_ERR_INVALID, "workspace too small");
    VMP_FLUSH_SMALL(ctx);
    hipStream_t s = ctx->stream;
    double *ws = reinterpret_cast<double *>(workspace);
    if (Y) {
        VMP_REQUIRE(ctx, B != nullptr && D >= 1 && D <= 256, VMP_ERR_UNSUPPORTED,
                    "vmp_gaussian_shared_update: D = %d (built for 1..256)", D);
        VMP_REQUIRE(ctx, y_sn == 1 || y_sd == 1, VMP_ERR_UNSUPPORTED,
                    "vmp_gaussian_shared_update: Y needs a unit stride along its plates or its rows");
        const int DP = 32 * pow2_blocks(D, 32), KP = 16 * pow2_blocks(K, 16);
        const int DB = DP / 32, KT = KP / 16;
        double *Apad = ws, *cpad = Apad + (int64_t)KP * DP, *P = cpad + KP;
        const size_t plds = (size_t)(K * K + 64 * (K + 1)) * sizeof(double);
        static bool praised[64] = {false};
        if (plds > 48 * 1024 && !praised[ctx->device & 63]) {
            VMP_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)gshared_prepare_kernel,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   160 * 1024));
            praised[ctx->device & 63] = true;
        }
        hipLaunchKernelGGL(gshared_prepare_kernel, dim3(1), dim3(NT), plds, s, K, D, KP, DP, cov, B,
                           b_sd, b_sk, p0, Apad, cpad);
        VMP_HIP_CHECK(ctx, hipGetLastError());
        const int64_t ntiles = (N + TN - 1) / TN;
        const int64_t blk = (int64_t)(DP + KP) * KP + KP;
        int64_t g = ntiles;
        const int per_cu = vmp_tune_get("gshared_wgs_per_cu", DB * KT <= 2 ? 3 : 2);
        const int64_t gmax = (int64_t)ctx->num_cu * per_cu;
        if (g > gmax) g = gmax;
        if (g > 768) g = 768;
        if (g < 1) g = 1;
        // 16-byte accesses where every address they touch is 16-byte aligned
        const int yvec = ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) &&
                         ((y_sn == 1 ? y_sd : y_sn) % 2 == 0);
        const int xvec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (x_sn % 2 == 0);
#define VMP_GS(db, kt)                                                                          \
    if (DB == db && KT == kt)                                                                   \
        hipLaunchKernelGGL((gshared_pass_kernel<db, kt>), dim3((unsigned)g), dim3(NT), 0, s, Y, \
                           y_sd, y_sn, N, D, K, Apad, cpad, x, x_sn, x_sk, P, ntiles, yvec, xvec); \
    else
        VMP_GS(1, 1) VMP_GS(1, 2) VMP_GS(1, 4) VMP_GS(2, 1) VMP_GS(2, 2) VMP_GS(2, 4)
        VMP_GS(4, 1) VMP_GS(4, 2) VMP_GS(4, 4) VMP_GS(8, 1) VMP_GS(8, 2) VMP_GS(8, 4)
        {
            VMP_SET_ERR(ctx, "no kernel instance for DB=%d KT=%d", DB, KT);
            return VMP_ERR_UNSUPPORTED;
        }
#undef VMP_GS
        VMP_HIP_CHECK(ctx, hipGetLastError());
        const int total = K + K * K + D * K;
        hipLaunchKernelGGL(gshared_reduce_kernel, dim3((total + NT / 16 - 1) / (NT / 16)), dim3(NT),
                           0, s, P, (int)g, blk, K, D, KP, DP, 1, stats);
        VMP_HIP_CHECK(ctx, hipGetLastError());
        return VMP_OK;
    }
    int TR = K <= 16 ? 256 : (K <= 32 ? 128 : 64);
    while (TR > 64 && TR / 2 >= N) TR /= 2;          // (a few plates: a replicated node's rows)
    const int64_t ntiles = (N + TR - 1) / TR;
    int64_t g = ntiles;
    const int64_t gmax = (int64_t)ctx->num_cu * 2;
    if (g > gmax) g = gmax;
    if (g > 512) g = 512;
    if (g < 1) g = 1;
    const size_t lds = (size_t)(K * (K + 1) + K + 2 * TR * (K + 1)) * sizeof(double);
    static bool raised[64] = {false};
    const int dev = ctx->device & 63;
    if (!raised[dev] && lds > 48 * 1024) {
        VMP_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)gshared_rows_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024));
        raised[dev] = true;
    }
    const int len = K + K * K;
    double *P = (g == 1) ? stats : ws;
    hipLaunchKernelGGL(gshared_rows_kernel, dim3((unsigned)g), dim3(NT), lds, s, N, K, TR, m0, m0_sn,
                       m0_sk, p0, cov, x, x_sn, x_sk, P, ntiles);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    if (g > 1) {
        hipLaunchKernelGGL(gshared_rows_reduce_kernel, dim3((len + NT - 1) / NT), dim3(NT), 0, s, P,
                           (int)g, len, stats);
        VMP_HIP_CHECK(ctx, hipGetLastError());
    }
    return VMP_OK;
}

}  // extern "C"
