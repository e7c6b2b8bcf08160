// vmp_pca.hip -- fused probabilistic-PCA / factor-analysis VB block for gfx950.
//
// Two forms of X.update()'s plate work are built:
//   * vmp_pca_xpass  (default): X = A Y only; the messages to W come from the
//     constant Gram matrix G = Y Y^T (HBM-bound: read Y once, write <x>);
//   * vmp_pca_pass   (streaming statistics): additionally accumulates
//     sum y<x>^T and sum <x><x>^T on the fly (fp64-MFMA-bound) -- also the kernel
//     that builds G and the statistics of a given X.
//
// Replaces, for the model block of bayespy/demos/pca.py:22-61 with a scalar
// observation mask, the NumPy call sites E1-E11, E14, E16, E17 of SURVEY.md 2.2:
//   dot.py:355,403,581 (SumMultiply moments + messages),
//   gaussian.py:628-635,649-706,2344-2369 (GaussianARD / WrapToGaussianGamma),
//   gamma.py:116-148, expfamily.py:400-480 (lower bound), utils/linalg.py:31-223.
//
// Data layout in HBM (all fp64):
//   Y  (D, N) row-major, leading dimension ldy  -- N (the sharded observation
//       plate) is the contiguous axis, exactly the reference's plates (D, N);
//   X  (K, N) row-major, leading dimension ldx  -- <x_n>, structure-of-arrays
//       along the plate so stores are coalesced (host view transposes to the
//       reference's (1, N, K));
//   state: one block of doubles, layout = vmp_pca_layout (include/vmp_hip.h).
//
// The streaming pass keeps a (DP+KP) x 32-column tile Z = [Y_tile ; X_tile] in
// LDS (row stride 34 doubles => both MFMA operand access patterns below are
// bank-conflict free for ds_read_b64) and issues v_mfma_f64_16x16x4_f64 for
//   role 1:  X_tile = A * Y_tile            (contraction over d)
//   role 2:  S     += Z_tile * X_tile^T     (contraction over n)
// with the S accumulators resident in registers for the whole kernel.
#include "vmp_common.h"

namespace {

constexpr int TN = 32;        // columns (plate elements) per tile
constexpr int SZ = TN + 2;    // LDS row stride in doubles (== 2 mod 32)
constexpr int NT = 256;       // threads per workgroup (4 wavefronts)
constexpr int MAX_KP = 64;
constexpr int MAX_DP = 256;

inline int pow2_blocks(int x, int unit)
{
    int b = (x + unit - 1) / unit;
    int p = 1;
    while (p < b) p <<= 1;
    return p;
}

inline void fill_layout(int D, int K, vmp_pca_layout *L)
{
    const int64_t DP = 32 * pow2_blocks(D, 32);
    const int64_t KP = 16 * pow2_blocks(K, 16);
    int64_t o = 0;
    L->DP = DP;
    L->KP = KP;
    L->off_S = o;      L->len_S = (DP + KP) * KP; o += L->len_S;
    L->off_Syy = o;    o += 8;
    L->off_tau = o;    o += 8;
    L->off_alpha = o;  o += 4 * KP;
    L->off_W = o;      o += (int64_t)D * KP;
    L->off_CW = o;     o += KP * KP;
    L->off_Sww = o;    o += KP * KP;
    L->off_CX = o;     o += KP * KP;
    L->off_A = o;      o += KP * DP;
    L->off_G = o;      o += DP * DP;
    L->off_scal = o;   o += 8;
    L->off_L = o;      o += 8;
    L->off_mu = o;     o += (int64_t)D * KP;
    L->off_mstat = o;  o += 2 * KP;
    L->total = (o + 7) / 8 * 8;
}

__device__ inline v4f64 mfma_f64(double a, double b, v4f64 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// The streaming pass.
//   DB = DP/32 in {1,2,4,8}, KT = KP/16 in {1,2,4}.
//   COMPUTE_X: true  -> X.update() (x = A y, write X, accumulate S)
//              false -> statistics of a given X (initialize_from_value)
// ---------------------------------------------------------------------------
template <int DB, int KT, bool COMPUTE_X>
__global__ void __launch_bounds__(NT, (DB >= 8 || DB * KT > 8) ? 1 : 2)
pca_pass_kernel(const double *__restrict__ Y, int64_t ldy, int64_t N, int D, int K,
                const double *__restrict__ Apad, double *__restrict__ X, int64_t ldx,
                double *__restrict__ P, int64_t ntiles)
{
    constexpr int DP = 32 * DB, KP = 16 * KT, ZR = DP + KP;
    constexpr int YP = DP / 16;          // Y load passes of 16 rows
    constexpr int XP = KP / 16;          // X load passes (COMPUTE_X == false)
    constexpr int KS1 = DP / 4;          // role-1 MFMA k-steps
    constexpr int T1 = KT * (TN / 16);   // role-1 output tiles
    constexpr int R1 = (T1 + 3) / 4;
    constexpr int T2 = (2 * DB + KT) * KT;  // role-2 output tiles
    constexpr int R2 = (T2 + 3) / 4;

    __shared__ double Z[ZR * SZ];

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int l15 = l & 15, l4 = l >> 4;
    const int lrow = tid >> 4;         // row inside a 16-row load pass
    const int lcol = (tid & 15) * 2;   // first of this thread's two columns
    const int it1 = w % KT;            // role-1 row tile (k) of this wave
    const int jt2 = w % KT;            // role-2 column tile (k) of this wave

    // A operand fragments of role 1: lane holds A[it1*16 + l15][d(q, l4)],
    // d(q, kk) = 32*(q/8) + (q%8) + 8*kk  (the four d-slices of one MFMA are 8
    // rows apart so the matching Y reads from LDS hit disjoint banks).
    double afrag[COMPUTE_X ? KS1 : 1];
    if (COMPUTE_X) {
        const double *arow = Apad + (int64_t)(it1 * 16 + l15) * DP;
#pragma unroll
        for (int q = 0; q < KS1; ++q) afrag[q] = arow[32 * (q >> 3) + (q & 7) + 8 * l4];
    }

    v4f64 acc2[R2];
#pragma unroll
    for (int m = 0; m < R2; ++m) acc2[m] = v4f64{0.0, 0.0, 0.0, 0.0};

    v2f64 yreg[YP];

    auto load_tile = [&](int64_t tile) {
        const int64_t n = tile * TN + lcol;
#pragma unroll
        for (int p = 0; p < YP; ++p) {
            const int row = p * 16 + lrow;
            v2f64 v = v2f64{0.0, 0.0};
            if (row < D) {
                const double *src = Y + (int64_t)row * ldy + n;
                if (n + 1 < N) v = *reinterpret_cast<const v2f64 *>(src);
                else if (n < N) v.x = src[0];
            }
            yreg[p] = v;
        }
    };

    int64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);

    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * TN;
        // ---- stage the Y tile (and, for given-X statistics, the X tile) ----
#pragma unroll
        for (int p = 0; p < YP; ++p)
            *reinterpret_cast<v2f64 *>(&Z[(p * 16 + lrow) * SZ + lcol]) = yreg[p];
        if (!COMPUTE_X) {
#pragma unroll
            for (int p = 0; p < XP; ++p) {
                const int row = p * 16 + lrow;
                const int64_t n = n0 + lcol;
                v2f64 v = v2f64{0.0, 0.0};
                if (row < K) {
                    const double *src = X + (int64_t)row * ldx + n;
                    if (n + 1 < N) { v.x = src[0]; v.y = src[1]; }
                    else if (n < N) v.x = src[0];
                }
                *reinterpret_cast<v2f64 *>(&Z[(DP + row) * SZ + lcol]) = v;
            }
        }
        __syncthreads();

        // ---- prefetch the next Y tile into registers (in flight during MFMAs)
        const int64_t next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);

        // ---- role 1: X_tile = A * Y_tile --------------------------------------
        if (COMPUTE_X) {
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
                    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15,
                    // row = (lane>>4) + 4*reg.
                    const int n = jt * 16 + l15;
                    const bool nok = (n0 + n) < N;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = it1 * 16 + l4 + 4 * r;
                        Z[(DP + k) * SZ + n] = c0[r];
                        if (nok && k < K) X[(int64_t)k * ldx + n0 + n] = c0[r];
                    }
                }
            }
            __syncthreads();
        }

        // ---- role 2: S += Z_tile * X_tile^T  (n-slices of one MFMA adjacent) ---
        {
            const double *zbB = Z + (DP + jt2 * 16 + l15) * SZ + l4;
            const double *zbA = Z + l15 * SZ + l4;
#pragma unroll
            for (int q = 0; q < TN / 4; ++q) {
                const double b = zbB[4 * q];
#pragma unroll
                for (int m = 0; m < R2; ++m) {
                    const int t2 = w + 4 * m;
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

    // ---- per-workgroup partial statistics (reduced in fixed order later) -----
    double *Pb = P + (int64_t)blockIdx.x * (ZR * KP);
#pragma unroll
    for (int m = 0; m < R2; ++m) {
        const int t2 = w + 4 * m;
        if (t2 < T2) {
            const int it2 = t2 / KT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = it2 * 16 + l4 + 4 * r;
                Pb[row * KP + jt2 * 16 + l15] = acc2[m][r];
            }
        }
    }
}

// S[e] = sum_b P[b][e] in fixed order b = 0..nb-1 (deterministic).
__global__ void __launch_bounds__(NT)
reduce_partials_kernel(const double *__restrict__ P, int nb, int len, double *__restrict__ S)
{
    const int e = blockIdx.x * NT + threadIdx.x;
    if (e >= len) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = 0;
    for (; b + 3 < nb; b += 4) {
        s0 += P[(int64_t)b * len + e];
        s1 += P[(int64_t)(b + 1) * len + e];
        s2 += P[(int64_t)(b + 2) * len + e];
        s3 += P[(int64_t)(b + 3) * len + e];
    }
    for (; b < nb; ++b) s0 += P[(int64_t)b * len + e];
    S[e] = (s0 + s1) + (s2 + s3);
}

// partial[b] = sum over this workgroup's grid-stride share of y^2.
__global__ void __launch_bounds__(NT)
sumsq_kernel(const double *__restrict__ Y, int64_t ldy, int64_t N, int D,
             double *__restrict__ partial)
{
    __shared__ double red[NT / 64];
    double s = 0.0;
    const int64_t npairs = (N + 1) / 2;
    const int64_t total = npairs * D;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * NT) {
        const int64_t d = i / npairs;
        const int64_t n = (i - d * npairs) * 2;
        const double *src = Y + d * ldy + n;
        if (n + 1 < N) {
            const v2f64 v = *reinterpret_cast<const v2f64 *>(src);
            s += v.x * v.x + v.y * v.y;
        } else {
            s += src[0] * src[0];
        }
    }
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(NT)
sum_partials_kernel(const double *__restrict__ partial, int n, double *__restrict__ out)
{
    __shared__ double red[NT / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += NT) s += partial[i];
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

// ---------------------------------------------------------------------------
// X.update(), plate half, Gram form: X = A Y, nothing else.  HBM-bound.
//
// With a scalar mask the posterior mean of every x_n is the SAME linear map of
// y_n, so the messages to W collapse onto the constant Gram matrix G = Y Y^T:
//     sum_n y_n <x_n>^T   = G A^T,      sum_n <x_n><x_n>^T = A G A^T
// (SURVEY.md 2.2 notes the same collapse for E2).  The per-iteration plate work
// is then exactly SURVEY.md 8(d)'s algorithmic traffic: read Y once, write <x>.
//
// Each wavefront owns 32-column tiles and never touches LDS for Y: lane
// (l15, l4) loads the 16 bytes Y[4q + l4][n0 + 2 l15 .. +1], whose two halves are
// the B operands of two MFMAs (even / odd columns), and stores 16 bytes of X per
// accumulator register.  A (K x D, L2 resident) sits in LDS in fragment order.
// ---------------------------------------------------------------------------
//
// GUARD = false: only full 32-column tiles, D == DP and K == KP -- no predication
// anywhere in the loop; GUARD = true handles ragged D, K and the last partial tile.
//
// LAY bit 0: Y is the tile-major copy made by vmp_pca_tile_y ([tile][DP][32] doubles, zero
// padded): a wavefront then streams ONE contiguous 32 KB span per tile and every load
// instruction covers 1 KB of consecutive addresses (lane l reads bytes 16 l .. 16 l + 15 of
// k-step q's 1 KB) instead of four 256-byte row segments that lie ldy*8 bytes apart.
// LAY bit 1: X is tile-major as well ([tile][KP][32]); every store instruction writes 1 KB.
// MF = 1 (round 3): the product on v_mfma_f64_4x4x4_4b_f64 -- four independent 4 x 4 x 4 blocks
// per instruction, 1.4-1.5x the flop rate of the 16x16x4 form on this chip (tools/mfma4_lab.hip).
// The blocks are the four 4-column groups of a 16-column half tile, so the B operand is the SAME
// 16 bytes of Y as before (lane = column pair + 16 k); the A operand is the fragment of the
// component group R (rows 16 it + 4 R + i of A) replicated over the blocks -- element
// 16 (l >> 4) + 4 R + (l & 3) of the (it, q) fragment in LDS -- and a result register holds rows
// 4 R + (l >> 4), i.e. exactly the rows register r = R held before: the stores are unchanged.
// The A fragments take KT DB 4 KB of LDS: 128 KB at D = 256, K = 64 (one workgroup per CU), 64 KB
// at D = 256, K <= 32 or D = 128, K = 64 (two).  The register budget asked of the compiler follows
// that limit; the pass itself is as fast at one workgroup per CU as at four
// (profiles/r03/xpass_lab_n1e7.txt: one wavefront per SIMD holds 8 KB of loads in flight).
constexpr int xpass_occ(int DB, int KT, int OCC)
{
    return KT * DB * 4096 > 80 * 1024 ? 1 : (KT * DB * 4096 > 53 * 1024 ? (OCC < 2 ? OCC : 2) : OCC);
}

// IL (round 4 experiment, tune key "pca_interleave"): ONE array [tile][DP + KP][32] -- the <x> rows of
// a tile directly behind its data rows -- so that the relative placement of the read and the write
// stream is fixed by construction instead of by two allocations.
template <int DB, int KT, bool GUARD, int OCC, int NTM = 0, int LAY = 0, int MF = 0, bool IL = false>
__global__ void __launch_bounds__(NT, xpass_occ(DB, KT, OCC))
pca_xpass_kernel(const double *__restrict__ Y, int64_t ldy, int64_t N, int D, int K,
                 const double *__restrict__ Apad, double *__restrict__ X, int64_t ldx,
                 int64_t tile0, int64_t tile1)
{
    constexpr int DP = 32 * DB;
    constexpr int KS = DP / 4;                 // MFMA k-steps per tile
    constexpr int NCH = (DB < 2) ? 2 : DB;     // chunks per tile (even)
    constexpr int CH = KS / NCH;               // k-steps per chunk
    constexpr bool YT = (LAY & 1) != 0, XT = (LAY & 2) != 0;
    static_assert(!(GUARD && LAY != 0), "tile-major arrays are padded: no guarded instance");

    __shared__ double Af[KT * KS * 64];        // A fragments: [(it*KS + q)*64 + lane]

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int l15 = l & 15, l4 = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id, scalar

    for (int e = tid; e < KT * KS * 64; e += NT) {
        const int lane = e & 63, fq = e >> 6;
        const int it = fq / KS, q = fq - it * KS;
        Af[e] = Apad[(int64_t)(it * 16 + (lane & 15)) * DP + 4 * q + (lane >> 4)];
    }
    __syncthreads();

    const double *Afl = MF ? Af + 16 * l4 + (l & 3) : Af + l;
    // per-lane byte offsets (32-bit; the host checks 3*ld*8 + 256 < 2^32): the
    // wave-uniform part of every address stays in SGPRs
    const uint32_t yoff = (uint32_t)(((int64_t)l4 * ldy + 2 * l15) * 8);
    const uint32_t xoff = (uint32_t)(((int64_t)l4 * ldx + 2 * l15) * 8);

    v2f64 buf[2][CH];
    const int64_t stride = (int64_t)gridDim.x * 4;

    auto issue = [&](int64_t tile, int c, v2f64 *dst) {
        if (YT) {
            const char *base = reinterpret_cast<const char *>(Y)
                               + ((tile * ((DP + (IL ? 16 * KT : 0)) * TN)
                                   + (int64_t)(c * CH) * (4 * TN)) << 3);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const v2f64 *src = reinterpret_cast<const v2f64 *>(base + i * (4 * TN * 8) + l * 16);
                dst[i] = (NTM & 1) ? __builtin_nontemporal_load(src) : *src;
            }
            return;
        }
        const char *base = reinterpret_cast<const char *>(Y)
                           + ((tile * TN + (int64_t)(4 * c * CH) * ldy) << 3);
        if (!GUARD) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const v2f64 *src = reinterpret_cast<const v2f64 *>(
                    base + ((int64_t)(4 * i) * ldy << 3) + yoff);
                dst[i] = (NTM & 1) ? __builtin_nontemporal_load(src) : *src;
            }
        } else {
            const int64_t n = tile * TN + 2 * l15;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int row = 4 * (c * CH + i) + l4;
                v2f64 v = v2f64{0.0, 0.0};
                if (row < D) {
                    const double *src = reinterpret_cast<const double *>(
                        base + ((int64_t)(4 * i) * ldy << 3) + yoff);
                    if (n + 1 < N) v = *reinterpret_cast<const v2f64 *>(src);
                    else if (n < N) v.x = src[0];
                }
                dst[i] = v;
            }
        }
    };

    // STG (round 3, measured and left off): tile t starts at chunk t % NCH and wraps, so that
    // wavefronts advancing in step do not sit exactly 32 KB apart.  A bare stream kernel with this
    // tile pattern gains 6 % from it at one workgroup per CU (tools/microbench.hip, "tile
    // pattern": 6.0 -> 6.4 TB/s); this kernel does not (profiles/r03/xpass_lab_n1e7.txt: 5.41
    // against 5.51 TB/s on the same box), and the results would no longer be bit-identical to the
    // in-order sum.
    constexpr bool STG = false;
    auto first_chunk = [&](int64_t t) { return STG ? (int)(t & (NCH - 1)) : 0; };
    int64_t tile = tile0 + (int64_t)blockIdx.x * 4 + w;
    if (tile < tile1) issue(tile, first_chunk(tile), buf[0]);

    for (; tile < tile1; tile += stride) {
        v4f64 acc[KT][2];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            acc[it][0] = v4f64{0.0, 0.0, 0.0, 0.0};
            acc[it][1] = v4f64{0.0, 0.0, 0.0, 0.0};
        }
        const int rot = first_chunk(tile);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // prefetch the next chunk (the next tile's first chunk at the end) so
            // that its loads are in flight while this chunk's MFMAs run
            const int cc = STG ? ((c + rot) & (NCH - 1)) : c;
            if (c + 1 < NCH) issue(tile, STG ? ((c + 1 + rot) & (NCH - 1)) : c + 1, buf[(c + 1) & 1]);
            else if (tile + stride < tile1) issue(tile + stride, first_chunk(tile + stride), buf[(c + 1) & 1]);
            // compiler-only barrier: keeps the fragment reads of A inside the loop
            // (otherwise all KT*KS of them are hoisted into spilled registers)
            asm volatile("" ::: "memory");
            const double *Afc = Afl + cc * (CH * 64);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const v2f64 b = buf[c & 1][i];
#pragma unroll
                for (int it = 0; it < KT; ++it) {
                    if (MF) {
#pragma unroll
                        for (int R = 0; R < 4; ++R) {
                            const double a = Afc[(it * KS + i) * 64 + 4 * R];
                            acc[it][0][R] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b.x, acc[it][0][R], 0, 0, 0);
                            acc[it][1][R] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b.y, acc[it][1][R], 0, 0, 0);
                        }
                    } else {
                        const double a = Afc[(it * KS + i) * 64];
                        acc[it][0] = mfma_f64(a, b.x, acc[it][0]);
                        acc[it][1] = mfma_f64(a, b.y, acc[it][1]);
                    }
                }
            }
        }
        // C/D layout: col = lane&15 -> column pair, row = (lane>>4) + 4*reg -> k
        char *xbase = reinterpret_cast<char *>(X) + ((tile * TN) << 3);
        if (XT) {
            char *xt = reinterpret_cast<char *>(X)
                       + ((tile * (((IL ? DP : 0) + 16 * KT) * TN)) << 3) + l * 16;
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v2f64 *dstp = reinterpret_cast<v2f64 *>(xt + (it * 16 + 4 * r) * (TN * 8));
                    const v2f64 val = v2f64{acc[it][0][r], acc[it][1][r]};
                    if (NTM & 2) __builtin_nontemporal_store(val, dstp);
                    else *dstp = val;
                }
        } else if (!GUARD) {
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v2f64 *dstp = reinterpret_cast<v2f64 *>(
                        xbase + ((int64_t)(it * 16 + 4 * r) * ldx << 3) + xoff);
                    const v2f64 val = v2f64{acc[it][0][r], acc[it][1][r]};
                    if (NTM & 2) __builtin_nontemporal_store(val, dstp);
                    else *dstp = val;
                }
        } else {
            const int64_t n = tile * TN + 2 * l15;
#pragma unroll
            for (int it = 0; it < KT; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = it * 16 + l4 + 4 * r;
                    if (k < K) {
                        double *dst = reinterpret_cast<double *>(
                            xbase + ((int64_t)(it * 16 + 4 * r) * ldx << 3) + xoff);
                        if (n + 1 < N)
                            *reinterpret_cast<v2f64 *>(dst) = v2f64{acc[it][0][r], acc[it][1][r]};
                        else if (n < N)
                            dst[0] = acc[it][0][r];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Set-up and Gram-form statistics kernels (the replicated-node updates live in
// vmp_pca_small.hip).
// ---------------------------------------------------------------------------
constexpr int LD = MAX_KP + 1;

__global__ void __launch_bounds__(NT)
pca_init_state_kernel(vmp_pca_layout L, int K, double a0t, double b0t, double a0a, double b0a,
                      double *st)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        st[L.off_tau + 0] = a0t;
        st[L.off_tau + 1] = b0t;
        st[L.off_tau + 2] = a0t / b0t;
        st[L.off_tau + 3] = vmp_digamma(a0t) - log(b0t);
    }
    for (int k = tid; k < K; k += NT) {
        st[L.off_alpha + 0 * L.KP + k] = a0a;
        st[L.off_alpha + 1 * L.KP + k] = b0a;
        st[L.off_alpha + 2 * L.KP + k] = a0a / b0a;
        st[L.off_alpha + 3 * L.KP + k] = vmp_digamma(a0a) - log(b0a);
    }
}

// Messages to W from the Gram matrix: workgroup b owns GR rows of G.
//   Syx[r][k]      = sum_d G[r][d] A[k][d]
//   Pxx_b[k][k']   = sum_{r in b} A[k][r] Syx[r][k']     (reduced in fixed order later)
constexpr int GR = 8;
__global__ void __launch_bounds__(NT)
pca_gram_stats_kernel(vmp_pca_layout L, int D, int K, const double *__restrict__ st_in,
                      double *__restrict__ st, double *__restrict__ P)
{
    extern __shared__ double lds[];
    const int DP = (int)L.DP, KP = (int)L.KP;
    const int LA = DP + 1;
    double *As = lds;                    // KP x LA
    double *Gs = As + KP * LA;           // GR x DP
    double *Ss = Gs + GR * DP;           // GR x LD
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * GR;
    const double *A = st_in + L.off_A;
    const double *G = st_in + L.off_G;
    for (int e = tid; e < K * DP; e += NT) {
        const int k = e / DP, d = e - k * DP;
        As[k * LA + d] = A[(int64_t)k * DP + d];
    }
    for (int e = tid; e < GR * DP; e += NT) Gs[e] = G[(int64_t)r0 * DP + e];
    __syncthreads();
    for (int e = tid; e < GR * K; e += NT) {
        const int r = e / K, k = e - r * K;
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += Gs[r * DP + d] * As[k * LA + d];
        Ss[r * LD + k] = s;
        if (r0 + r < D) st[L.off_S + (int64_t)(r0 + r) * KP + k] = s;
    }
    __syncthreads();
    double *Pb = P + (int64_t)blockIdx.x * KP * KP;
    for (int e = tid; e < KP * KP; e += NT) {
        const int k = e / KP, k2 = e - k * KP;
        double s = 0.0;
        if (k < K && k2 < K) {
#pragma unroll
            for (int r = 0; r < GR; ++r) s += As[k * LA + r0 + r] * Ss[r * LD + k2];
        }
        Pb[e] = s;
    }
}

// G[:, j0:j0+Kx] <- block of the statistics kernel's output (set-up only).
__global__ void __launch_bounds__(NT)
pca_gram_copy_kernel(const double *__restrict__ Stmp, int KPx, int D, int Kx, int j0, int DP,
                     double *__restrict__ G)
{
    const int e = blockIdx.x * NT + threadIdx.x;
    if (e >= D * Kx) return;
    const int i = e / Kx, j = e - i * Kx;
    G[(int64_t)i * DP + j0 + j] = Stmp[(int64_t)i * KPx + j];
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
int env_int(const char *name, int dflt, int lo, int hi)
{
    const char *e = getenv(name);
    int v = e ? atoi(e) : dflt;
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

int wgs_per_cu()
{
    static int v = -1;
    if (v < 0) v = env_int("VMP_PCA_WGS_PER_CU", 2, 1, 8);
    return v;
}

int pca_interleave() { return vmp_tune_get("pca_interleave", 0); }

int xpass_wgs_per_cu()
{
    static int v = -1;
    if (v < 0) v = env_int("VMP_PCA_XPASS_WGS_PER_CU", 3, 1, 8);
    return vmp_tune_get("xpass_wgs_per_cu", v);
}

int xpass_occupancy()
{
    static int v = -1;
    if (v < 0) v = env_int("VMP_PCA_XPASS_OCC", 3, 2, 3);
    return vmp_tune_get("xpass_occ", v);
}

int64_t max_grid(vmp_ctx *ctx) { return (int64_t)ctx->num_cu * wgs_per_cu(); }

int64_t partial_len(const vmp_pca_layout &L) { return (L.DP + MAX_KP) * MAX_KP; }

template <int DB, int KT>
void launch_fused(bool compute_x, dim3 grid, hipStream_t s, const double *Y, int64_t ldy,
                  int64_t N, int D, int K, const double *A, double *X, int64_t ldx, double *P,
                  int64_t ntiles)
{
    if (compute_x)
        hipLaunchKernelGGL((pca_pass_kernel<DB, KT, true>), grid, dim3(NT), 0, s, Y, ldy, N, D, K,
                           A, X, ldx, P, ntiles);
    else
        hipLaunchKernelGGL((pca_pass_kernel<DB, KT, false>), grid, dim3(NT), 0, s, Y, ldy, N, D,
                           K, A, X, ldx, P, ntiles);
}

#define VMP_FOR_EACH_INSTANCE(M)                                       \
    M(1, 1) M(2, 1) M(4, 1) M(8, 1) M(1, 2) M(2, 2) M(4, 2) M(8, 2)    \
    M(1, 4) M(2, 4) M(4, 4) M(8, 4)

int32_t check_pass_args(vmp_ctx *ctx, const void *Y, const void *X, const void *state,
                        const void *workspace, int64_t ldy, int64_t ldx, int64_t N, int D, int K)
{
    VMP_REQUIRE(ctx, ctx != nullptr, VMP_ERR_INVALID, "null context");
    VMP_REQUIRE(ctx, Y && X && state && workspace, VMP_ERR_INVALID, "null pointer argument");
    VMP_REQUIRE(ctx, D >= 1 && K >= 1 && N >= 0, VMP_ERR_INVALID, "bad dims D=%d K=%d N=%lld", D,
                K, (long long)N);
    VMP_REQUIRE(ctx, D <= MAX_DP && K <= MAX_KP, VMP_ERR_UNSUPPORTED,
                "fused PCA block supports D <= %d, K <= %d (got D=%d, K=%d)", MAX_DP, MAX_KP, D, K);
    VMP_REQUIRE(ctx, ldy >= N && ldx >= N, VMP_ERR_INVALID, "leading dimension smaller than N");
    VMP_REQUIRE(ctx, (ldy % 2) == 0 && ((uintptr_t)Y % 16) == 0, VMP_ERR_INVALID,
                "Y must be 16-byte aligned with an even leading dimension (ldy=%lld)",
                (long long)ldy);
    VMP_REQUIRE(ctx, (ldx % 2) == 0 && ((uintptr_t)X % 16) == 0, VMP_ERR_INVALID,
                "X must be 16-byte aligned with an even leading dimension (ldx=%lld)",
                (long long)ldx);
    VMP_REQUIRE(ctx, ldy < (int64_t)170000000 && ldx < (int64_t)170000000, VMP_ERR_UNSUPPORTED,
                "leading dimension >= 1.7e8 elements per shard is not supported "
                "(32-bit lane offsets); shard the plate over more ranks");
    return VMP_OK;
}

// The statistics kernel: out_S ((DPy+KPx) x KPx) <- [Y Xs^T ; Xs Xs^T] (or the
// fused compute_x form).  Xs has Kx rows.
int32_t run_stats(vmp_ctx *ctx, bool compute_x, const double *Y, int64_t ldy, int64_t N, int D,
                  int Kx, double *Xs, int64_t ldx, const double *A, double *out_S, double *P,
                  bool timed)
{
    vmp_pca_layout Lx;
    fill_layout(D, Kx, &Lx);
    const int DB = (int)(Lx.DP / 32), KT = (int)(Lx.KP / 16);
    const int64_t ntiles = (N + TN - 1) / TN;
    int64_t g = ntiles < max_grid(ctx) ? ntiles : max_grid(ctx);
    if (g < 1) g = 1;
    hipStream_t s = ctx->stream;
    dim3 grid((unsigned)g);
    hipEvent_t *ev = (timed && ctx->timing) ? vmp_next_events(ctx) : nullptr;
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[0], s));
#define VMP_CASE(db, kt)                                                                     \
    if (DB == db && KT == kt)                                                                \
        launch_fused<db, kt>(compute_x, grid, s, Y, ldy, N, D, Kx, A, Xs, ldx, P, ntiles);   \
    else
    VMP_FOR_EACH_INSTANCE(VMP_CASE)
    {
        VMP_SET_ERR(ctx, "no kernel instance for DB=%d KT=%d", DB, KT);
        return VMP_ERR_UNSUPPORTED;
    }
#undef VMP_CASE
    VMP_HIP_CHECK(ctx, hipGetLastError());
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[1], s));
    const int len = (int)Lx.len_S;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((len + NT - 1) / NT), dim3(NT), 0, s, P,
                       (int)g, len, out_S);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[2], s));
    return VMP_OK;
}

// Plate stream of the context, created on first use.  VMP_PCA_RESERVE_CUS=n (default 0)
// creates it with a CU mask that keeps n compute units free for the one-workgroup
// replicated-node kernels of the main stream.  Measured on MI355X (profiles/r01/README):
// the mask unbalances the persistent streaming grid across XCDs (pass +9..14 % for n = 8)
// while the step at shard size gains only 3-5 %, so the default is an unmasked non-blocking
// stream: the small kernels that fit beside the streaming workgroups overlap, the others
// run between two passes.
int32_t ensure_plate_stream(vmp_ctx *ctx)
{
    if (ctx->xs) return VMP_OK;
    static const int reserve = env_int("VMP_PCA_RESERVE_CUS", 0, 0, 64);
    hipStream_t xs = nullptr;
    const int ncu = ctx->num_cu;
    ctx->xs_cus = ncu;
    if (reserve > 0 && ncu > 2 * reserve) {
        const int words = (ncu + 31) / 32;
        uint32_t mask[64];
        for (int w = 0; w < words && w < 64; ++w) mask[w] = 0;
        for (int c = 0; c < ncu; ++c) mask[c >> 5] |= (1u << (c & 31));
        const int step = ncu / reserve;
        for (int r = 0; r < reserve; ++r) {
            const int c = r * step + step / 2;
            mask[c >> 5] &= ~(1u << (c & 31));
        }
        if (hipExtStreamCreateWithCUMask(&xs, (uint32_t)words, mask) != hipSuccess) {
            (void)hipGetLastError();
            xs = nullptr;
        } else {
            ctx->xs_cus = ncu - reserve;
        }
    }
    if (!xs) VMP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&xs, hipStreamNonBlocking));
    ctx->xs = xs;
    VMP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_xfork, hipEventDisableTiming));
    VMP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_xdone, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i)
        VMP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_xbuf[i], hipEventDisableTiming));
    return VMP_OK;
}

int64_t plate_stream_A_offset(vmp_ctx *ctx, const vmp_pca_layout &L)
{
    return (max_grid(ctx) + 1) * partial_len(L) + 4096;
}

double *plate_stream_A(vmp_ctx *ctx, const vmp_pca_layout &L, void *workspace)
{
    return reinterpret_cast<double *>(workspace) + plate_stream_A_offset(ctx, L);
}

// anything on the main stream that touches X must follow the outstanding latent pass
int32_t join_plate_stream(vmp_ctx *ctx)
{
    if (ctx->x_pending) {
        VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_xdone, 0));
        ctx->x_pending = 0;
    }
    return VMP_OK;
}

int32_t run_gram_stats(vmp_ctx *ctx, const vmp_pca_layout &L, int D, int K, double *state,
                       double *P, hipStream_t s)
{
    const int nb = (int)(L.DP / GR);
    const size_t lds = ((size_t)L.KP * (L.DP + 1) + (size_t)GR * L.DP + (size_t)GR * LD)
                       * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        VMP_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)pca_gram_stats_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(pca_gram_stats_kernel, dim3(nb), dim3(NT), lds, s, L, D, K, state, state,
                       P);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    const int len = (int)(L.KP * L.KP);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((len + NT - 1) / NT), dim3(NT), 0, s, P, nb,
                       len, state + L.off_S + L.DP * L.KP);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

// row-major (rows, ld) <-> tile-major [tile][RP][32] (rows >= `rows` and columns >= N are
// zero in the tile-major copy).  One workgroup per tile; 16 bytes per lane on both sides.
template <bool TO_TILED>
__global__ void __launch_bounds__(NT)
pca_tile_kernel(double *__restrict__ R, int64_t ld, int64_t N, int rows, int RP,
                double *__restrict__ T, int64_t ntiles, int64_t tstride)
{
    const int tid = threadIdx.x;
    const int c2 = (tid & 15) * 2;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n = tile * TN + c2;
        double *tb = T + tile * tstride;
        for (int row = tid >> 4; row < RP; row += NT / 16) {
            double *rp = R + (int64_t)row * ld + n;
            v2f64 *tp = reinterpret_cast<v2f64 *>(tb + row * TN + c2);
            if (TO_TILED) {
                v2f64 v = v2f64{0.0, 0.0};
                if (row < rows) {
                    if (n + 1 < N) v = *reinterpret_cast<const v2f64 *>(rp);
                    else if (n < N) v.x = rp[0];
                }
                *tp = v;
            } else if (row < rows) {
                const v2f64 v = *tp;
                if (n + 1 < N) *reinterpret_cast<v2f64 *>(rp) = v;
                else if (n < N) rp[0] = v.x;
            }
        }
    }
}

// lay: 0 = row-major Y and X; 1 = tile-major Y, row-major X with KP rows; 3 = both tile-major
int32_t run_xpass(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N, int D, int K, double *X,
                  int64_t ldx, double *state, void *workspace, int lay)
{
    int32_t rc;
    vmp_pca_layout L;
    fill_layout(D, K, &L);
    const int DB = (int)(L.DP / 32), KT = (int)(L.KP / 16);
    const int64_t ntiles = (N + TN - 1) / TN;
    // full tiles with unpadded D, K run the predication-free instance
    const bool exact = (D == L.DP && K == L.KP);
    // a ragged last tile also runs predication-free when both arrays are padded to whole
    // tiles (its pad columns of X are then written too: <x> of whatever Y holds there)
    const bool padded = (ldy >= ntiles * TN && ldx >= ntiles * TN);
    const int64_t nfast = lay ? ntiles : (exact ? (padded ? ntiles : N / TN) : 0);
    const int occ = xpass_occupancy();
    const int ntm = vmp_tune_get("xpass_nt", env_int("VMP_PCA_XPASS_NT", 3, 0, 3));
    const int mf4 = vmp_tune_get("xpass_mfma4", env_int("VMP_PCA_XPASS_MFMA4", 0, 0, 1));
    hipStream_t m = ctx->stream;
    // VMP_PCA_PLATE_STREAM=0: everything in order on the caller's stream (A/B measurements)
    const int overlap = vmp_tune_get("plate_stream", env_int("VMP_PCA_PLATE_STREAM", 1, 0, 1));
    hipStream_t s = m;
    const double *A = state + L.off_A;
    int64_t gmax = (int64_t)ctx->num_cu * xpass_wgs_per_cu();
    if (overlap) {
        rc = ensure_plate_stream(ctx);
        if (rc != VMP_OK) return rc;
        gmax = (int64_t)ctx->xs_cus * xpass_wgs_per_cu();
        // Nothing in a Gram-form iteration reads X: the latent pass is a pure by-product of
        // (A, Y).  It runs on the plate stream from a private copy of A, so the
        // replicated-node kernels of the NEXT iteration (main stream, reserved CUs) overlap
        // it; the only ordering kept is pass(i) before pass(i+1) and before anything that
        // touches X.
        // TWO private copies, used in turn: the copy for pass i+1 is made while pass i still
        // reads the other one, so the main stream never waits for the pass in flight (only for
        // the one before it) and consecutive passes run back to back
        const int b = (int)(ctx->x_count & 1);
        ctx->x_count += 1;
        double *Ax = plate_stream_A(ctx, L, workspace) + (int64_t)b * L.KP * L.DP;
        if (ctx->x_buf_pending[b]) {
            VMP_HIP_CHECK(ctx, hipStreamWaitEvent(m, ctx->ev_xbuf[b], 0));
            ctx->x_buf_pending[b] = 0;
        }
        VMP_HIP_CHECK(ctx, hipMemcpyAsync(Ax, state + L.off_A,
                                          (size_t)(L.KP * L.DP) * sizeof(double),
                                          hipMemcpyDeviceToDevice, m));
        VMP_HIP_CHECK(ctx, hipEventRecord(ctx->ev_xfork, m));
        VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->xs, ctx->ev_xfork, 0));
        s = ctx->xs;
        A = Ax;
    }
    hipEvent_t *ev = ctx->timing ? vmp_next_events(ctx) : nullptr;
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[0], s));
    for (int pass = 0; pass < 2; ++pass) {
        const bool guard = (pass == 1);
        const int64_t t0 = guard ? nfast : 0, t1 = guard ? ntiles : nfast;
        if (t1 <= t0) continue;
        int64_t g = (t1 - t0 + 3) / 4;
        if (g > gmax) g = gmax;
        const dim3 grid((unsigned)g);
#define VMP_XP(db, kt, gd, oc, nt, ly)                                                          \
    hipLaunchKernelGGL((pca_xpass_kernel<db, kt, gd, oc, nt, ly>), grid, dim3(NT), 0, s, Y, ldy, \
                       N, D, K, A, X, ldx, t0, t1)
#define VMP_XP4(db, kt, oc, ly)                                                                  \
    hipLaunchKernelGGL((pca_xpass_kernel<db, kt, false, oc, 3, ly, 1>), grid, dim3(NT), 0, s, Y,  \
                       ldy, N, D, K, A, X, ldx, t0, t1)
        if (lay == 3 && pca_interleave()) {
            // the experiment's instances: the headline and config-2 shapes
            if (DB == 4 && KT == 2)
                hipLaunchKernelGGL((pca_xpass_kernel<4, 2, false, 3, 3, 3, 0, true>), grid, dim3(NT),
                                   0, s, Y, ldy, N, D, K, A, X, ldx, t0, t1);
            else if (DB == 2 && KT == 1)
                hipLaunchKernelGGL((pca_xpass_kernel<2, 1, false, 3, 3, 3, 0, true>), grid, dim3(NT),
                                   0, s, Y, ldy, N, D, K, A, X, ldx, t0, t1);
            else {
                VMP_SET_ERR(ctx, "pca_interleave: no instance for DB=%d KT=%d", DB, KT);
                return VMP_ERR_UNSUPPORTED;
            }
            VMP_HIP_CHECK(ctx, hipGetLastError());
            continue;
        }
#define VMP_CASE(db, kt)                                                                        \
    if (DB == db && KT == kt) {                                                                 \
        if (guard) VMP_XP(db, kt, true, 2, 0, 0);                                               \
        else if (mf4 && lay == 1 && ntm == 3) VMP_XP4(db, kt, 2, 1);                            \
        else if (mf4 && lay == 0 && ntm == 3) VMP_XP4(db, kt, 2, 0);                            \
        else if (lay == 3 && kt < 4 && ntm == 3) VMP_XP(db, kt, false, 3, 3, 3);                \
        else if (lay == 3 && kt < 4) VMP_XP(db, kt, false, 3, 0, 3);                            \
        else if (lay == 3) VMP_XP(db, kt, false, 2, 0, 3);                                      \
        else if (lay == 1 && kt < 4 && ntm == 3) VMP_XP(db, kt, false, 3, 3, 1);                \
        else if (lay == 1 && kt < 4) VMP_XP(db, kt, false, 3, 0, 1);                            \
        else if (lay == 1) VMP_XP(db, kt, false, 2, 0, 1);                                      \
        else if (ntm == 3 && kt < 4) VMP_XP(db, kt, false, 3, 3, 0);                            \
        else if (occ >= 3 && kt < 4) VMP_XP(db, kt, false, 3, 0, 0);                            \
        else VMP_XP(db, kt, false, 2, 0, 0);                                                    \
    } else
        VMP_FOR_EACH_INSTANCE(VMP_CASE)
        {
            VMP_SET_ERR(ctx, "no kernel instance for DB=%d KT=%d", DB, KT);
            return VMP_ERR_UNSUPPORTED;
        }
#undef VMP_CASE
#undef VMP_XP4
#undef VMP_XP
        VMP_HIP_CHECK(ctx, hipGetLastError());
    }
    if (ev) {
        VMP_HIP_CHECK(ctx, hipEventRecord(ev[1], s));
        VMP_HIP_CHECK(ctx, hipEventRecord(ev[2], s));
    }
    if (overlap) {
        VMP_HIP_CHECK(ctx, hipEventRecord(ctx->ev_xdone, s));
        ctx->x_pending = 1;
        const int b = (int)((ctx->x_count - 1) & 1);
        VMP_HIP_CHECK(ctx, hipEventRecord(ctx->ev_xbuf[b], s));
        ctx->x_buf_pending[b] = 1;
    }
    // messages to W from (G, A): main stream, concurrent with the pass.  Where the LDS-resident
    // tail kernel applies they are left to it (pca_tail_fast_kernel<KP, true> forms S itself: one
    // launch instead of three beside the pass); anything else that reads S first
    // calls vmp_pca_ensure_gram
    // (measured, same process, tools/c2_fuse_ab.py, profiles/r06/c2_fuse_ab.txt: the ONE workgroup that
    // then does the work of pca_gram_stats' DP / 8 workgroups beside the pass is slower -- config 2
    // 0.120 -> 0.138 ms per iteration, the 8-rank shard 0.320 -> 0.337 -- so the default stays 0)
    if (vmp_tune_get("pca_fuse_gram", env_int("VMP_PCA_FUSE_GRAM", 0, 0, 1)) && L.KP <= 32 &&
        D <= 128) {
        ctx->gram_pending = 1;
        ctx->gram_D = D;
        ctx->gram_K = K;
        ctx->gram_state = state;
        ctx->gram_P = reinterpret_cast<double *>(workspace);
        return VMP_OK;
    }
    return run_gram_stats(ctx, L, D, K, state, reinterpret_cast<double *>(workspace), m);
}

}  // namespace

int32_t vmp_pca_ensure_gram(vmp_ctx *ctx)
{
    if (!ctx || !ctx->gram_pending) return VMP_OK;
    ctx->gram_pending = 0;
    vmp_pca_layout L;
    fill_layout(ctx->gram_D, ctx->gram_K, &L);
    return run_gram_stats(ctx, L, ctx->gram_D, ctx->gram_K, ctx->gram_state, ctx->gram_P,
                          ctx->stream);
}

extern "C" {

int32_t vmp_pca_get_layout(int32_t D, int32_t K, vmp_pca_layout *out)
{
    if (!out || D < 1 || K < 1) return VMP_ERR_INVALID;
    if (D > MAX_DP || K > MAX_KP) return VMP_ERR_UNSUPPORTED;
    fill_layout(D, K, out);
    return VMP_OK;
}

int32_t vmp_pca_workspace_bytes(vmp_ctx *ctx, int32_t D, int32_t K, size_t *bytes)
{
    VMP_REQUIRE(ctx, ctx && bytes, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, D >= 1 && K >= 1, VMP_ERR_INVALID, "bad dims");
    VMP_REQUIRE(ctx, D <= MAX_DP && K <= MAX_KP, VMP_ERR_UNSUPPORTED,
                "fused PCA block supports D <= %d, K <= %d", MAX_DP, MAX_KP);
    vmp_pca_layout L;
    fill_layout(D, K, &L);
    // per-workgroup partial statistics + one scratch statistics block (Gram set-up)
    // (+ the plate stream's private copy of A)
    *bytes = (size_t)(plate_stream_A_offset(ctx, L) + 2 * L.KP * L.DP) * sizeof(double);
    return VMP_OK;
}

int32_t vmp_pca_init_state(vmp_ctx *ctx, int32_t D, int32_t K, double a0_tau, double b0_tau,
                           double a0_alpha, double b0_alpha, double *state)
{
    VMP_REQUIRE(ctx, ctx && state, VMP_ERR_INVALID, "null argument");
    vmp_pca_layout L;
    int32_t rc = vmp_pca_get_layout(D, K, &L);
    VMP_REQUIRE(ctx, rc == VMP_OK, rc, "unsupported dims D=%d K=%d", D, K);
    VMP_REQUIRE(ctx, a0_tau > 0 && b0_tau > 0 && a0_alpha > 0 && b0_alpha > 0, VMP_ERR_INVALID,
                "Gamma prior parameters must be positive");
    VMP_HIP_CHECK(ctx, hipMemsetAsync(state, 0, (size_t)L.total * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(pca_init_state_kernel, dim3(1), dim3(NT), 0, ctx->stream, L, K, a0_tau,
                       b0_tau, a0_alpha, b0_alpha, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_pca_syy(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N, int32_t D, int32_t K,
                    double *state, void *workspace)
{
    VMP_REQUIRE(ctx, ctx && Y && state && workspace, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, (ldy % 2) == 0 && ((uintptr_t)Y % 16) == 0 && ldy >= N, VMP_ERR_INVALID,
                "Y must be 16-byte aligned with an even leading dimension >= N");
    vmp_pca_layout L;
    int32_t rc = vmp_pca_get_layout(D, K, &L);
    VMP_REQUIRE(ctx, rc == VMP_OK, rc, "unsupported dims D=%d K=%d", D, K);
    double *partial = reinterpret_cast<double *>(workspace);
    int64_t work = ((N + 1) / 2) * D;
    int64_t g = (work + NT - 1) / NT;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)g), dim3(NT), 0, ctx->stream, Y, ldy, N, D,
                       partial);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(NT), 0, ctx->stream, partial, (int)g,
                       state + L.off_Syy);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_pca_gram(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N, int32_t D, int32_t K,
                     double *state, void *workspace)
{
    if (ctx) ctx->gram_pending = 0;          // (S is written anew / G changes)
    int32_t rc = check_pass_args(ctx, Y, Y, state, workspace, ldy, ldy, N, D, K);
    if (rc != VMP_OK) return rc;
    vmp_pca_layout L;
    fill_layout(D, K, &L);
    double *P = reinterpret_cast<double *>(workspace);
    double *tmpS = P + max_grid(ctx) * partial_len(L);
    for (int j0 = 0; j0 < D; j0 += MAX_KP) {
        const int Kx = (D - j0) < MAX_KP ? (D - j0) : MAX_KP;
        vmp_pca_layout Lx;
        fill_layout(D, Kx, &Lx);
        double *Xs = const_cast<double *>(Y) + (int64_t)j0 * ldy;
        rc = run_stats(ctx, false, Y, ldy, N, D, Kx, Xs, ldy, state + L.off_A, tmpS, P, false);
        if (rc != VMP_OK) return rc;
        const int n = D * Kx;
        hipLaunchKernelGGL(pca_gram_copy_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, ctx->stream,
                           tmpS, (int)Lx.KP, D, Kx, j0, (int)L.DP, state + L.off_G);
        VMP_HIP_CHECK(ctx, hipGetLastError());
    }
    return VMP_OK;
}

int32_t vmp_pca_stats_from_x(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N, int32_t D,
                             int32_t K, const double *X, int64_t ldx, double *state,
                             void *workspace)
{
    if (ctx) ctx->gram_pending = 0;          // (S is written anew / G changes)
    int32_t rc = check_pass_args(ctx, Y, X, state, workspace, ldy, ldx, N, D, K);
    if (rc != VMP_OK) return rc;
    vmp_pca_layout L;
    fill_layout(D, K, &L);
    rc = join_plate_stream(ctx);
    if (rc != VMP_OK) return rc;
    return run_stats(ctx, false, Y, ldy, N, D, K, const_cast<double *>(X), ldx, state + L.off_A,
                     state + L.off_S, reinterpret_cast<double *>(workspace), false);
}

int32_t vmp_pca_pass(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N, int32_t D, int32_t K,
                     double *X, int64_t ldx, double *state, void *workspace)
{
    if (ctx) ctx->gram_pending = 0;          // (S is written anew / G changes)
    int32_t rc = check_pass_args(ctx, Y, X, state, workspace, ldy, ldx, N, D, K);
    if (rc != VMP_OK) return rc;
    vmp_pca_layout L;
    fill_layout(D, K, &L);
    rc = join_plate_stream(ctx);
    if (rc != VMP_OK) return rc;
    return run_stats(ctx, true, Y, ldy, N, D, K, X, ldx, state + L.off_A, state + L.off_S,
                     reinterpret_cast<double *>(workspace), true);
}

int32_t vmp_pca_xpass(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N, int32_t D, int32_t K,
                      double *X, int64_t ldx, double *state, void *workspace)
{
    int32_t rc = check_pass_args(ctx, Y, X, state, workspace, ldy, ldx, N, D, K);
    if (rc != VMP_OK) return rc;
    return run_xpass(ctx, Y, ldy, N, D, K, X, ldx, state, workspace, 0);
}

int32_t vmp_pca_tiled_doubles(int32_t D, int32_t K, int64_t N, int64_t *y_doubles,
                              int64_t *x_doubles)
{
    if (D < 1 || K < 1 || N < 0) return VMP_ERR_INVALID;
    if (D > MAX_DP || K > MAX_KP) return VMP_ERR_UNSUPPORTED;
    vmp_pca_layout L;
    fill_layout(D, K, &L);
    const int64_t ntiles = (N + TN - 1) / TN;
    if (y_doubles) *y_doubles = ntiles * (L.DP + (pca_interleave() ? L.KP : 0)) * TN;
    if (x_doubles) *x_doubles = ntiles * L.KP * TN;
    return VMP_OK;
}

int32_t vmp_pca_tile_y(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N, int32_t D, int32_t K,
                       double *Yt)
{
    VMP_REQUIRE(ctx, ctx && Y && Yt, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, (ldy % 2) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)Yt % 16) == 0
                && ldy >= N, VMP_ERR_INVALID,
                "Y must be 16-byte aligned with an even leading dimension >= N");
    vmp_pca_layout L;
    int32_t rc = vmp_pca_get_layout(D, K, &L);
    VMP_REQUIRE(ctx, rc == VMP_OK, rc, "unsupported dims D=%d K=%d", D, K);
    const int64_t ntiles = (N + TN - 1) / TN;
    if (ntiles == 0) return VMP_OK;
    int64_t g = ntiles < (int64_t)ctx->num_cu * 16 ? ntiles : (int64_t)ctx->num_cu * 16;
    hipLaunchKernelGGL(pca_tile_kernel<true>, dim3((unsigned)g), dim3(NT), 0, ctx->stream,
                       const_cast<double *>(Y), ldy, N, D, (int)L.DP, Yt, ntiles,
                       (int64_t)(L.DP + (pca_interleave() ? L.KP : 0)) * TN);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_pca_tile_x(vmp_ctx *ctx, int32_t to_tiled, double *X, int64_t ldx, int64_t N,
                       int32_t D, int32_t K, double *Xt)
{
    VMP_REQUIRE(ctx, ctx && X && Xt, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, (ldx % 2) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Xt % 16) == 0
                && ldx >= N, VMP_ERR_INVALID,
                "X must be 16-byte aligned with an even leading dimension >= N");
    vmp_pca_layout L;
    int32_t rc = vmp_pca_get_layout(D, K, &L);
    VMP_REQUIRE(ctx, rc == VMP_OK, rc, "unsupported dims D=%d K=%d", D, K);
    rc = join_plate_stream(ctx);
    if (rc != VMP_OK) return rc;
    const int64_t ntiles = (N + TN - 1) / TN;
    if (ntiles == 0) return VMP_OK;
    int64_t g = ntiles < (int64_t)ctx->num_cu * 16 ? ntiles : (int64_t)ctx->num_cu * 16;
    // interleaved layout: Xt points at the <x> rows of tile 0 inside the [tile][DP + KP][32] array
    const int64_t xts = (int64_t)(L.KP + (pca_interleave() ? L.DP : 0)) * TN;
    if (to_tiled)
        hipLaunchKernelGGL(pca_tile_kernel<true>, dim3((unsigned)g), dim3(NT), 0, ctx->stream, X,
                           ldx, N, K, (int)L.KP, Xt, ntiles, xts);
    else
        hipLaunchKernelGGL(pca_tile_kernel<false>, dim3((unsigned)g), dim3(NT), 0, ctx->stream, X,
                           ldx, N, K, (int)L.KP, Xt, ntiles, xts);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_pca_xpass_tiled(vmp_ctx *ctx, const double *Yt, int64_t N, int32_t D, int32_t K,
                            double *X, int64_t ldx, int32_t x_tiled, double *state,
                            void *workspace)
{
    VMP_REQUIRE(ctx, ctx != nullptr, VMP_ERR_INVALID, "null context");
    VMP_REQUIRE(ctx, Yt && X && state && workspace, VMP_ERR_INVALID, "null pointer argument");
    VMP_REQUIRE(ctx, D >= 1 && K >= 1 && N >= 0, VMP_ERR_INVALID, "bad dims D=%d K=%d N=%lld", D,
                K, (long long)N);
    VMP_REQUIRE(ctx, D <= MAX_DP && K <= MAX_KP, VMP_ERR_UNSUPPORTED,
                "fused PCA block supports D <= %d, K <= %d (got D=%d, K=%d)", MAX_DP, MAX_KP, D, K);
    VMP_REQUIRE(ctx, ((uintptr_t)Yt % 16) == 0 && ((uintptr_t)X % 16) == 0, VMP_ERR_INVALID,
                "tile-major arrays must be 16-byte aligned");
    const int64_t ntiles = (N + TN - 1) / TN;
    if (!x_tiled) {
        VMP_REQUIRE(ctx, (ldx % 2) == 0 && ldx >= ntiles * TN, VMP_ERR_INVALID,
                    "row-major X beside a tile-major Y needs KP rows of an even leading "
                    "dimension >= 32 ceil(N/32) (ldx=%lld)", (long long)ldx);
        VMP_REQUIRE(ctx, ldx < (int64_t)170000000, VMP_ERR_UNSUPPORTED,
                    "leading dimension >= 1.7e8 elements per shard is not supported");
    }
    return run_xpass(ctx, Yt, 0, N, D, K, X, ldx, state, workspace, x_tiled ? 3 : 1);
}

int32_t vmp_pca_xjoin(vmp_ctx *ctx)
{
    {
        const int32_t rcg = vmp_pca_ensure_gram(ctx);
        if (rcg != VMP_OK) return rcg;
    }
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null argument");
    if (ctx->x_pending) {
        VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_xdone, 0));
        ctx->x_pending = 0;
    }
    return VMP_OK;
}

int32_t vmp_pca_last_pass_ms(vmp_ctx *ctx, double *ms_pass, double *ms_reduce)
{
    VMP_REQUIRE(ctx, ctx && ctx->timing && ctx->ev[0] && ctx->ev_n > 0, VMP_ERR_INVALID,
                "no timed pass (vmp_ctx_set_timing)");
    hipEvent_t *e = ctx->ev + 3 * ((ctx->ev_n - 1) % VMP_EV_RING);
    VMP_HIP_CHECK(ctx, hipEventSynchronize(e[2]));
    float a = 0.f, b = 0.f;
    VMP_HIP_CHECK(ctx, hipEventElapsedTime(&a, e[0], e[1]));
    VMP_HIP_CHECK(ctx, hipEventElapsedTime(&b, e[1], e[2]));
    if (ms_pass) *ms_pass = a;
    if (ms_reduce) *ms_reduce = b;
    return VMP_OK;
}

}  // extern "C"
