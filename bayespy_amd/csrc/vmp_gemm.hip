// vmp_gemm.hip -- strided, batched fp64 contraction on the matrix cores.
//
// Where a SumMultiply / sum_multiply call collapses to a dense M x N x K contraction
//     C[b, m, n] = scale * sum_k A[b, m, k] * B[b, k, n]
// (the messages and moments of dot.py:355, :403, :581 with array masks, the weighted
// mixture statistics of mixture.py:126-158, the chain statistics of
// gaussian_markov_chain.py:462-475, ...), the generic engine routes it here instead of the
// scalar einsum loop the reference uses (np.einsum(optimize=False), utils/misc.py:906).
//
// Arbitrary element strides on every axis (stride 0 = broadcast batch axis), up to three
// batch axes, split-K with fixed-order combination (deterministic).  64 x 64 x 16 tiles
// staged through LDS with conflict-free strides (18 / 80 doubles), each of the four
// wavefronts owns a 32 x 32 block = 2 x 2 v_mfma_f64_16x16x4_f64 accumulators.
#include "vmp_common.h"

namespace {

constexpr int NT = 256;
constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDA_S = BK + 2;     // 18: A tile [BM][18]
constexpr int LDB_S = BN + 16;    // 80: B tile [BK][80]

struct GemmArgs {
    int64_t M, N, K;
    int nb;                          // batch axes (<= 3)
    int64_t bshape[3];
    int64_t a_bs[3], b_bs[3], c_bs[3];
    int64_t a_ms, a_ks, b_ks, b_ns, c_ms, c_ns;
    const double *A, *B;
    double *C;                       // final output, or the partial buffer when nsplit > 1
    int nsplit;
    int64_t kchunk;                  // K elements per split (multiple of BK)
    double scale;
};

__device__ inline v4f64 mfma_f64(double a, double b, v4f64 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// TM x TN output tile: 64 x 64 (wavefronts 2 x 2) or 128 x 32 (4 x 1: outputs of at most 32 columns --
// the K = 32 components of a mixture -- ran the 64-wide tile half empty); every wavefront a 32 x 32 block
template <int TM, int TN>
__global__ void __launch_bounds__(NT, 2)
gemm_kernel(GemmArgs g)
{
    constexpr int BM = TM, BN = TN, LDB_S = TN + 16;
    constexpr int EA = TM * BK / NT, EB = BK * TN / NT;      // elements per thread of the tile loads
    __shared__ double As[BM * LDA_S];
    __shared__ double Bs[BK * LDB_S];
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, l15 = l & 15, l4 = l >> 4;
    const int wm = (TN == 64 ? (w >> 1) : w) * 32, wn = (TN == 64 ? (w & 1) : 0) * 32;
    const int64_t tiles_n = (g.N + BN - 1) / BN;
    const int64_t tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int64_t m0 = tm * BM, n0 = tn * BN;
    const int sp = blockIdx.y;
    // batch offsets
    int64_t bz = blockIdx.z, aoff = 0, boff = 0, coff = 0;
    for (int d = g.nb - 1; d >= 0; --d) {
        const int64_t q = bz / g.bshape[d], c = bz - q * g.bshape[d];
        bz = q;
        aoff += c * g.a_bs[d];
        boff += c * g.b_bs[d];
        coff += c * g.c_bs[d];
    }
    const double *A = g.A + aoff;
    const double *B = g.B + boff;
    const int64_t k_begin = (int64_t)sp * g.kchunk;
    const int64_t k_end = (k_begin + g.kchunk < g.K) ? k_begin + g.kchunk : g.K;

    // thread -> element maps for the tile loads; lanes run along the dense axis
    const bool a_kfast = (g.a_ks == 1) || (g.a_ms != 1);     // else M is the dense axis of A
    const bool b_nfast = (g.b_ns == 1) || (g.b_ks != 1);

    v4f64 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};

    // Tile loads go global -> registers -> LDS; the registers of tile k0+BK are filled while the
    // MFMAs of tile k0 run (the loads' latency hides behind the matrix pipe).
    double ra[EA], rb[EB];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + e * NT;
            const int mi = a_kfast ? idx / BK : idx % BM;
            const int ki = a_kfast ? idx % BK : idx / BM;
            const int64_t m = m0 + mi, k = k0 + ki;
            ra[e] = (m < g.M && k < k_end) ? A[m * g.a_ms + k * g.a_ks] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int idx = tid + e * NT;
            const int ki = b_nfast ? idx / BN : idx % BK;
            const int ni = b_nfast ? idx % BN : idx / BK;
            const int64_t k = k0 + ki, n = n0 + ni;
            rb[e] = (k < k_end && n < g.N) ? B[k * g.b_ks + n * g.b_ns] : 0.0;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + e * NT;
            const int mi = a_kfast ? idx / BK : idx % BM;
            const int ki = a_kfast ? idx % BK : idx / BM;
            As[mi * LDA_S + ki] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int idx = tid + e * NT;
            const int ki = b_nfast ? idx / BN : idx % BK;
            const int ni = b_nfast ? idx % BN : idx / BK;
            Bs[ki * LDB_S + ni] = rb[e];
        }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
        stage();
        __syncthreads();
        if (k0 + BK < k_end) fetch(k0 + BK);
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
            double af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[(wm + i * 16 + l15) * LDA_S + 4 * q + l4];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[(4 * q + l4) * LDB_S + wn + j * 16 + l15];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }
    // C/D layout: col = lane&15, row = (lane>>4) + 4*reg
    if (g.nsplit == 1) {
        double *C = g.C + coff;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + wm + i * 16 + l4 + 4 * r, n = n0 + wn + j * 16 + l15;
                    if (m < g.M && n < g.N) C[m * g.c_ms + n * g.c_ns] = g.scale * acc[i][j][r];
                }
    } else {
        // partial buffer: [split][batch][M][N] dense
        double *P = g.C + (((int64_t)sp * gridDim.z + blockIdx.z) * g.M) * g.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + wm + i * 16 + l4 + 4 * r, n = n0 + wn + j * 16 + l15;
                    if (m < g.M && n < g.N) P[m * g.N + n] = acc[i][j][r];
                }
    }
}

// K-long contractions with a small output (ceil(M/16) * ceil(N/16) <= 4 blocks of 16 x 16, e.g. the
// plate sums  sum_n <x_n><x_n>^T  and  sum_n y_dn <x_n>^T  of a PCA model with N = 1e6): the 64 x 64
// tile above would run 1/16 .. 1/4 full and read with a quarter of its lanes.  Here every wavefront
// owns a K slice and keeps the whole output in MT x NB accumulators; operands go global -> registers:
// lane (l15, l4) holds A[16 i + l15][kb + 4 l4 + j] and B[kb + 4 l4 + j][16 n + l15], j = 0..3 -- the
// instruction contracts lane group l4 of A with lane group l4 of B, so any assignment of k's to
// (l4, j) that is the same on both sides is valid, and this one gives a lane four consecutive k's
// (32 contiguous bytes of a K-contiguous operand) and sixteen consecutive m's or n's across l15
// (a 128-byte line of an M- or N-contiguous operand).  The four wavefronts of a workgroup meet in LDS
// in fixed order, the workgroups through the partial buffer and gemm_finish_kernel: deterministic.
template <int MT, int NB>
__global__ void __launch_bounds__(NT)
gemm_skinny_kernel(GemmArgs g)
{
    __shared__ double red[(NT / 64 - 1) * MT * NB * 256];
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, l15 = l & 15, l4 = l >> 4;
    const int64_t gw = (int64_t)blockIdx.x * (NT / 64) + w;
    const int64_t k_begin = gw * g.kchunk;
    const int64_t k_end = (k_begin + g.kchunk < g.K) ? k_begin + g.kchunk : g.K;
    const double *__restrict__ A = g.A;
    const double *__restrict__ B = g.B;
    v4f64 acc[MT][NB];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[i][n] = v4f64{0.0, 0.0, 0.0, 0.0};
    int64_t arow[MT], bcol[NB];
    bool aok[MT], bok[NB];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        aok[i] = 16 * i + l15 < g.M;
        arow[i] = aok[i] ? (int64_t)(16 * i + l15) * g.a_ms : 0;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        bok[n] = 16 * n + l15 < g.N;
        bcol[n] = bok[n] ? (int64_t)(16 * n + l15) * g.b_ns : 0;
    }
    const bool avec = g.a_ks == 1, bvec = g.b_ks == 1;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += 16) {
        const int64_t kk = k0 + 4 * l4;
        double a[MT][4], b[NB][4];
        if (kk + 4 <= k_end) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const double *p = A + arow[i] + kk * g.a_ks;
                if (avec) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[i][j] = p[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[i][j] = p[j * g.a_ks];
                }
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const double *p = B + bcol[n] + kk * g.b_ks;
                if (bvec) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[n][j] = p[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[n][j] = p[j * g.b_ks];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool kok = kk + j < k_end;
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    a[i][j] = kok ? A[arow[i] + (kk + j) * g.a_ks] : 0.0;
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    b[n][j] = kok ? B[bcol[n] + (kk + j) * g.b_ks] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!aok[i]) a[i][j] = 0.0;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!bok[n]) b[n][j] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[i][n] = mfma_f64(a[i][j], b[n][j], acc[i][n]);
    }
    // wavefronts 1..3 hand their blocks to wavefront 0, which adds them in order
    if (w > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[(((w - 1) * MT + i) * NB + n) * 256 + r * 64 + l] = acc[i][n][r];
    }
    __syncthreads();
    if (w > 0) return;
#pragma unroll
    for (int ww = 0; ww < NT / 64 - 1; ++ww)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][n][r] += red[((ww * MT + i) * NB + n) * 256 + r * 64 + l];
    const bool direct = gridDim.x == 1;
    double *out = direct ? g.C : g.C + (int64_t)blockIdx.x * g.M * g.N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = 16 * i + l4 + 4 * r, nn = 16 * n + l15;
                if (m < g.M && nn < g.N) {
                    if (direct) out[m * g.c_ms + nn * g.c_ns] = g.scale * acc[i][n][r];
                    else out[m * g.N + nn] = acc[i][n][r];
                }
            }
}

// Many rows, few columns, short contraction (M >= 4096, N <= 64, K <= 64: a K x K matrix applied to
// every plate, the Dot message to the plate-side parent): the 64 x 64 tile runs its B side 1/4 full
// and pays two barriers per 16 k's for a kernel that only streams A in and C out.  Here B lives in
// registers (KQ x NB fragments per lane, loaded once per wavefront), a wavefront walks blocks of 16
// rows: KQ loads of A (128-byte lines of an M-contiguous operand, 32-byte pieces of a K-contiguous
// one), KQ x NB matrix instructions, NB x 4 stores; the next block's loads are issued before the
// current block's instructions.
template <int KQ, int NB>
__global__ void __launch_bounds__(NT)
gemm_tall_kernel(GemmArgs g)
{
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, l15 = l & 15, l4 = l >> 4;
    const int64_t nblk = (g.M + 15) / 16;
    const int64_t nw = (int64_t)gridDim.x * (NT / 64);
    const double *__restrict__ A = g.A;
    const double *__restrict__ B = g.B;
    double b[KQ][NB];
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int64_t k = 4 * q + l4, nn = 16 * n + l15;
            b[q][n] = (k < g.K && nn < g.N) ? B[k * g.b_ks + nn * g.b_ns] : 0.0;
        }
    int64_t koff[KQ];
    bool kok[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        kok[q] = 4 * q + l4 < g.K;
        koff[q] = kok[q] ? (int64_t)(4 * q + l4) * g.a_ks : 0;
    }
    auto load = [&](int64_t blk, double *a) {
        const int64_t m = blk * 16 + l15;
        const bool ok = blk < nblk && m < g.M;
        const double *p = A + (ok ? m : 0) * g.a_ms;
#pragma unroll
        for (int q = 0; q < KQ; ++q) a[q] = (ok && kok[q]) ? p[koff[q]] : 0.0;
    };
    double a0[KQ], a1[KQ];
    int64_t blk = (int64_t)blockIdx.x * (NT / 64) + w;
    load(blk, a0);
    for (; blk < nblk; blk += nw) {
        load(blk + nw, a1);
        v4f64 acc[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = mfma_f64(a0[q], b[q][n], acc[n]);
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = blk * 16 + l4 + 4 * r, nn = 16 * n + l15;
                if (m < g.M && nn < g.N) g.C[m * g.c_ms + nn * g.c_ns] = g.scale * acc[n][r];
            }
#pragma unroll
        for (int q = 0; q < KQ; ++q) a0[q] = a1[q];
    }
}

// Combination of the K slices: 16 outputs per workgroup, 16 lanes per output walk the slices with a
// stride of 16 (independent loads), then the 16 partial sums are added in order -- the result depends
// on the number of slices only.  (One thread per output walking all slices was 0.24 ms for 256
// outputs x 1024 slices: a chain of dependent-latency loads on a single workgroup.)
__global__ void __launch_bounds__(NT)
gemm_finish_kernel(GemmArgs g, const double *__restrict__ P, double *__restrict__ C,
                   int64_t nbatch)
{
    __shared__ double part[16][17];
    const int64_t total = nbatch * g.M * g.N;
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (int64_t e0 = (int64_t)blockIdx.x * 16; e0 < total; e0 += (int64_t)gridDim.x * 16) {
        const int64_t e = e0 + o;
        double s = 0.0;
        if (e < total)
            for (int sp = sl; sp < g.nsplit; sp += 16) s += P[(int64_t)sp * total + e];
        part[sl][o] = s;
        __syncthreads();
        if (sl == 0 && e < total) {
            s = part[0][o];
#pragma unroll
            for (int q = 1; q < 16; ++q) s += part[q][o];
        }
        __syncthreads();
        if (sl != 0 || e >= total) continue;
        const int64_t bz = e / (g.M * g.N);
        const int64_t r = e - bz * g.M * g.N;
        const int64_t m = r / g.N, n = r - m * g.N;
        int64_t b = bz, coff = 0;
        for (int d = g.nb - 1; d >= 0; --d) {
            const int64_t q = b / g.bshape[d], c = b - q * g.bshape[d];
            b = q;
            coff += c * g.c_bs[d];
        }
        C[coff + m * g.c_ms + n * g.c_ns] = g.scale * s;
    }
}

}  // namespace

extern "C" {

int32_t vmp_gemm_strided(vmp_ctx *ctx, int32_t nbatch_dims, const int64_t *bshape, int64_t M,
                         int64_t N, int64_t K, const double *A, const int64_t *a_bstride,
                         int64_t a_ms, int64_t a_ks, const double *B, const int64_t *b_bstride,
                         int64_t b_ks, int64_t b_ns, double *C, const int64_t *c_bstride,
                         int64_t c_ms, int64_t c_ns, double scale, void *workspace,
                         size_t workspace_bytes)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && A && B && C, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, nbatch_dims >= 0 && nbatch_dims <= 3, VMP_ERR_UNSUPPORTED,
                "at most 3 batch axes");
    VMP_REQUIRE(ctx, M >= 0 && N >= 0 && K >= 0, VMP_ERR_INVALID, "bad dims");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = M; g.N = N; g.K = K;
    g.nb = nbatch_dims;
    int64_t nbatch = 1;
    for (int d = 0; d < nbatch_dims; ++d) {
        g.bshape[d] = bshape[d];
        g.a_bs[d] = a_bstride[d];
        g.b_bs[d] = b_bstride[d];
        g.c_bs[d] = c_bstride[d];
        nbatch *= bshape[d];
    }
    if (M == 0 || N == 0 || nbatch == 0) return VMP_OK;
    g.a_ms = a_ms; g.a_ks = a_ks; g.b_ks = b_ks; g.b_ns = b_ns; g.c_ms = c_ms; g.c_ns = c_ns;
    g.A = A; g.B = B; g.scale = scale;
    hipStream_t s = ctx->stream;
    // a small output under a long contraction: K slices per wavefront, operands in registers
    const int64_t mt = (M + 15) / 16, nbk = (N + 15) / 16;
    if (nbatch == 1 && mt * nbk <= 4 && K >= 4096) {
        int64_t waves = (int64_t)ctx->num_cu * 8;
        const int64_t maxw = (K + 63) / 64;
        if (waves > maxw) waves = maxw;
        int64_t wgs = (waves + NT / 64 - 1) / (NT / 64);
        if (wgs > 1 && (!workspace || (size_t)wgs * M * N * sizeof(double) > workspace_bytes))
            wgs = workspace ? (int64_t)(workspace_bytes / ((size_t)M * N * sizeof(double))) : 1;
        if (wgs < 1) wgs = 1;
        waves = wgs * (NT / 64);
        g.kchunk = (((K + waves - 1) / waves + 15) / 16) * 16;
        g.nsplit = (int)wgs;
        g.C = wgs > 1 ? reinterpret_cast<double *>(workspace) : C;
#define VMP_SKINNY(a, b)                                                                          \
    if (mt == a && nbk == b)                                                                      \
        hipLaunchKernelGGL((gemm_skinny_kernel<a, b>), dim3((unsigned)wgs), dim3(NT), 0, s, g)
        VMP_SKINNY(1, 1); VMP_SKINNY(1, 2); VMP_SKINNY(1, 3); VMP_SKINNY(1, 4);
        VMP_SKINNY(2, 1); VMP_SKINNY(2, 2); VMP_SKINNY(3, 1); VMP_SKINNY(4, 1);
#undef VMP_SKINNY
        if (wgs > 1) {
            int64_t gb = (M * N + 15) / 16;
            hipLaunchKernelGGL(gemm_finish_kernel, dim3((unsigned)gb), dim3(NT), 0, s, g,
                               reinterpret_cast<const double *>(workspace), C, (int64_t)1);
        }
        VMP_HIP_CHECK(ctx, hipGetLastError());
        return VMP_OK;
    }
    // many rows, few columns, short contraction: B in registers, A streamed by row blocks
    {
        const int64_t kq = K <= 16 ? 4 : (K <= 32 ? 8 : 16), nb = N <= 16 ? 1 : (N <= 32 ? 2 : 4);
        if (nbatch == 1 && M >= 4096 && N <= 64 && K <= 64 && kq * nb <= 32) {
            g.C = C;
            g.nsplit = 1;
            int64_t wgs = ((M + 15) / 16 + NT / 64 - 1) / (NT / 64);
            const int64_t cap = (int64_t)ctx->num_cu * 8;
            if (wgs > cap) wgs = cap;
#define VMP_TALL(a, b)                                                                            \
    if (kq == a && nb == b)                                                                       \
        hipLaunchKernelGGL((gemm_tall_kernel<a, b>), dim3((unsigned)wgs), dim3(NT), 0, s, g)
            VMP_TALL(4, 1); VMP_TALL(4, 2); VMP_TALL(4, 4); VMP_TALL(8, 1); VMP_TALL(8, 2);
            VMP_TALL(8, 4); VMP_TALL(16, 1); VMP_TALL(16, 2);
#undef VMP_TALL
            VMP_HIP_CHECK(ctx, hipGetLastError());
            return VMP_OK;
        }
    }
    // outputs of at most 32 columns: the 128 x 32 tile (tune key gemm_narrow_tile)
    const bool narrow = N <= 32 && M > 64 && vmp_tune_get("gemm_narrow_tile", 1) != 0;
    const int64_t TMh = narrow ? 128 : BM, TNh = narrow ? 32 : BN;
    const int64_t tiles = ((M + TMh - 1) / TMh) * ((N + TNh - 1) / TNh);
    VMP_REQUIRE(ctx, nbatch <= 65535, VMP_ERR_UNSUPPORTED, "too many batch elements (%lld)",
                (long long)nbatch);
    // split K when the output alone cannot fill the chip
    int64_t nsplit = 1;
    const int64_t ksteps = (K + BK - 1) / BK;
    const int64_t want = ((int64_t)ctx->num_cu * 4 + tiles * nbatch - 1) / (tiles * nbatch);
    if (want > 1 && ksteps >= 8) {
        nsplit = want < ksteps / 4 ? want : ksteps / 4;
        if (nsplit > 1024) nsplit = 1024;
        if (nsplit < 1) nsplit = 1;
        const size_t need = (size_t)nsplit * nbatch * M * N * sizeof(double);
        if (!workspace || need > workspace_bytes) {
            nsplit = workspace ? (int64_t)(workspace_bytes / ((size_t)nbatch * M * N * sizeof(double)))
                               : 1;
            if (nsplit < 1) nsplit = 1;
        }
    }
    g.nsplit = (int)nsplit;
    g.kchunk = ((ksteps + nsplit - 1) / nsplit) * BK;
    if (g.kchunk < BK) g.kchunk = BK;
    g.C = nsplit > 1 ? reinterpret_cast<double *>(workspace) : C;
    const dim3 grid((unsigned)tiles, (unsigned)nsplit, (unsigned)nbatch);
    if (narrow)
        hipLaunchKernelGGL((gemm_kernel<128, 32>), grid, dim3(NT), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_kernel<64, 64>), grid, dim3(NT), 0, s, g);
    if (nsplit > 1) {
        const int64_t total = nbatch * M * N;
        int64_t gb = (total + 15) / 16;
        if (gb > (int64_t)ctx->num_cu * 16) gb = (int64_t)ctx->num_cu * 16;
        hipLaunchKernelGGL(gemm_finish_kernel, dim3((unsigned)gb), dim3(NT), 0, s, g,
                           reinterpret_cast<const double *>(workspace), C, nbatch);
    }
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // extern "C"
