// vmp_mpca.hip -- fused probabilistic-PCA / factor-analysis VB block WITH MISSING VALUES (gfx950).
//
// Model block of bayespy/demos/pca.py:22-61 with Y.observe(y, mask=array) (demos/pca.py:80-82,
// how the reference's PCA is normally used).  An array mask gives every plate its own posterior
// covariance; the reference then
//   * builds (1,N,K,K) / (D,1,K,K) arrays of second moments and contracts them with einsum
//     (dot.py:355,403,581: 2 N D K^2 flops each for the message to W and to X),
//   * loops in PYTHON over the N + D plates calling SciPy's Cholesky (utils/linalg.py:31-63),
//   * applies masks by multiplication and plate sums (node.py:457-526, :650; misc.py:805).
// Here one X.update() is three kernels per chunk of plates, all on the fp64 matrix cores, and
// no (N,K,K) array ever exists:
//   mpca_lambda   Lam~_n = sum_d m_dn <w w^T>_d  and  rhs~_n = sum_d m_dn y_dn <w_d>
//                 GEMM (n x d) . (d x P), P = packed lower triangle of the K x K matrix
//   mpca_sweep    per plate: Lam_n = c I + <tau> Lam~_n inverted in registers by the symmetric
//                 sweep operator (vmp_sweep.h), <x_n>, <x x^T>_n = Cov_n + <x_n><x_n>^T, log|Cov_n|
//   mpca_stats    M_d = sum_n m_dn <x x^T>_n,  r_d = sum_n m_dn y_dn <x_n>
//                 GEMM (d x n) . (n x P), the only things W, tau and the bound consume
// W.update() is one wavefront per row d (the same sweep), tau / alpha / bound are one small kernel.
//
// Layouts (all fp64 unless noted; "plate" = the observation axis n; sub = 16 consecutive plates):
//   Ymt  [tile][DP][32]      m * y, tile-major, zero where masked / padded  (as vmp_pca_tile_y)
//   Mb1  [sub][64] uint32    lane l, bit q      = m[4q + (l>>4)][16 sub + (l&15)]    (A operand of mpca_lambda)
//   Mb2  [sub][64] uint32    lane l, bit 4dt+qq = m[16dt + (l&15)][16 sub + 4qq + (l>>4)] (A operand of mpca_stats)
//   Xm   [n][KP]             <x_n>, plate-major (the reference's (1,N,K) view)
//   packed symmetric index   p(i,j) = i(i+1)/2 + j, i >= j, over the PADDED size KP; PT = ceil(P/16)
//   panel [c][DQ/2][64][2]   B operands of mpca_lambda in fragment order: column tile c < PT of the
//                            packed <w w^T>_d, then KT tiles of <w_d>; element (d, col) at
//                            ((c*(DQ/2) + (d/8))*64 + (d%4)*16 + col%16)*2 + (d/4)%2
//   Lam  [n][LR]             scratch of a chunk, LR = 16 (PT + KT): packed Lam~_n | rhs~_n
//   XXf  [n/4][PT2][64][2]   scratch of a chunk: packed <x x^T>_n in the B-operand order of mpca_stats,
//                            PT2 = ceil(PT / 2) PAIRS of column tiles per 16 bytes: element (n, p) at
//                            (((n/4)*PT2 + p/32)*64 + (n%4)*16 + p%16)*2 + (p/16)%2
//                            -- every 1 KB block belongs to ONE group of four plates, i.e. to one
//                            wavefront-iteration of the per-plate stage (round 3: with two k-steps
//                            per 16 bytes a block was completed by two wavefronts at different
//                            times and left the L2 half-written: 8.3 GB written per 2^20 plates
//                            instead of 4.7, profiles/r03/pmc_blk4_before.txt)
//   Mst  [DP][LR]            packed M_d | r_d  (what ranks all-reduce)
#include "vmp_sweep.h"

#include <type_traits>

namespace {

using namespace vmp_sweep;

constexpr int NT = 256;
constexpr int TN = 32;          // plates per tile of Ymt

__host__ __device__ inline int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// XXf (see the layout note above): offset of packed entry p inside the block row of its plate
// group, and the start of plate n's lanes in that block row
__host__ __device__ inline int xxf_off(int p) { return (p >> 5) * 128 + (p & 15) * 2 + ((p >> 4) & 1); }
__host__ __device__ inline int64_t xxf_base(int64_t n, int PT)
{
    return ((n >> 2) * ((PT + 1) / 2) * 64 + (n & 3) * 16) * 2;
}

__device__ __forceinline__ double mfma4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    return __builtin_amdgcn_update_dpp(0.0, x, CTRL, 0xf, 0xf, true);
}

struct mpca_dims {
    int D, K, DP, KP, DQ, DT, KT, P, PT, CT, LR;
};

inline mpca_dims make_dims(int D, int K)
{
    mpca_dims m;
    m.D = D;
    m.K = K;
    int db = 1;
    while (32 * db < D) db <<= 1;
    m.DP = 32 * db;
    m.KP = K <= 16 ? 16 : 32;
    m.DQ = m.DP / 4;
    m.DT = m.DP / 16;
    m.KT = m.KP / 16;
    m.P = m.KP * (m.KP + 1) / 2;
    m.PT = (m.P + 15) / 16;
    m.CT = m.PT + m.KT;
    m.LR = 16 * m.CT;
    return m;
}

inline void fill_layout(int D, int K, vmp_mpca_layout *L)
{
    const mpca_dims m = make_dims(D, K);
    int64_t o = 0;
    L->DP = m.DP;
    L->KP = m.KP;
    L->P = m.P;
    L->PT = m.PT;
    L->LR = m.LR;
    L->off_tau = o;      o += 8;
    L->off_alpha = o;    o += 4 * m.KP;
    L->off_scal = o;     o += 16;
    L->off_L = o;        o += 8;
    L->off_W = o;        o += (int64_t)m.DP * m.KP;
    L->off_WW = o;       o += (int64_t)m.DP * m.KP * m.KP;
    L->off_ldW = o;      o += m.DP;
    L->off_M = o;        o += (int64_t)m.DP * m.LR;
    L->off_panel = o;    o += (int64_t)m.CT * (m.DQ / 2) * 128;
    L->off_panel_x = o;  o += (int64_t)m.CT * (m.DQ / 2) * 128;
    L->off_Sxx = o;      o += (int64_t)m.KP * m.KP;
    L->off_rowobs = o;   o += m.DP;
    L->total = (o + 7) / 8 * 8;
}

// scal block: [0] sum m y^2  [1] sum m  [2] sum_n tr<xx>_n  [3] sum_n log|Cov_n|  [4] plates N
//             [5] status     [6] <tau> used by the last X.update()   [7] residual (diagnostic)
enum { SC_SYY = 0, SC_NOBS, SC_TRXX, SC_LDX, SC_N, SC_STATUS, SC_TAUX, SC_RESID };

__device__ inline int64_t panel_index(int DQ, int c, int d, int col16)
{
    const int q = d >> 2;
    return (((int64_t)c * (DQ / 2) + (q >> 1)) * 64 + (d & 3) * 16 + col16) * 2 + (q & 1);
}

// -------------------------------------------------------------------------------------------
// set-up: masked tile-major data, the two bit layouts of the mask, sum m y^2, sum m
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT)
mpca_prepare_kernel(const double *__restrict__ Y, int64_t ldy, const uint8_t *__restrict__ mask,
                    int64_t ldm, int64_t N, int D, int DP, double *__restrict__ Ymt,
                    uint32_t *__restrict__ Mb1, uint32_t *__restrict__ Mb2,
                    double *__restrict__ partial, int64_t ntiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t ms[128 * TN];
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x;
    double syy = 0.0, nobs = 0.0;
    int rowobs = 0;              // thread d < DP: observations of dimension d in this workgroup's tiles
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        double *tb = Ymt + tile * ((int64_t)DP * TN);
        for (int e = tid; e < DP * TN; e += NT) {
            const int d = e / TN, j = e - d * TN;
            const int64_t n = tile * TN + j;
            uint8_t mv = 0;
            double v = 0.0;
            if (d < D && n < N) {
                mv = mask ? (mask[(int64_t)d * ldm + n] != 0) : 1;
                if (mv) v = Y[(int64_t)d * ldy + n];
            }
            ms[e] = mv;
            tb[e] = v;
            syy += v * v;
            nobs += (double)mv;
        }
        __syncthreads();
        if (tid < DP) {
            const uint32_t *mr = reinterpret_cast<const uint32_t *>(ms + tid * TN);
#pragma unroll
            for (int q = 0; q < TN / 4; ++q) rowobs += __popc(mr[q]);
        }
        // bit words: 2 subtiles x 64 lanes x 2 layouts = 256 words per tile, one per thread
        {
            const int lay = tid >> 7, s = (tid >> 6) & 1, l = tid & 63;
            const int l15 = l & 15, l4 = l >> 4;
            uint32_t w = 0;
            if (lay == 0) {
                for (int q = 0; q < DP / 4; ++q) w |= (uint32_t)ms[(4 * q + l4) * TN + 16 * s + l15] << q;
                Mb1[(tile * 2 + s) * 64 + l] = w;
            } else {
                for (int dt = 0; dt < DP / 16; ++dt)
                    for (int qq = 0; qq < 4; ++qq)
                        w |= (uint32_t)ms[(16 * dt + l15) * TN + 16 * s + 4 * qq + l4] << (4 * dt + qq);
                Mb2[(tile * 2 + s) * 64 + l] = w;
            }
        }
    }
    syy = block_sum<NT>(syy, red);
    nobs = block_sum<NT>(nobs, red);
    if (tid == 0) {
        partial[2 * blockIdx.x] = syy;
        partial[2 * blockIdx.x + 1] = nobs;
    }
    if (tid < DP) partial[2 * (int64_t)gridDim.x + (int64_t)blockIdx.x * DP + tid] = (double)rowobs;
}

// out[j] (+)= sum_b partial[b * stride + j], fixed order (deterministic).  A workgroup owns KX
// neighbouring outputs; its NT / KX row-lanes split the partial blocks (eight loads in flight each)
// and meet in LDS.  (With a thread per output the 1024 partial blocks of the sweep stage were summed
// one dependent load after the other: 0.4-0.5 ms per chunk on the critical path.)
template <int KX>
__global__ void __launch_bounds__(NT)
mpca_reduce_kernel(const double *__restrict__ partial, int nb, int64_t stride, int64_t len,
                   double *__restrict__ out, int accumulate)
{
    constexpr int RY = NT / KX;
    __shared__ double tile[RY][KX + 1];
    const int kx = threadIdx.x % KX, ry = threadIdx.x / KX;
    const int64_t e = (int64_t)blockIdx.x * KX + kx;
    double acc = 0.0;
    if (e < len) {
        int b = ry;
        for (; b + 7 * RY < nb; b += 8 * RY) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * RY) * stride + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; b < nb; b += RY) acc += partial[(int64_t)b * stride + e];
    }
    tile[ry][kx] = acc;
    __syncthreads();
    if (ry == 0 && e < len) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < RY; ++j) s += tile[j][kx];
        out[e] = accumulate ? out[e] + s : s;
    }
}

void launch_reduce(hipStream_t s, const double *partial, int nb, int64_t stride, int64_t len,
                   double *out, int accumulate)
{
    if (len >= 16384)
        hipLaunchKernelGGL(mpca_reduce_kernel<64>, dim3((unsigned)((len + 63) / 64)), dim3(NT), 0, s,
                           partial, nb, stride, len, out, accumulate);
    else
        hipLaunchKernelGGL(mpca_reduce_kernel<16>, dim3((unsigned)((len + 15) / 16)), dim3(NT), 0, s,
                           partial, nb, stride, len, out, accumulate);
}

// -------------------------------------------------------------------------------------------
// mpca_lambda: Lam~[n][:] = sum_d m_dn panel[d][:]   (columns: packed <ww>, then <w>)
//   A operand (n x d): mask bits -> 0.0 / 1.0 (packed columns), m*y from Ymt (the KT last columns)
//   B operand (d x col): the panel in fragment order, 16 bytes per lane = two k-steps
// One workgroup = 16*NSUB plates; wavefront w owns column tiles c = w, w+4, ...
// -------------------------------------------------------------------------------------------
// The accumulation of ONE group of 16 NSUB plates by the four wavefronts of a workgroup (wavefront
// w: column tiles c = w, w + 4, ...), handed to ``store(c, s, R, wt, value)`` in the result layout
// of the instructions: the value is entry (plate 16 s + (wt ? l4 + 4 R : 4 R + l4), column
// 16 c + (l & 15)) of the group (l4 = l >> 4; wt = the <w> tile of the 16x16x4 form).  Shared by
// mpca_lambda_kernel (-> Lam in HBM) and the fused form of mpca_blk4_kernel (-> its staging area).
template <int DB, int KT, int NSUB, typename ST>
__device__ __forceinline__ void
mpca_lambda_group(const double *__restrict__ Ymt, const uint32_t *__restrict__ Mb1,
                  const double *__restrict__ panel, int64_t sub0, int64_t nsub_chunk, int64_t grp,
                  ST &&store)
{
    constexpr int DP = 32 * DB, DQ = DP / 4;
    constexpr int KP = 16 * KT, P = KP * (KP + 1) / 2, PT = (P + 15) / 16, CT = PT + KT;
    constexpr int LR = 16 * CT;
    constexpr int CW = (CT + 3) / 4;            // column tiles per wavefront (upper bound)
    // Column tile c = w + 4 ci.  The slots ci < CW - 1 always hold packed-<ww> tiles (A operand =
    // the mask); only the LAST slot differs between the wavefronts: a packed tile, a <w> tile
    // (A operand = m*y) or nothing.  The kind is fixed per wavefront, so the k-loop is compiled
    // three times and each copy is straight-line code (with the tests inside the loop every MFMA
    // carried a scalar branch or two selects: 5.8 vector instructions per MFMA, which on this chip
    // add to the matrix time).
    static_assert(4 * (CW - 1) <= PT && PT + KT <= 4 * CW, "the <w> tiles sit in the last slot");
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l4 = l >> 4;
    const int c_last = w + 4 * (CW - 1);
    const int kind = c_last >= CT ? 0 : (c_last >= PT ? 2 : 1);
    // B fragments: panel + ((c (DQ/2) + q2) 64 + l) 2 doubles; the lane part is the only VGPR term
    // (a 32-bit offset beside a scalar base: 64-bit vector addresses cost two registers per tile)
    const char *pl = reinterpret_cast<const char *>(panel);
    const uint32_t loff = (uint32_t)l * 16u;
    constexpr size_t TILE_B = (size_t)(DQ / 2) * 64 * 16;      // bytes of one column tile
    // v_mfma_f64_4x4x4_4b_f64 (round 3): the four blocks are the four 4-column groups of a column
    // tile (B operand unchanged: lane = column + 16 k), the A operand is the mask of the plate
    // group R (plates 16 sub + 4 R + i) replicated over the blocks = the mask word of lane
    // 4 R + i + 16 k, and a 16 x 16 result tile is four registers, one per R.
    int src[4];
#pragma unroll
    for (int R = 0; R < 4; ++R) src[R] = 4 * ((l & 0x30) | (4 * R) | (l & 3));
    {
        double acc[CW][NSUB][4];
#pragma unroll
        for (int ci = 0; ci < CW; ++ci)
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
#pragma unroll
                for (int R = 0; R < 4; ++R) acc[ci][s][R] = 0.0;
        uint32_t mr[NSUB][4], mown[NSUB];
        const double *yb[NSUB];
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const bool ok = (grp * NSUB + s) < nsub_chunk;
            // global 16-plate subtile (one beyond the chunk: any valid one, its mask word is zero)
            const int64_t sub = sub0 + (ok ? grp * NSUB + s : grp * NSUB);
            const uint32_t mw = ok ? Mb1[sub * 64 + l] : 0u;
            mown[s] = mw;
#pragma unroll
            for (int R = 0; R < 4; ++R)
                mr[s][R] = (uint32_t)__builtin_amdgcn_ds_bpermute(src[R], (int)mw);
            // Ymt element (d = 4q + l4, n = 16 sub + l15): tile = sub/2, column (sub&1)*16 + l15
            yb[s] = Ymt + (sub >> 1) * ((int64_t)DP * TN) + (int64_t)l4 * TN + (sub & 1) * 16 + (l & 15);
        }
        auto kloop = [&](auto KIND) {
            constexpr int LASTK = decltype(KIND)::value;          // 0 none, 1 mask, 2 m*y
            constexpr int NC = LASTK ? CW : CW - 1;
            constexpr int NC4 = LASTK == 1 ? CW : CW - 1;         // slots on the 4x4x4 instruction
            const char *pw = pl + (size_t)w * TILE_B;
            // ONE buffer of B fragments, refilled slot by slot: within a step (two k-steps) the
            // column slots are the outer loop, so the fragment of slot ci is dead after its 16
            // MFMAs and the fragment of the NEXT step is requested into the same registers right
            // there -- it has the other slots' MFMAs (~0.9 us) to arrive.  (Two alternating buffers
            // spill: 144 accumulator + 2 x 36 fragment registers; and left to the compiler the
            // loads of a step sit in front of their first use and every step pays the L2 latency.)
            v2f64 bf[NC];
            double yc[2][NSUB];
            v4f64 accw[NSUB];                   // a <w> tile (last slot, kind 2): 16x16x4 form
#pragma unroll
            for (int s = 0; s < NSUB; ++s) accw[s] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ci = 0; ci < NC; ++ci)
                bf[ci] = *reinterpret_cast<const v2f64 *>(pw + (size_t)(4 * ci) * TILE_B + loff);
            if (LASTK == 2) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int s = 0; s < NSUB; ++s) yc[e][s] = yb[s][(int64_t)(4 * e) * TN];
            }
#pragma unroll 1
            for (int q2 = 0; q2 < DQ / 2; ++q2) {
                const int q2n = q2 + 1 < DQ / 2 ? q2 + 1 : q2;
                double am[2][NSUB][4];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int s = 0; s < NSUB; ++s)
#pragma unroll
                        for (int R = 0; R < 4; ++R)
                            am[e][s][R] = (double)((mr[s][R] >> (2 * q2 + e)) & 1u);
#pragma unroll
                for (int ci = 0; ci < NC4; ++ci) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int s = 0; s < NSUB; ++s)
#pragma unroll
                            for (int R = 0; R < 4; ++R)
                                acc[ci][s][R] = mfma4(am[e][s][R], e ? bf[ci].y : bf[ci].x,
                                                      acc[ci][s][R]);
                    __builtin_amdgcn_sched_barrier(0);
                    bf[ci] = *reinterpret_cast<const v2f64 *>(pw + (size_t)(4 * ci) * TILE_B
                                                              + (size_t)q2n * (64 * 16) + loff);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (LASTK == 2) {
                    // masked entries of Ymt are zero already; subtiles beyond the chunk have a zero
                    // mask word and must not contribute either
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int s = 0; s < NSUB; ++s) {
                            const double al = ((mown[s] >> (2 * q2 + e)) & 1u) ? yc[e][s] : 0.0;
                            accw[s] = mfma(al, e ? bf[CW - 1].y : bf[CW - 1].x, accw[s]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    bf[CW - 1] = *reinterpret_cast<const v2f64 *>(pw + (size_t)(4 * (CW - 1)) * TILE_B
                                                                  + (size_t)q2n * (64 * 16) + loff);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int s = 0; s < NSUB; ++s)
                            yc[e][s] = yb[s][(int64_t)(4 * (2 * q2n + e)) * TN];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (LASTK == 2) {
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[CW - 1][s][r] = accw[s][r];
            }
        };
        if (kind == 2) kloop(std::integral_constant<int, 2>{});
        else if (kind == 1) kloop(std::integral_constant<int, 1>{});
        else kloop(std::integral_constant<int, 0>{});
        // result layout of the 4x4x4 instruction: row (plate) 4 R + (l >> 4), column l & 15 of the
        // tile; of the 16x16x4 one: row (l >> 4) + 4 r
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) {
            const int c = w + 4 * ci;
            if (c < CT) {
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
#pragma unroll
                    for (int R = 0; R < 4; ++R)
                        store(c, s, R, (ci == CW - 1) && kind == 2, acc[ci][s][R]);
            }
        }
    }
}

template <int DB, int KT, int NSUB>
__global__ void __launch_bounds__(NT, 2)
mpca_lambda_kernel(const double *__restrict__ Ymt, const uint32_t *__restrict__ Mb1,
                   const double *__restrict__ panel, int64_t sub0, int64_t nsub_chunk,
                   int64_t nplates_chunk, double *__restrict__ Lam)
{
    constexpr int KP = 16 * KT, P = KP * (KP + 1) / 2, PT = (P + 15) / 16, LR = 16 * (PT + KT);
    const int l = threadIdx.x & 63, l4 = l >> 4;
    const int64_t ngroups = (nsub_chunk + NSUB - 1) / NSUB;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x)
        mpca_lambda_group<DB, KT, NSUB>(Ymt, Mb1, panel, sub0, nsub_chunk, grp,
                                        [&](int c, int s, int R, bool wt, double v) {
            const int64_t n = (grp * NSUB + s) * 16 + (wt ? l4 + 4 * R : 4 * R + l4);
            if (n < nplates_chunk) __builtin_nontemporal_store(v, &Lam[n * LR + 16 * c + (l & 15)]);
        });
}

// -------------------------------------------------------------------------------------------
// mpca_sweep: per plate  Lam_n = c I + tau Lam~_n  ->  Cov_n, <x_n>, <xx>_n, log|Cov_n|, tr<xx>_n
// One wavefront per plate (two plates in flight per wavefront to hide the serial pivot chains).
// FROM_VALUE: delta moments of a given <x> (initialize_from_value): <xx> = x x^T, no inverse.
// -------------------------------------------------------------------------------------------
// The packed row of one plate as this lane needs it: 16 matrix elements (the 2 x 2 accumulator
// tiles) and its 8 right-hand-side entries.
struct plate_raw {
    double a[2][2][4];
};

template <int KT>
__device__ __forceinline__ void load_raw(plate_raw &q, const double *row, int K, int l15, int l4)
{
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tr + l4 + 4 * r, j = 16 * tc + l15;
                const int a = i > j ? i : j, b = i > j ? j : i;
                q.a[tr][tc][r] = (i < K && j < K) ? row[tri(a, b)] : 0.0;
            }
}

// One packed row (LR doubles, 16-byte aligned) as whole 16-byte lanes: the copy of the NEXT
// plate's row that a wavefront keeps in flight while it sweeps the current one.
template <int LR>
struct row_copy {
    static constexpr int NV = (LR / 2 + 63) / 64;
    v2f64 v[NV];
    __device__ __forceinline__ void load(const double *__restrict__ row, int l)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int pidx = l + 64 * i;
            v[i] = (pidx < LR / 2) ? __builtin_nontemporal_load(
                                         reinterpret_cast<const v2f64 *>(row) + pidx)
                                   : v2f64{0.0, 0.0};
        }
    }
    __device__ __forceinline__ void store(double *stage, int l) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int pidx = l + 64 * i;
            if (pidx < LR / 2) reinterpret_cast<v2f64 *>(stage)[pidx] = v[i];
        }
    }
};

__device__ __forceinline__ void tiles_from_raw(v4f64 (&T)[2][2], const plate_raw &q, int K,
                                               double diag, double scale, int l15, int l4)
{
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tr + l4 + 4 * r, j = 16 * tc + l15;
                T[tr][tc][r] = (i < K && j < K) ? scale * q.a[tr][tc][r] + ((i == j) ? diag : 0.0)
                                                : ((i == j) ? 1.0 : 0.0);
            }
}

struct plate_result {
    double x0, x1;        // <x>[l15], <x>[16 + l15] in the lanes l4 == 0
    double logdet;        // log|Lam|
    int bad;
};

// T (= Lam) -> T = Cov + x x^T, x = Cov rhs
__device__ __forceinline__ plate_result finish_plate(v4f64 (&T)[2][2], const double *pb,
                                                     double scale, int K, int l15, int l4,
                                                     double prod, double ld, int bad)
{
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tr + l4 + 4 * r, j = 16 * tc + l15;
                T[tr][tc][r] = (i < K && j < K) ? -T[tr][tc][r] : T[tr][tc][r];
            }
    // x = Cov rhs on the vector ALU: this lane holds rows {16 tr + 4 r + l4} of the columns
    // {l15, 16 + l15}; eight products per column, then the four lane groups (l4) are summed with
    // two lane exchanges.  (On the matrix core the same product takes 16 MFMAs with one useful
    // row of 16 each.)
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * tr + 4 * r + l4;
            const double h = (m < K) ? scale * pb[m] : 0.0;
            p0 += h * T[tr][0][r];
            p1 += h * T[tr][1][r];
        }
    p0 += __shfl_xor(p0, 16, 64);
    p1 += __shfl_xor(p1, 16, 64);
    p0 += __shfl_xor(p0, 32, 64);
    p1 += __shfl_xor(p1, 32, 64);
    plate_result res;
    res.x0 = (l4 == 0) ? p0 : 0.0;
    res.x1 = (l4 == 0) ? p1 : 0.0;
    T[0][0] = mfma(res.x0, res.x0, T[0][0]);
    T[0][1] = mfma(res.x0, res.x1, T[0][1]);
    T[1][0] = mfma(res.x1, res.x0, T[1][0]);
    T[1][1] = mfma(res.x1, res.x1, T[1][1]);
    res.logdet = sweep_logdet(prod, ld);
    res.bad = bad;
    return res;
}

template <int KT>
__device__ __forceinline__ void store_plate(const v4f64 (&T)[2][2], const plate_result &res,
                                            int64_t n_chunk, int64_t n_glob, int K,
                                            double *__restrict__ XXf, double *__restrict__ Xm,
                                            int l15, int l4, double &trl, double *sa)
{
    constexpr int KP = 16 * KT, P = KP * (KP + 1) / 2, PT = (P + 15) / 16;
    // sum over the plates of <x x^T>_n (RotateGaussianARD needs it, transformations.py:476-640):
    // this wavefront's K x K accumulator lives in LDS (every element has one owner lane, so
    // read-add-write needs no atomics); in registers it cost 32 VGPRs of a kernel that is bound
    // by the number of plates it can keep in flight
#pragma unroll
    for (int tr = 0; tr < KT; ++tr)
#pragma unroll
        for (int tc = 0; tc < KT; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tr + l4 + 4 * r, j = 16 * tc + l15;
                sa[i * KP + j] += T[tr][tc][r];
            }
    double *xb = XXf + xxf_base(n_chunk, PT);
#pragma unroll
    for (int tr = 0; tr < KT; ++tr)
#pragma unroll
        for (int tc = 0; tc <= tr; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tr + l4 + 4 * r, j = 16 * tc + l15;
                if (i >= j) {
                    const int p = tri(i, j);
                    const double v = (i < K && j < K) ? T[tr][tc][r] : 0.0;
                    xb[xxf_off(p)] = v;
                    if (i == j) trl += v;
                }
            }
    if (l4 == 0 && Xm) {
        double *xr = Xm + n_glob * KP;
        xr[l15] = (l15 < K) ? res.x0 : 0.0;
        if (KT > 1) xr[16 + l15] = (16 + l15 < K) ? res.x1 : 0.0;
    }
}

// NM = plates in flight per wavefront (their sweeps interleaved in one instruction stream: the
// serial pivot chain of one fills the latency gaps of the others), OCC = wavefronts per SIMD the
// register allocation is held to.
template <int KT, bool FROM_VALUE, int NM, int OCC>
__global__ void __launch_bounds__(NT, OCC)
mpca_sweep_kernel(const double *__restrict__ Lam, int LR, int64_t n0, int64_t nplates_chunk, int K,
                  double x_prec, double xx_diag, const double *__restrict__ tau_ptr,
                  double *__restrict__ XXf, double *__restrict__ Xm, int write_x,
                  double *__restrict__ partial, double *__restrict__ partial_sxx)
{
    constexpr int KP = 16 * KT, P = KP * (KP + 1) / 2, PT = (P + 15) / 16;
    constexpr int LRC = 16 * (PT + KT);
    __shared__ double red[NT / 64];
    __shared__ double sxs[4 * KP * KP];
    __shared__ __attribute__((aligned(16))) double stg[4][NM][LRC];
    double *Xw = write_x ? Xm : nullptr;
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = l & 15, l4 = l >> 4;
    const double tau = FROM_VALUE ? 0.0 : tau_ptr[0];
    const int nblocks = (K + 3) / 4;
    double trl = 0.0, ldsum = 0.0;
    int anybad = 0;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    double *sa = sxs + w * KP * KP;          // this wavefront's sum of <x x^T>_n
    for (int e = l; e < KP * KP; e += 64) sa[e] = 0.0;
    lds_fence();
    if (FROM_VALUE) {
        for (int64_t n = (int64_t)blockIdx.x * 4 + w; n < nplates_chunk; n += nwaves) {
            const double *xr = Xm + (n0 + n) * KP;
            // <xx> = x x^T through the same store path: T = outer product
            v4f64 T[2][2];
            const double xa = (l4 == 0 && l15 < K) ? xr[l15] : 0.0;
            const double xb = (KT > 1 && l4 == 0 && 16 + l15 < K) ? xr[16 + l15] : 0.0;
            const v4f64 z = {0.0, 0.0, 0.0, 0.0};
            T[0][0] = mfma(xa, xa, z);
            T[0][1] = mfma(xa, xb, z);
            T[1][0] = mfma(xb, xa, z);
            T[1][1] = mfma(xb, xb, z);
            if (xx_diag != 0.0) {      // prior moments: <xx> = I / c + x x^T
#pragma unroll
                for (int tr = 0; tr < 2; ++tr)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * tr + l4 + 4 * r == 16 * tr + l15) T[tr][tr][r] += xx_diag;
            }
            plate_result res;
            res.x0 = xa;
            res.x1 = xb;
            res.logdet = 0.0;
            res.bad = 0;
            store_plate<KT>(T, res, n, n0 + n, K, XXf, Xw, l15, l4, trl, sa);
        }
    } else {
        // A wavefront works on NM plates at a time, their sweeps interleaved in one instruction
        // stream (the serial pivot chain of one fills the latency gaps of the other).  The packed
        // rows of the NEXT NM plates are copied HBM -> registers (16-byte lanes, coalesced) while
        // the current ones are swept, parked in LDS at the end of the iteration and gathered
        // from there: without the prefetch every plate paid the full memory latency before its
        // sweep started (measured: 2.6 of 9.0 ms per 2^20 plates, plus 1 ms for the right-hand
        // side that was fetched after the sweep).
        row_copy<LRC> rc[NM];
        int64_t n = ((int64_t)blockIdx.x * 4 + w) * NM;
#pragma unroll
        for (int m = 0; m < NM; ++m)
            if (n + m < nplates_chunk) {
                rc[m].load(Lam + (n + m) * LRC, l);
                rc[m].store(stg[w][m], l);
            }
        for (; n < nplates_chunk; n += nwaves * NM) {
            const int64_t nn = n + nwaves * NM;
#pragma unroll
            for (int m = 0; m < NM; ++m)
                if (nn + m < nplates_chunk) rc[m].load(Lam + (nn + m) * LRC, l);
            lds_fence();
            v4f64 T[NM][2][2];
            double pr[NM], lg[NM];
            int bd[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                plate_raw cur;
                // a plate beyond the end re-does the previous content of its slot; not stored
                load_raw<KT>(cur, stg[w][m], K, l15, l4);
                tiles_from_raw(T[m], cur, K, x_prec, tau, l15, l4);
                pr[m] = 1.0;
                lg[m] = 0.0;
                bd[m] = 0;
            }
            sweep_multi<0, NM>(T, nblocks, l15, l4, pr, lg, bd);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const plate_result A = finish_plate(T[m], stg[w][m] + 16 * PT, tau, K, l15, l4,
                                                    pr[m], lg[m], bd[m]);
                if (n + m < nplates_chunk) {
                    store_plate<KT>(T[m], A, n + m, n0 + n + m, K, XXf, Xw, l15, l4, trl, sa);
                    ldsum -= A.logdet;
                    anybad |= A.bad;
                }
            }
            lds_fence();
#pragma unroll
            for (int m = 0; m < NM; ++m)
                if (nn + m < nplates_chunk) rc[m].store(stg[w][m], l);
        }
    }
    // per-workgroup partial sums: tr<xx>, log|Cov| (uniform per wavefront: count once), status
    const double tr = block_sum<NT>(trl, red);
    const double ldw = block_sum<NT>(l == 0 ? ldsum : 0.0, red);
    const double bd = block_sum<NT>((double)anybad, red);
    if (threadIdx.x == 0) {
        partial[3 * blockIdx.x + 0] = tr;
        partial[3 * blockIdx.x + 1] = ldw;
        partial[3 * blockIdx.x + 2] = bd;
    }
    // sum_n <xx>_n of this workgroup: the four wavefronts' LDS accumulators, fixed order
    __syncthreads();
    for (int e = threadIdx.x; e < KP * KP; e += NT) {
        const int i = e / KP, j = e - i * KP;
        const double v = (sxs[e] + sxs[KP * KP + e]) + (sxs[2 * KP * KP + e] + sxs[3 * KP * KP + e]);
        partial_sxx[(int64_t)blockIdx.x * KP * KP + e] = (i < K && j < K) ? v : 0.0;
    }
}

// -------------------------------------------------------------------------------------------
// mpca_rows: the same stage for 16 < K <= 32 on the vector ALU only -- a MEASURED ALTERNATIVE
// (vmp_tune_set("mpca_rows", 1)), not the default.  The sweep form above spends ~2,400
// instructions per plate, most of them on the 4 x 4 pivot blocks, and fp64 vector and matrix
// instructions do not overlap on this chip.  Here a group of 16 lanes (one DPP row) owns a plate,
// lane i of the group holds rows i and 16 + i of the 32 x 32 matrix in 128 VGPRs, and a
// Gauss-Jordan step is: the pivot lane scales its row, every element of that row is broadcast to
// the group with v_mov_b64_dpp row_newbcast (one instruction, no LDS), and each lane updates its
// two rows (64 FMAs): ~150 instructions per step, 32 steps, FOUR plates per wavefront = 1,375
// instructions per plate including gather, <x>, <x x^T> and the stores (counted: SQ_INSTS_VALU).
// One wavefront per workgroup (4 x 4.4 KB of staged rows + the packed K x K accumulator = 22 KB
// of LDS: seven wavefronts per CU).  Result: bit-identical bound, 5.9-6.8 ms per 2^20 plates
// against 5.8 ms of the sweep form -- the vector ALU issues only 58 % of the time (1.75
// wavefronts per SIMD, half of their cycles in s_waitcnt: LDS gather, the HBM round trip of the
// next rows, a 45 KB loop body in the instruction cache).  What was needed to get there: the next
// rows are fetched AFTER the elimination (held across it, their 36 registers spill the matrix, and
// a spill reload waits for every outstanding store), the outer-product FMAs are kept out of one
// big `if (valid)` block (the compiler sinks them next to the stores and keeps 32 broadcasts
// alive), per-lane index bases are re-materialised every iteration (otherwise 128 offsets are
// hoisted), the logarithm is taken once per wavefront.
// -------------------------------------------------------------------------------------------
template <int LANE>
__device__ __forceinline__ double row_bcast(double x)
{
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + LANE, 0xf, 0xf, true);
}

template <int P>
__device__ __forceinline__ void rows2_steps(double (&m)[2][32], int gl, int K, double &prod,
                                            double &ld, int &bad)
{
    if constexpr (P < 32) {
        if (P < K) {
            constexpr int HP = P >> 4, LP = P & 15;
            const double piv = row_bcast<LP>(m[HP][P]);
            if (!(piv > 0.0)) bad = 1;
            // running determinant as mantissa x 2^exponent
            const double q = prod * piv;
            ld += (double)__builtin_amdgcn_frexp_exp(q);
            prod = __builtin_amdgcn_frexp_mant(q);
            const double d = fast_recip3(piv);
            const bool isp = (gl == LP);
            // multipliers of this lane's two rows (the pivot row itself is only scaled)
            double c0 = m[0][P], c1 = m[1][P];
            if (HP == 0) c0 = isp ? 0.0 : c0;
            else c1 = isp ? 0.0 : c1;
            // the pivot lane scales its pivot row by 1 / piv (the other lanes multiply by one: no
            // select, no branch); the pivot element becomes 1 / piv, and column p of every other
            // row becomes 0 - c_i * (1 / piv) in the loop below
            const double dd = isp ? d : 1.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                m[HP][j] *= dd;
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            m[HP][P] = isp ? d : 0.0;
            m[1 - HP][P] = 0.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const double r = row_bcast<LP>(m[HP][j]);
                m[0][j] = __builtin_fma(-c0, r, m[0][j]);
                m[1][j] = __builtin_fma(-c1, r, m[1][j]);
                // keep broadcast and use together: hoisting the 32 broadcasts of a step costs 64
                // VGPRs on top of the 128 of the matrix
                if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        rows2_steps<P + 1>(m, gl, K, prod, ld, bad);
    }
}

template <int J>
__device__ __forceinline__ void rows2_outer(double (&m)[2][32], double x0, double x1)
{
    if constexpr (J < 32) {
        const double xj = row_bcast<(J & 15)>(J < 16 ? x0 : x1);
        m[0][J] = __builtin_fma(x0, xj, m[0][J]);
        m[1][J] = __builtin_fma(x1, xj, m[1][J]);
        if ((J & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        rows2_outer<J + 1>(m, x0, x1);
    }
}

constexpr int RW_NT = 64;

template <bool FULLK>
__global__ void __launch_bounds__(RW_NT, 2)
mpca_rows_kernel(const double *__restrict__ Lam, int64_t n0, int64_t nplates_chunk, int K,
                 double x_prec, const double *__restrict__ tau_ptr, double *__restrict__ XXf,
                 double *__restrict__ Xm, int write_x, double *__restrict__ partial,
                 double *__restrict__ partial_sxx)
{
    constexpr int KP = 32, P = KP * (KP + 1) / 2, PT = (P + 15) / 16, LRC = 16 * (PT + 2);
    constexpr int NPAIR = 4 * LRC / 2, NV = (NPAIR + 63) / 64;
    __shared__ __attribute__((aligned(16))) double stg[4 * LRC];
    __shared__ double sa[P];                     // packed sum of <x x^T>_n of this workgroup
    const int l = threadIdx.x, gl = l & 15, grp = l >> 4;
    const double tau = tau_ptr[0];
    const int i0c = gl, i1c = 16 + gl;
    const int b0c = tri(i0c, 0), b1c = tri(i1c, 0);
    for (int e = l; e < P; e += RW_NT) sa[e] = 0.0;
    double pm = 1.0, le = 0.0;                   // product of the pivots of this group's plates
    int anybad = 0;
    v2f64 nxt[NV];
    auto fetch = [&](int64_t n) {
        const int64_t left = nplates_chunk - n;
        const v2f64 *src = reinterpret_cast<const v2f64 *>(Lam + n * LRC);
        if (left >= 4) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int pidx = l + 64 * i;
                nxt[i] = (i < NV - 1 || pidx < NPAIR) ? __builtin_nontemporal_load(src + pidx)
                                                      : v2f64{0.0, 0.0};
            }
        } else {
            const int npair = (int)(left * (LRC / 2));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int pidx = l + 64 * i;
                nxt[i] = (pidx < npair) ? __builtin_nontemporal_load(src + pidx) : v2f64{0.0, 0.0};
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int pidx = l + 64 * i;
            if (pidx < NPAIR) reinterpret_cast<v2f64 *>(stg)[pidx] = nxt[i];
        }
    };
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t n = (int64_t)blockIdx.x * 4;
    if (n < nplates_chunk) {
        fetch(n);
        park();
    }
    for (; n < nplates_chunk; n += step) {
        lds_fence();
        // the per-lane index bases are re-materialised every iteration: left loop-invariant, the
        // compiler hoists the 64 gather offsets and the 64 store offsets of a lane out of the loop
        // (192 VGPRs) and spills the matrix instead
        int i0 = i0c, i1 = i1c, b0 = b0c, b1 = b1c;
        asm volatile("" : "+v"(i0), "+v"(i1), "+v"(b0), "+v"(b1));
        const double *row = stg + grp * LRC;
        const bool valid = n + grp < nplates_chunk;
        // ---- gather: m[h][j] = c delta_ij + tau Lam~[i][j], i = 16 h + gl ------------------------
        double m[2][32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int cj = tri(j, 0);
            const double v0 = row[(i0 >= j) ? b0 + j : cj + i0];
            const double u1 = row[(i1 >= j) ? b1 + j : cj + i1];
            double a0 = __builtin_fma(tau, v0, (j == i0) ? x_prec : 0.0);
            double a1 = __builtin_fma(tau, u1, (j == i1) ? x_prec : 0.0);
            if (!FULLK) {
                a0 = (i0 < K && j < K) ? a0 : ((j == i0) ? 1.0 : 0.0);
                a1 = (i1 < K && j < K) ? a1 : ((j == i1) ? 1.0 : 0.0);
            }
            m[0][j] = a0;
            m[1][j] = a1;
            if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        double prod = 1.0, ld = 0.0;
        int bad = 0;
        rows2_steps<0>(m, gl, K, prod, ld, bad);
        // ---- <x> = Cov (tau rhs), <x x^T> = Cov + <x><x>^T ----------------------------------------
        double x0 = 0.0, x1 = 0.0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const double hj = row[16 * PT + j];
            x0 = __builtin_fma(m[0][j], hj, x0);
            x1 = __builtin_fma(m[1][j], hj, x1);
            if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        x0 *= tau;
        x1 *= tau;
        rows2_outer<0>(m, x0, x1);
        // ---- stores (predicated per element: inside one `if (valid)` block the compiler sinks the
        //      64 outer-product FMAs next to the stores and keeps all 32 broadcasts alive) ------------
        {
            const int64_t nc = n + grp;
            double *xb = XXf + xxf_base(nc, PT);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (valid && j <= i0) {
                    const int p = b0 + j;
                    const double v = (FULLK || i0 < K) ? m[0][j] : 0.0;
                    xb[xxf_off(p)] = v;
                    __hip_atomic_fetch_add(&sa[p], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (valid && j <= i1) {
                    const int p = b1 + j;
                    const double v = (FULLK || (i1 < K && j < K)) ? m[1][j] : 0.0;
                    xb[xxf_off(p)] = v;
                    __hip_atomic_fetch_add(&sa[p], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (valid && write_x) {
                double *xr = Xm + (n0 + nc) * KP;
                xr[i0] = x0;
                xr[i1] = x1;
            }
            if (valid) {
                // log|Lam_n| = log(prod) + ld ln 2: mantissas and exponents are accumulated, the
                // logarithm is taken once per wavefront
                const double q = pm * prod;
                le += ld + (double)__builtin_amdgcn_frexp_exp(q);
                pm = __builtin_amdgcn_frexp_mant(q);
                anybad |= bad;
            }
        }
        lds_fence();
        // the next four packed rows: HBM -> registers -> LDS at the end of the iteration (the
        // other wavefront of the SIMD runs meanwhile); held across the elimination, their 36
        // registers spill the matrix, and a spill reload waits for every outstanding store
        if (n + step < nplates_chunk) {
            fetch(n + step);
            park();
        }
    }
    lds_fence();
    // per-workgroup partials: tr<xx> = trace of the accumulated sum, log|Cov|, status
    double tr = (l < K) ? sa[tri(l, l)] : 0.0;
    tr = wave_sum(tr);
    const double ldw = wave_sum(gl == 0 ? -(log(pm) + le * 0.69314718055994530942) : 0.0);
    const double bd = wave_sum((double)anybad);
    if (l == 0) {
        partial[3 * blockIdx.x + 0] = tr;
        partial[3 * blockIdx.x + 1] = ldw;
        partial[3 * blockIdx.x + 2] = bd;
    }
    for (int e = l; e < KP * KP; e += RW_NT) {
        const int i = e / KP, j = e - i * KP;
        const int a = i > j ? i : j, b = i > j ? j : i;
        partial_sxx[(int64_t)blockIdx.x * KP * KP + e] = (i < K && j < K) ? sa[tri(a, b)] : 0.0;
    }
}

// -------------------------------------------------------------------------------------------
// mpca_blk4: the per-plate stage on v_mfma_f64_4x4x4_4b_f64 -- the DEFAULT for the real update
// (round 3).  That instruction multiplies FOUR independent 4 x 4 x 4 blocks per wavefront and is
// the cheapest fp64 flop on this chip (tools/mfma4_lab.hip, profiles/r03/mfma4_lab.txt: 7.2 ns per
// instruction and SIMD = 71 flop/ns against 59 for v_fma_f64 and 46 for the 16x16x4 form).  A
// wavefront therefore inverts four plates at a time, one per block of the instruction, with the
// K x K matrix of a plate cut into 4 x 4 blocks of which only the lower block triangle is kept
// (36 registers for K = 32 instead of the 64 of a full matrix): lane l of a wavefront holds
// element (4 I + (l >> 4), 4 J + (l & 3)) of block (I, J) of plate (l >> 2) & 3 -- the result
// layout of the instruction (D[b][i][j] in lane j + 4 b + 16 i; probed, see the lab).  A register
// X in that layout read as the B operand is X, read as the A operand it is X^T, so
// mfma4(X, Y, Z) = X^T Y + Z and the symmetric sweep with 4 x 4 pivot blocks is, per pivot p:
//     E = A_pp^-1                                  the only vector-ALU part (LDL^T, uniform per plate;
//                                                  its 10 entries reach all 16 lanes of the plate by
//                                                  4 MFMAs with a row selector + quad broadcasts)
//     C~_I = A_pI (I < p, stored) or A_Ip^T (I > p: one MFMA with the identity)
//     Q~_I = -E C~_I                               NB - 1 MFMAs
//     A_IJ += Q~_I^T C~_J = A_IJ - A_Ip E A_pJ     for I >= J, both != p: NB (NB - 1) / 2 MFMAs
//     A_pI = -Q~_I (I < p),  A_Ip = C~_I^T E (I > p),  A_pp = -E
// Whole blocks are updated, so the symmetry of the matrix halves the work for free (K^3 flops),
// and the pivot-block algebra is paid once per FOUR plates: ~300 vector + 370 matrix instructions
// per wavefront-iteration = 170 per plate against ~2,400 of the 16x16x4 sweep form above.
// <x> = Cov rhs: the half of the product that contracts the row index of a stored block runs on
// the matrix core (36 MFMAs), the other half on the vector ALU (28 FMAs + quad reductions);
// <x x^T> = Cov + <x><x>^T with the column form of <x> from one more transpose per block row.
// -------------------------------------------------------------------------------------------
__host__ __device__ constexpr int bidx(int I, int J) { return I * (I + 1) / 2 + J; }

// One sweep over pivot block p of the NB x NB block matrix S (lower block triangle).
template <int NB, int p>
__device__ __forceinline__ void blk4_step(double (&S)[NB * (NB + 1) / 2], int li, int lj,
                                          const double (&sel)[4], double ident, double &prod,
                                          double &ld, int &bad)
{
    // ---- the pivot block in every lane of its plate: rw[a][i][j] = D[a][j], then the quad ------
    const double Dp = S[bidx(p, p)];
    const double rw0 = mfma4(sel[0], Dp, 0.0), rw1 = mfma4(sel[1], Dp, 0.0);
    const double rw2 = mfma4(sel[2], Dp, 0.0), rw3 = mfma4(sel[3], Dp, 0.0);
    const double d00 = dpp_f64<0x00>(rw0), d01 = dpp_f64<0x55>(rw0), d02 = dpp_f64<0xAA>(rw0);
    const double d03 = dpp_f64<0xFF>(rw0);
    const double d11 = dpp_f64<0x55>(rw1), d12 = dpp_f64<0xAA>(rw1), d13 = dpp_f64<0xFF>(rw1);
    const double d22 = dpp_f64<0xAA>(rw2), d23 = dpp_f64<0xFF>(rw2);
    const double d33 = dpp_f64<0xFF>(rw3);
    // D = L diag(p) L^T, unit lower L (as vmp_sweep.h sweep_block)
    const double p0 = d00;
    const double r0 = fast_recip3(p0);
    const double l10 = d01 * r0, l20 = d02 * r0, l30 = d03 * r0;
    const double p1 = __builtin_fma(-l10, d01, d11);
    const double r1 = fast_recip3(p1);
    const double u21 = __builtin_fma(-l20, d01, d12);
    const double u31 = __builtin_fma(-l30, d01, d13);
    const double l21 = u21 * r1, l31 = u31 * r1;
    const double p2 = __builtin_fma(-l21, u21, __builtin_fma(-l20, d02, d22));
    const double r2 = fast_recip3(p2);
    const double u32 = __builtin_fma(-l31, u21, __builtin_fma(-l30, d02, d23));
    const double l32 = u32 * r2;
    const double p3 = __builtin_fma(-l32, u32, __builtin_fma(-l31, u31,
                                                             __builtin_fma(-l30, d03, d33)));
    const double r3 = fast_recip3(p3);
    if (!(p0 > 0.0 && p1 > 0.0 && p2 > 0.0 && p3 > 0.0)) bad = 1;
    const double q0 = prod * (p0 * p1);
    ld += (double)__builtin_amdgcn_frexp_exp(q0);
    const double q1 = __builtin_amdgcn_frexp_mant(q0) * (p2 * p3);
    ld += (double)__builtin_amdgcn_frexp_exp(q1);
    prod = __builtin_amdgcn_frexp_mant(q1);
    // column c = lj of D^-1: L y = e_c, z = y / p, L^T x = z; this lane keeps entry li
    const double e0 = (lj == 0) ? 1.0 : 0.0, e1 = (lj == 1) ? 1.0 : 0.0;
    const double e2 = (lj == 2) ? 1.0 : 0.0, e3 = (lj == 3) ? 1.0 : 0.0;
    const double y1 = __builtin_fma(-l10, e0, e1);
    const double y2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, e0, e2));
    const double y3 = __builtin_fma(-l32, y2, __builtin_fma(-l31, y1, __builtin_fma(-l30, e0, e3)));
    const double x3 = y3 * r3;
    const double x2 = __builtin_fma(-l32, x3, y2 * r2);
    const double x1 = __builtin_fma(-l31, x3, __builtin_fma(-l21, x2, y1 * r1));
    const double x0 = __builtin_fma(-l30, x3, __builtin_fma(-l20, x2, __builtin_fma(-l10, x1, e0 * r0)));
    const double E = (li == 0) ? x0 : (li == 1) ? x1 : (li == 2) ? x2 : x3;
    const double nE = -E;
    // ---- panels and the rank-4 update of every other block ------------------------------------
    double Ct[NB], nQ[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        if (I < p) Ct[I] = S[bidx(p, I)];
        else if (I > p) Ct[I] = mfma4(S[bidx(I, p)], ident, 0.0);
    }
#pragma unroll
    for (int I = 0; I < NB; ++I)
        if (I != p) nQ[I] = mfma4(nE, Ct[I], 0.0);
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J)
            if (I != p && J != p) S[bidx(I, J)] = mfma4(nQ[I], Ct[J], S[bidx(I, J)]);
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        if (I < p) S[bidx(p, I)] = -nQ[I];
        else if (I > p) S[bidx(I, p)] = mfma4(Ct[I], E, 0.0);
    }
    S[bidx(p, p)] = nE;
}

// Packed index of element (4 I + li, 4 J + lj) of the symmetric matrix from four per-lane
// constants (a table of the 36 indices per lane costs 36 registers of a kernel that needs them for
// the matrix): off-diagonal blocks have row > column, diagonal blocks take (max, min).
struct blk4_lane {
    int li4, t0;      // I > J:  pk = 8 I^2 + 2 I + 4 J + I * li4 + t0,   t0 = li (li + 1) / 2 + lj
    int m4, t0d;      // I == J: pk = 8 I^2 + 6 I     + I * m4  + t0d,  m = max(li, lj), n = min
};

template <int I, int J>
__device__ __forceinline__ int blk4_pk(const blk4_lane &c)
{
    if constexpr (I == J) return 8 * I * I + 6 * I + I * c.m4 + c.t0d;
    else return 8 * I * I + 2 * I + 4 * J + I * c.li4 + c.t0;
}

template <int NB, int p, typename F>
__device__ __forceinline__ void blk4_sweep(double (&S)[NB * (NB + 1) / 2], int li, int lj,
                                           const double (&sel)[4], double ident, double &prod,
                                           double &ld, int &bad, F &&after_step)
{
    if constexpr (p < NB) {
        blk4_step<NB, p>(S, li, lj, sel, ident, prod, ld, bad);
        after_step(std::integral_constant<int, p>{});
        blk4_sweep<NB, p + 1>(S, li, lj, sel, ident, prod, ld, bad, after_step);
    }
}

template <int NB, int I, int J, typename F>
__device__ __forceinline__ void blk4_for_blocks(F &&f)
{
    if constexpr (I < NB) {
        f(std::integral_constant<int, I>{}, std::integral_constant<int, J>{});
        if constexpr (J < I) blk4_for_blocks<NB, I, J + 1>(f);
        else blk4_for_blocks<NB, I + 1, 0>(f);
    }
}

// DBF > 0: the FUSED form (round 5) -- the precision GEMM runs inside this kernel and Lam~ never
// exists in HBM.  A workgroup owns the 16 plates of a subtile; per subtile:
//   phase A  mpca_lambda_group<DBF, KT, 1>: the four wavefronts split the column tiles exactly as
//            mpca_lambda_kernel does (B fragments straight from L2 into registers, no staging, no
//            barrier inside the k-loop), each accumulating its tiles for all 16 plates;
//   exchange the accumulators go to the staging area as the four packed rows of each plate group --
//            what the LDS-DMA of the separate form delivers from HBM -- between two barriers;
//   phase B  wavefront w runs the per-plate stage below, unchanged, on plate group w.
// Two workgroups share a CU; the odd ones start late (half a subtile), so that a SIMD mostly holds
// one wavefront in phase A (matrix pipe) beside one in phase B (mostly vector ALU).
// Saves the 4.4 GB written + 4.4 GB read per 2^20 plates of the Lam~ round trip; costs the B
// fragments twice as often from L2 (a workgroup of the separate GEMM owns 32 plates).
template <int NB, bool FULLK, int DBF>
__global__ void __launch_bounds__(NT, 2)
mpca_blk4_kernel(const double *__restrict__ Lam, int64_t n0, int64_t nplates_chunk, int K,
                 double x_prec, const double *__restrict__ tau_ptr, double *__restrict__ XXf,
                 double *__restrict__ Xm, int write_x, double *__restrict__ partial,
                 double *__restrict__ partial_sxx, const double *__restrict__ Ymt,
                 const uint32_t *__restrict__ Mb1, const double *__restrict__ panel, int stagger)
{
    constexpr int KT = NB <= 4 ? 1 : 2, KP = 16 * KT, P = KP * (KP + 1) / 2, PT = (P + 15) / 16;
    constexpr int LRC = 16 * (PT + KT), NBB = NB * (NB + 1) / 2, PT2 = (PT + 1) / 2;
    constexpr int NPAIR = 4 * LRC / 2, NV = (NPAIR + 63) / 64;
    static_assert(PT2 * 128 <= 4 * LRC, "a block row of XXf fits in the staging area");
    // staging of one wavefront: four packed rows (matrix | right-hand side) as they lie in Lam;
    // later in the iteration the block row of XXf that leaves
    __shared__ __attribute__((aligned(16))) double stg[4][4 * LRC];
    __shared__ double sa[P];                     // packed sum of <x x^T>_n of this workgroup
    __shared__ double red[NT / 64];
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = l >> 4, lb = (l >> 2) & 3, lj = l & 3;
    const double tau = tau_ptr[0];
    for (int e = threadIdx.x; e < P; e += NT) sa[e] = 0.0;
    // operand constants: row selectors, the identity (both in the result layout)
    double sel[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) sel[a] = (li == a) ? 1.0 : 0.0;
    const double ident = (li == lj) ? 1.0 : 0.0;
    const double dgv = (li == lj) ? x_prec : 0.0;
    blk4_lane lc;
    {
        const int m = li > lj ? li : lj, n = li > lj ? lj : li;
        lc.li4 = 4 * li;
        lc.t0 = li * (li + 1) / 2 + lj;
        lc.m4 = 4 * m;
        lc.t0d = m * (m + 1) / 2 + n;
    }
    // FULLK: K == 4 NB, no padded rows / columns in the last block row
    const bool padr = !FULLK && 4 * (NB - 1) + li >= K, padc = !FULLK && 4 * (NB - 1) + lj >= K;
    double pm = 1.0, le = 0.0;                   // product of the pivots of this lane's plates
    double trl = 0.0;                            // this lane's share of sum_n tr<xx>_n
    int anybad = 0;
    // The packed rows of plates 4 q .. 4 q + 3, contiguous in Lam, travel HBM -> staging area as
    // LDS-DMA (global_load_lds_dwordx4: 1 KB per instruction, no registers; rows beyond the chunk:
    // zeros).  Through registers -- requested before the stores of an iteration, parked after
    // them -- the compiler spilled two thirds of the 36 registers to scratch in the loop, and that
    // traffic (3 GB written, 3.6 GB read per 2^20 plates, profiles/r03/pmc_blk4_spill.txt) cost
    // more than the latency the prefetch hid.
    auto fetch = [&](int64_t q) {
        const int64_t left = nplates_chunk - 4 * q;
        const int npair = left >= 4 ? NPAIR : (int)(left * (LRC / 2));
        const v2f64 *src = reinterpret_cast<const v2f64 *>(Lam + 4 * q * LRC);
        if (left < 4) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int pidx = l + 64 * i;
                if (pidx >= npair && pidx < NPAIR)
                    reinterpret_cast<v2f64 *>(stg[w])[pidx] = v2f64{0.0, 0.0};
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int pidx = l + 64 * i;
            if (pidx < npair)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + pidx),
                    (__attribute__((address_space(3))) void *)(stg[w] + 128 * i), 16, 0, 2);
        }
    };
    const int64_t ngroups = (nplates_chunk + 3) / 4;
    const blk4_lane lc0 = lc;
    // the per-plate stage of the four plates 4 q .. 4 q + 3 whose packed rows lie in stg[w]
    auto plate_stage = [&](int64_t q) {
        // the per-lane index constants are re-materialised every iteration: left loop-invariant,
        // the compiler hoists the 36 gather and the 36 store offsets of a lane out of the loop and
        // spills the matrix instead
        lc = lc0;
        asm volatile("" : "+v"(lc.li4), "+v"(lc.t0), "+v"(lc.m4), "+v"(lc.t0d));
        const double *row = stg[w] + lb * LRC;
        const bool valid = 4 * q + lb < nplates_chunk;
        // ---- gather: S = c I + tau Lam~ (identity on the padding) ------------------------------
        double S[NBB];
        blk4_for_blocks<NB, 0, 0>([&](auto Ic, auto Jc) {
            constexpr int I = decltype(Ic)::value, J = decltype(Jc)::value;
            const double v = row[blk4_pk<I, J>(lc)];
            double a = (I == J) ? __builtin_fma(tau, v, dgv) : tau * v;
            if (!FULLK && I == NB - 1) {
                if (J == NB - 1) a = (padr || padc) ? ident : a;
                else a = padr ? 0.0 : a;
            }
            S[bidx(I, J)] = a;
        });
        lds_fence();
        // (the rows stay in the staging area until this iteration's <x x^T> is assembled there)
        double prod = 1.0, ld = 0.0;
        int bad = 0;
        blk4_sweep<NB, 0>(S, li, lj, sel, ident, prod, ld, bad, [&](auto) {});
        // ---- <x> = Cov (tau rhs), Cov = -S ---------------------------------------------------------
        double hq[NB], hr[NB];                    // tau rhs at the lane's column / row index
#pragma unroll
        for (int J = 0; J < NB; ++J) {
            const double a = row[16 * PT + 4 * J + lj];
            const double b = row[16 * PT + 4 * J + li];
            hq[J] = (FULLK || 4 * J + lj < K) ? tau * a : 0.0;
            hr[J] = (FULLK || 4 * J + li < K) ? tau * b : 0.0;
        }
        double xrow[NB];
#pragma unroll
        for (int J = 0; J < NB; ++J) {
            // contributions that contract the ROW index of a stored block: sum_{I >= J} S_IJ^T h_I
            double t = 0.0;
#pragma unroll
            for (int I = J; I < NB; ++I) t = mfma4(S[bidx(I, J)], hr[I], t);
            xrow[J] = t;
        }
#pragma unroll
        for (int I = 1; I < NB; ++I) {
            // ... and the COLUMN index: sum_{J < I} S_IJ h_J, summed over the quad
            double t = 0.0;
#pragma unroll
            for (int J = 0; J < I; ++J) t = __builtin_fma(S[bidx(I, J)], hq[J], t);
            t += dpp_f64<0xB1>(t);
            t += dpp_f64<0x4E>(t);
            xrow[I] += t;
        }
        double xcol[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            xrow[I] = -xrow[I];
            xcol[I] = mfma4(xrow[I], ident, 0.0);
        }
        // ---- <x x^T> = Cov + <x><x>^T; stores ----------------------------------------------------
        {
            // This wavefront's plates 4 q .. 4 q + 3 own the whole block row q of XXf (PT2 blocks
            // of 1 KB).  The row is assembled in the staging area and leaves as whole 16-byte
            // lanes, 1 KB per instruction: written from the registers as 8-byte scatters (two
            // stores of two different blocks complete a 16-byte slot) the lines reached the HBM
            // controllers partially filled on most boxes of the pool -- 6.1 GB written and 5.8 GB
            // read per 2^20 plates for 4.7 + 4.9 (profiles/r03/pmc_blk4_scatter.txt), 3.3 ms
            // against 2.3 on the boxes where the L2 happened to merge them.
            double *ob = stg[w];
            blk4_for_blocks<NB, 0, 0>([&](auto Ic, auto Jc) {
                constexpr int I = decltype(Ic)::value, J = decltype(Jc)::value;
                double v = __builtin_fma(xrow[I], xcol[J], -S[bidx(I, J)]);
                if (!FULLK && I == NB - 1) v = (padr || (J == NB - 1 && padc)) ? 0.0 : v;
                if (I > J || li >= lj) {
                    const int pk = blk4_pk<I, J>(lc);
                    ob[lb * 32 + xxf_off(pk)] = v;
                    if (valid)
                        __hip_atomic_fetch_add(&sa[pk], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // tr<xx> (a bound term) per lane, combined in fixed order below: the shared
                // accumulator above (the rotation statistic sum_n <xx>_n) takes its additions in
                // the order the four wavefronts happen to arrive
                if (I == J && li == lj && valid) trl += v;
            });
            // packed entries beyond the data: block rows K/4 .. KP/4 - 1 and the padding of the
            // last tile pair (mpca_stats reads whole tiles)
            if (4 * NB < KP) {
#pragma unroll
                for (int I = NB; I < KP / 4; ++I)
#pragma unroll
                    for (int J = 0; J <= I; ++J) {
                        const int r = 4 * I + li, c = 4 * J + lj;
                        const int pk = tri(r > c ? r : c, r > c ? c : r);
                        if (I > J || li >= lj) ob[lb * 32 + xxf_off(pk)] = 0.0;
                    }
            }
            for (int pk = P + (l & 15); pk < 32 * PT2; pk += 16) ob[(l >> 4) * 32 + xxf_off(pk)] = 0.0;
            lds_fence();
            v2f64 *xo = reinterpret_cast<v2f64 *>(XXf + q * ((int64_t)PT2 * 128)) + l;
            if (q < ngroups) {
#pragma unroll
                for (int i0 = 0; i0 < PT2; i0 += 4) {
                    v2f64 t[4];
#pragma unroll
                    for (int i = i0; i < PT2 && i < i0 + 4; ++i)
                        t[i - i0] = reinterpret_cast<const v2f64 *>(ob)[i * 64 + l];
#pragma unroll
                    for (int i = i0; i < PT2 && i < i0 + 4; ++i) xo[i * 64] = t[i - i0];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (valid && write_x && lj == 0) {
                double *xr = Xm + (n0 + 4 * q + lb) * KP;
#pragma unroll
                for (int I = 0; I < NB; ++I)
                    xr[4 * I + li] = (FULLK || 4 * I + li < K) ? xrow[I] : 0.0;
                if (4 * NB < KP)
                    for (int r = 4 * NB + li; r < KP; r += 4) xr[r] = 0.0;
            }
            if (valid) {
                const double qq = pm * prod;
                le += ld + (double)__builtin_amdgcn_frexp_exp(qq);
                pm = __builtin_amdgcn_frexp_mant(qq);
                anybad |= bad;
            }
        }
        lds_fence();
    };
    if constexpr (DBF == 0) {
        const int64_t gstep = (int64_t)gridDim.x * 4;
        int64_t q = (int64_t)blockIdx.x * 4 + w;
        if (q < ngroups) fetch(q);
        __syncthreads();                              // sa zeroed
        for (; q < ngroups; q += gstep) {
            // the rows requested at the end of the previous iteration have landed in the staging
            // area (LDS-DMA completes on vmcnt; the compiler does not order the reads below behind it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_fence();
            plate_stage(q);
            if (q + gstep < ngroups) fetch(q + gstep);
        }
    } else {
        const int64_t sub0 = n0 / 16, nsub = (nplates_chunk + 15) / 16;
        // the odd workgroups of the grid (the second one of a CU) start half a subtile late
        if (stagger > 0 && (blockIdx.x & 1)) {
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
        }
        for (int64_t sb = blockIdx.x; sb < nsub; sb += gridDim.x) {
            __syncthreads();                          // the staging area is free (and sa zeroed)
            mpca_lambda_group<DBF, KT, 1>(Ymt, Mb1, panel, sub0, nsub, sb,
                                          [&](int c, int, int R, bool, double v) {
                // plate 4 R + (l >> 4) of the subtile in either result layout: group R, row l >> 4
                stg[R][(l >> 4) * LRC + 16 * c + (l & 15)] = v;
            });
            __syncthreads();
            plate_stage(4 * sb + w);
        }
    }
    // per-workgroup partials: tr<xx> = trace of the accumulated sum, log|Cov|, status
    __syncthreads();
    const double tr = block_sum<NT>(trl, red);
    const double ldw = block_sum<NT>((li == 0 && lj == 0)
                                         ? -(log(pm) + le * 0.69314718055994530942) : 0.0, red);
    const double bd = block_sum<NT>((double)anybad, red);
    if (threadIdx.x == 0) {
        partial[3 * blockIdx.x + 0] = tr;
        partial[3 * blockIdx.x + 1] = ldw;
        partial[3 * blockIdx.x + 2] = bd;
    }
    for (int e = threadIdx.x; e < KP * KP; e += NT) {
        const int i = e / KP, j = e - i * KP;
        const int a = i > j ? i : j, b = i > j ? j : i;
        partial_sxx[(int64_t)blockIdx.x * KP * KP + e] = (i < K && j < K) ? sa[tri(a, b)] : 0.0;
    }
}

// -------------------------------------------------------------------------------------------
// mpca_stats: M_d = sum_n m_dn <x x^T>_n (the packed columns), wavefronts split the COLUMN tiles.
// Wavefront w owns ONE PAIR of column tiles of its workgroup's slice and ALL row tiles dt: its B
// operands (16 bytes per lane = the two tiles of the pair for one k-step of four plates, XXf
// order) are loaded by no other wavefront, the A operand is the mask bit of (d, n) -- no loads at
// all.  r_d comes from mpca_ryx_kernel.
// (Round 3: the same GEMM on v_mfma_f64_4x4x4_4b_f64 -- the form mpca_lambda and the per-plate
// stage now use -- was built and measured here: 4.1 ms per 2^20 plates against 3.5.  A wavefront of
// this kernel owns two column tiles, so every mask operand (one v_bfe + v_cvt) feeds only two of
// the short instructions where it feeds two long ones here, and at two wavefronts per SIMD the
// vector instructions do not hide under the matrix ones: tools/mfma4_lab.hip, "bfe + cvt + 2
// mfma4x4x4, VGPR acc" = 10.1 ns per instruction against 7.2 bare.  mpca_lambda amortises each mask
// operand over nine column tiles.)
// -------------------------------------------------------------------------------------------
template <int DB, int KT>
__global__ void __launch_bounds__(NT, 2)
mpca_stats2_kernel(const uint32_t *__restrict__ Mb2, const double *__restrict__ XXf, int64_t sub0,
                   int64_t nsub_chunk, int nslices, double *__restrict__ partial)
{
    constexpr int DP = 32 * DB, DT = DP / 16;
    constexpr int KP = 16 * KT, P = KP * (KP + 1) / 2, PT = (P + 15) / 16, CT = PT + KT;
    constexpr int PT2 = (PT + 1) / 2, LR = 16 * CT;
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = l & 15, l4 = l >> 4;
    const int slice = blockIdx.x % nslices, wg = blockIdx.x / nslices, nwg = gridDim.x / nslices;
    const int cp = slice * 4 + w;                   // this wavefront's pair of column tiles
    const int c0 = 2 * cp;
    v4f64 acc[DT][2];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[i][t] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int64_t npair = (nsub_chunk + 1) / 2;     // 32 plates: 8 k-steps = 8 pair-loads
    v2f64 bcur[8], bnxt[8];
    auto issue = [&](int64_t pr, v2f64 (&b)[8]) {
        // plates 32 pr .. 32 pr + 31 of the chunk: plate groups n/4 = 8 pr + ks
        const double *base = XXf + (((pr * 8) * (int64_t)PT2 + cp) * 64 + l) * 2;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            b[ks] = (cp < PT2) ? *reinterpret_cast<const v2f64 *>(base + ((int64_t)ks * PT2 * 64) * 2)
                               : v2f64{0.0, 0.0};
    };
    int64_t pr = wg;
    if (pr < npair) issue(pr, bcur);
    for (; pr < npair; pr += nwg) {
        const int64_t s0 = sub0 + 2 * pr;
        const uint32_t mw0 = Mb2[s0 * 64 + l];
        const uint32_t mw1 = (2 * pr + 1 < nsub_chunk) ? Mb2[(s0 + 1) * 64 + l] : 0u;
        if (pr + nwg < npair) issue(pr + nwg, bnxt);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int qq = ks & 3;                              // k-step inside its subtile
            const uint32_t mw = (ks < 4) ? mw0 : mw1;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const double am = (double)((mw >> (4 * i + qq)) & 1u);
                acc[i][0] = mfma(am, bcur[ks].x, acc[i][0]);
                acc[i][1] = mfma(am, bcur[ks].y, acc[i][1]);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bcur[ks] = bnxt[ks];
    }
    double *pb = partial + (int64_t)wg * DP * LR;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c = c0 + t;
            if (c < PT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pb[(int64_t)(16 * i + l4 + 4 * r) * LR + 16 * c + l15] = acc[i][t][r];
            }
        }
}

// -------------------------------------------------------------------------------------------
// mpca_stats3: the same GEMM on v_mfma_f64_4x4x4_4b_f64 (round 3), shaped so that the mask
// operand pays: a wavefront owns 2 row tiles (32 rows d) and NP PAIRS of column tiles, so every
// mask operand (one v_bfe + v_cvt per 4 rows x 4 plates) feeds 2 NP of the short instructions
// (two in the column-split form above, where that form lost to the 16x16x4 instruction).  The four
// wavefronts of a workgroup take the four quarters of the rows and read the SAME B fragments
// (XXf pair order: 16 bytes per lane = one k-step of a tile pair; the second to fourth reader hit
// the caches).  The blocks of the instruction are the 4-column groups of a tile -- B operand as
// stored --, the A operand is the mask of rows 16 dt + 4 R + i replicated over the blocks = the
// mask word of lane 4 R + i + 16 k, fetched once per 16 plates (ds_bpermute).  B fragments: a
// window of four k-steps in registers, each slot refilled (four k-steps ahead) right after its
// last use; mask words one 32-plate step ahead; all loads unconditional from clamped indices.
// -------------------------------------------------------------------------------------------
template <int DB, int KT, int NP>
__global__ void __launch_bounds__(NT, 2)
mpca_stats3_kernel(const uint32_t *__restrict__ Mb2, const double *__restrict__ XXf, int64_t sub0,
                   int64_t nsub_chunk, int nslices, double *__restrict__ partial)
{
    constexpr int DP = 32 * DB, DT = DP / 16, DTW = (DT + 3) / 4;
    constexpr int KP = 16 * KT, P = KP * (KP + 1) / 2, PT = (P + 15) / 16, CT = PT + KT;
    constexpr int PT2 = (PT + 1) / 2, LR = 16 * CT;
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = blockIdx.x % nslices, wg = blockIdx.x / nslices, nwg = gridDim.x / nslices;
    const int cp0 = slice * NP;                     // first tile pair of this workgroup
    const int dt0 = w * DTW;                        // first row tile of this wavefront
    if (dt0 >= DT) return;
    double acc[DTW][4][NP][2];
#pragma unroll
    for (int i = 0; i < DTW; ++i)
#pragma unroll
        for (int R = 0; R < 4; ++R)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) acc[i][R][pp][0] = acc[i][R][pp][1] = 0.0;
    const int64_t npair = (nsub_chunk + 1) / 2;     // 32 plates: 8 k-steps
    int cpl[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) cpl[pp] = cp0 + pp < PT2 ? cp0 + pp : PT2 - 1;
    auto frag = [&](int64_t pr, int ks, int pp) {
        // plates 32 pr .. 32 pr + 31 of the chunk: plate groups n/4 = 8 pr + ks
        return *reinterpret_cast<const v2f64 *>(
            XXf + (((pr * 8 + ks) * (int64_t)PT2 + cpl[pp]) * 64 + l) * 2);
    };
    auto masks = [&](int64_t pr, uint32_t (&m)[2]) {
        const int64_t s0 = sub0 + 2 * pr;
        const int64_t s1 = (2 * pr + 1 < nsub_chunk) ? s0 + 1 : s0;
        m[0] = Mb2[s0 * 64 + l];
        m[1] = Mb2[s1 * 64 + l];
    };
    int src[4];
#pragma unroll
    for (int R = 0; R < 4; ++R) src[R] = 4 * ((l & 0x30) | (4 * R) | (l & 3));
    v2f64 bf[4][NP];
    uint32_t mcur[2], mnxt[2];
    int64_t pr = wg;
    if (pr < npair) {
        masks(pr, mcur);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) bf[ks][pp] = frag(pr, ks, pp);
    }
    for (; pr < npair; pr += nwg) {
        const int64_t prn = pr + nwg < npair ? pr + nwg : pr;
        masks(prn, mnxt);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * pr + 1 >= nsub_chunk) mcur[1] = 0u;     // an odd last subtile has no second half
        uint32_t mr[2][4];
#pragma unroll
        for (int R = 0; R < 4; ++R) {
            mr[0][R] = (uint32_t)__builtin_amdgcn_ds_bpermute(src[R], (int)mcur[0]) >> (4 * dt0);
            mr[1][R] = (uint32_t)__builtin_amdgcn_ds_bpermute(src[R], (int)mcur[1]) >> (4 * dt0);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int qq = ks & 3;                              // k-step inside its subtile
#pragma unroll
            for (int i = 0; i < DTW; ++i)
#pragma unroll
                for (int R = 0; R < 4; ++R) {
                    const double am = (double)((mr[ks >> 2][R] >> (4 * i + qq)) & 1u);
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) {
                        acc[i][R][pp][0] = mfma4(am, bf[ks & 3][pp].x, acc[i][R][pp][0]);
                        acc[i][R][pp][1] = mfma4(am, bf[ks & 3][pp].y, acc[i][R][pp][1]);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
                bf[ks & 3][pp] = ks < 4 ? frag(pr, ks + 4, pp) : frag(prn, ks - 4, pp);
            __builtin_amdgcn_sched_barrier(0);
        }
        mcur[0] = mnxt[0];
        mcur[1] = mnxt[1];
    }
    // result layout of the instruction: row 4 R + (l >> 4), column l & 15 of the tile
    double *pb = partial + (int64_t)wg * DP * LR;
#pragma unroll
    for (int i = 0; i < DTW; ++i)
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int c = 2 * (cp0 + pp) + t;
                if (cp0 + pp < PT2 && c < PT && dt0 + i < DT) {
#pragma unroll
                    for (int R = 0; R < 4; ++R)
                        pb[(int64_t)(16 * (dt0 + i) + 4 * R + (l >> 4)) * LR + 16 * c + (l & 15)] =
                            acc[i][R][pp][t];
                }
            }
}

// r_d = sum_n (m y)_dn <x_n>: the message product of the fully observed block on the masked data
// (zero where missing).  One 32-plate tile of Ymt (contiguous) and of Xm per trip through LDS,
// S += Ymt_tile . Xm_tile on the matrix cores (contraction over the plates).
template <int DB, int KT>
__global__ void __launch_bounds__(NT, 2)
mpca_ryx_kernel(const double *__restrict__ Ymt, const double *__restrict__ Xm, int64_t tile0,
                int64_t ntiles_chunk, double *__restrict__ partial, int LR, int col0)
{
    constexpr int DP = 32 * DB, DT = DP / 16, KP = 16 * KT;
    constexpr int SY = TN + 2;                        // LDS row strides (doubles)
    __shared__ double Ys[DP * SY];
    __shared__ double Xs[KP * SY];
    const int tid = threadIdx.x;
    const int l = tid & 63, w = tid >> 6, l15 = l & 15, l4 = l >> 4;
    constexpr int T2 = DT * KT, R2 = (T2 + 3) / 4;
    v4f64 acc[R2];
#pragma unroll
    for (int m = 0; m < R2; ++m) acc[m] = v4f64{0.0, 0.0, 0.0, 0.0};
    // The next tile travels HBM -> registers while the current one is multiplied out of LDS (the
    // first form loaded, synchronised, multiplied, synchronised: 1.5 ms per 2^20 plates against
    // 0.3 ms of traffic).
    constexpr int NY = DP * TN / 2 / NT;              // 16-byte pieces of the Ymt tile per thread
    constexpr int NX = TN * KP / NT;                  // doubles of the Xm tile per thread
    static_assert(DP * TN / 2 % NT == 0 && TN * KP % NT == 0, "tile sizes are multiples of the block");
    v2f64 ry[NY];
    double rx[NX];
    auto fetch = [&](int64_t tile) {
        const v2f64 *yt = reinterpret_cast<const v2f64 *>(Ymt + (tile0 + tile) * ((int64_t)DP * TN));
        const double *xt = Xm + (tile0 + tile) * ((int64_t)TN * KP);
#pragma unroll
        for (int i = 0; i < NY; ++i) ry[i] = __builtin_nontemporal_load(yt + tid + i * NT);
#pragma unroll
        for (int i = 0; i < NX; ++i) rx[i] = xt[tid + i * NT];
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int e = tid + i * NT;
            const int d = e / (TN / 2), j2 = (e - d * (TN / 2)) * 2;
            *reinterpret_cast<v2f64 *>(&Ys[d * SY + j2]) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + i * NT;
            const int n = e / KP, k = e - n * KP;
            Xs[k * SY + n] = rx[i];
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles_chunk) fetch(tile);
    for (; tile < ntiles_chunk; tile += gridDim.x) {
        __syncthreads();                              // the previous tile has been consumed
        park();
        __syncthreads();
        if (tile + gridDim.x < ntiles_chunk) fetch(tile + gridDim.x);
#pragma unroll
        for (int q = 0; q < TN / 4; ++q) {
#pragma unroll
            for (int m = 0; m < R2; ++m) {
                const int t2 = w + 4 * m;
                if (t2 < T2) {
                    const int dt = t2 / KT, kt = t2 - dt * KT;
                    // A[i = d][k = n]: lane (l15, l4);  B[k = n][j = kcol]: lane (l4, l15)
                    const double a = Ys[(16 * dt + l15) * SY + 4 * q + l4];
                    const double b = Xs[(16 * kt + l15) * SY + 4 * q + l4];
                    acc[m] = mfma(a, b, acc[m]);
                }
            }
        }
    }
    double *pb = partial + (int64_t)blockIdx.x * DP * LR;
#pragma unroll
    for (int m = 0; m < R2; ++m) {
        const int t2 = w + 4 * m;
        if (t2 < T2) {
            const int dt = t2 / KT, kt = t2 - dt * KT;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pb[(int64_t)(16 * dt + l4 + 4 * r) * LR + col0 + 16 * kt + l15] = acc[m][r];
        }
    }
}

// -------------------------------------------------------------------------------------------
// W.update(): one wavefront per row d.  Lam_d = diag<alpha> + <tau> M_d  (gaussian.py:649-706
// with the masked messages of dot.py:425-633), w_d = Cov_d <tau> r_d, <ww>_d = Cov_d + w w^T.
// Writes the plain moments (W, WW, log|Cov_d|) and the B-operand panel of mpca_lambda.
// mode 0: VB update;  mode 1: delta moments of the <w_d> already in st[off_W] (initialize_from_value)
// mode 2: prior moments: mean 0, covariance diag(1/<alpha>)
// -------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(NT, 2)
mpca_w_kernel(vmp_mpca_layout L, int D, int K, int DQ, int mode, double *__restrict__ st)
{
    constexpr int KP = 16 * KT, P = KP * (KP + 1) / 2, PT = (P + 15) / 16;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = l & 15, l4 = l >> 4;
    const int d = blockIdx.x * 4 + w;
    if (d >= D) return;
    const double tau = st[L.off_tau + 2];
    const double *alpha = st + L.off_alpha + 2 * KP;
    const double *mrow = st + L.off_M + (int64_t)d * L.LR;
    double *wrow = st + L.off_W + (int64_t)d * KP;
    v4f64 T[2][2];
    plate_result res;
    if (mode == 0) {
        plate_raw q;
        load_raw<KT>(q, mrow, K, l15, l4);
        tiles_from_raw(T, q, K, 0.0, tau, l15, l4);
        // + diag<alpha>
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tr + l4 + 4 * r, j = 16 * tr + l15;
                if (i == j && i < K) T[tr][tr][r] += alpha[i];
            }
        double prod = 1.0, ld = 0.0;
        int bad = 0;
        sweep_upto<0>(T, (K + 3) / 4, l15, l4, prod, ld, bad);
        res = finish_plate(T, mrow + 16 * PT, tau, K, l15, l4, prod, ld, bad);
    } else {
        const double xa = (mode == 1 && l4 == 0 && l15 < K) ? wrow[l15] : 0.0;
        const double xb = (mode == 1 && KT > 1 && l4 == 0 && 16 + l15 < K) ? wrow[16 + l15] : 0.0;
        const v4f64 z = {0.0, 0.0, 0.0, 0.0};
        T[0][0] = mfma(xa, xa, z);
        T[0][1] = mfma(xa, xb, z);
        T[1][0] = mfma(xb, xa, z);
        T[1][1] = mfma(xb, xb, z);
        double ldp = 0.0;
        if (mode == 2) {
#pragma unroll
            for (int tr = 0; tr < 2; ++tr)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * tr + l4 + 4 * r, j = 16 * tr + l15;
                    if (i == j && i < K) T[tr][tr][r] += 1.0 / alpha[i];
                }
            for (int k = 0; k < K; ++k) ldp += log(alpha[k]);     // log|Lam| of the prior
        }
        res.x0 = xa;
        res.x1 = xb;
        res.logdet = ldp;
        res.bad = 0;
    }
    // plain moments
    double *ww = st + L.off_WW + (int64_t)d * KP * KP;
    double *panel = st + L.off_panel;
#pragma unroll
    for (int tr = 0; tr < KT; ++tr)
#pragma unroll
        for (int tc = 0; tc < KT; ++tc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tr + l4 + 4 * r, j = 16 * tc + l15;
                const double v = (i < K && j < K) ? T[tr][tc][r] : 0.0;
                ww[i * KP + j] = v;
                if (i >= j) {
                    const int p = tri(i, j);
                    panel[panel_index(DQ, p >> 4, d, p & 15)] = v;
                }
            }
    if (l4 == 0) {
        const double v0 = (l15 < K) ? res.x0 : 0.0;
        wrow[l15] = v0;
        panel[panel_index(DQ, PT, d, l15)] = v0;
        if (KT > 1) {
            const double v1 = (16 + l15 < K) ? res.x1 : 0.0;
            wrow[16 + l15] = v1;
            panel[panel_index(DQ, PT + 1, d, l15)] = v1;
        }
    }
    if (l == 0) {
        st[L.off_ldW + d] = -res.logdet;            // log|Cov_d|
        if (res.bad) st[L.off_scal + SC_STATUS] = (double)VMP_ERR_NOT_POSDEF;
    }
}

// -------------------------------------------------------------------------------------------
// tau / alpha / lower bound: one workgroup (gamma.py:116-148, expfamily.py:400-480)
// -------------------------------------------------------------------------------------------
struct small_args {
    vmp_mpca_layout L;
    int D, K, nops;
    int ops[8];
    double x_prec, a0t, b0t, a0a, b0a;
};

__device__ inline double gamma_term(double a0, double b0, double a, double b, double x, double lx)
{
    // E[log p(x | a0, b0) - log q(x)],  q = Gamma(a, b)
    return (a0 * log(b0) - vmp_lgamma(a0)) - (a * log(b) - vmp_lgamma(a)) + (b - b0) * x
           + (a0 - a) * lx;
}

__global__ void __launch_bounds__(NT)
mpca_small_kernel(small_args A, double *__restrict__ st)
{
    __shared__ double red[NT / 64];
    const vmp_mpca_layout &L = A.L;
    const int D = A.D, K = A.K, KP = (int)L.KP, PP = 16 * (int)L.PT, LR = (int)L.LR;
    const int tid = threadIdx.x;
    double *sc = st + L.off_scal;
    for (int oi = 0; oi < A.nops; ++oi) {
        const int op = A.ops[oi];
        __syncthreads();
        if (op == VMP_MPCA_OP_TAU || op == VMP_MPCA_OP_ELBO) {
            // residual = sum m y^2 - 2 sum_d <w_d>.r_d + sum_d <ww>_d : M_d
            double s = 0.0;
            for (int e = tid; e < D * K * K; e += NT) {
                const int d = e / (K * K), ij = e - d * K * K, i = ij / K, j = ij - i * K;
                const int a = i > j ? i : j, b = i > j ? j : i;
                s += st[L.off_WW + ((int64_t)d * KP + i) * KP + j] * st[L.off_M + (int64_t)d * LR + tri(a, b)];
            }
            for (int e = tid; e < D * K; e += NT) {
                const int d = e / K, k = e - d * K;
                s -= 2.0 * st[L.off_W + (int64_t)d * KP + k] * st[L.off_M + (int64_t)d * LR + PP + k];
            }
            s = block_sum<NT>(s, red);
            if (tid == 0) sc[SC_RESID] = sc[SC_SYY] + s;
            __syncthreads();
        }
        if (op == VMP_MPCA_OP_TAU) {
            if (tid == 0) {
                const double a = A.a0t + 0.5 * sc[SC_NOBS], b = A.b0t + 0.5 * sc[SC_RESID];
                st[L.off_tau + 0] = a;
                st[L.off_tau + 1] = b;
                st[L.off_tau + 2] = a / b;
                st[L.off_tau + 3] = vmp_digamma(a) - log(b);
                if (!(b > 0.0)) sc[SC_STATUS] = (double)VMP_ERR_FLOATING;
            }
        } else if (op == VMP_MPCA_OP_ALPHA) {
            // rows of W without any observation are ignored plates of W (mask propagation,
            // node.py:457-526): no message to alpha (node.py:624-650), no bound term
            for (int k = tid; k < K; k += NT) {
                double s = 0.0;
                int de = 0;
                for (int d = 0; d < D; ++d)
                    if (st[L.off_rowobs + d] > 0.0) {
                        s += st[L.off_WW + ((int64_t)d * KP + k) * KP + k];
                        ++de;
                    }
                const double a = A.a0a + 0.5 * de, b = A.b0a + 0.5 * s;
                st[L.off_alpha + 0 * KP + k] = a;
                st[L.off_alpha + 1 * KP + k] = b;
                st[L.off_alpha + 2 * KP + k] = a / b;
                st[L.off_alpha + 3 * KP + k] = vmp_digamma(a) - log(b);
            }
        } else if (op == VMP_MPCA_OP_ELBO) {
            const double tau = st[L.off_tau + 2], ltau = st[L.off_tau + 3];
            // W term: sum_d [ 1/2 sum_k <log a_k> - 1/2 sum_k <a_k> <ww>_d[k][k] + 1/2 log|Cov_d| + K/2 ]
            double sw = 0.0;
            for (int e = tid; e < D * K; e += NT) {
                const int d = e / K, k = e - d * K;
                if (st[L.off_rowobs + d] > 0.0)
                    sw += 0.5 * st[L.off_alpha + 3 * KP + k]
                          - 0.5 * st[L.off_alpha + 2 * KP + k]
                                * st[L.off_WW + ((int64_t)d * KP + k) * KP + k];
            }
            for (int d = tid; d < D; d += NT)
                if (st[L.off_rowobs + d] > 0.0) sw += 0.5 * st[L.off_ldW + d] + 0.5 * K;
            sw = block_sum<NT>(sw, red);
            double sa = 0.0;
            for (int k = tid; k < K; k += NT)
                sa += gamma_term(A.a0a, A.b0a, st[L.off_alpha + k], st[L.off_alpha + KP + k],
                                 st[L.off_alpha + 2 * KP + k], st[L.off_alpha + 3 * KP + k]);
            sa = block_sum<NT>(sa, red);
            if (tid == 0) {
                const double N = sc[SC_N];
                const double LY = sc[SC_NOBS] * (-0.5 * 1.8378770664093453 + 0.5 * ltau)
                                  - 0.5 * tau * sc[SC_RESID];
                const double LX = -0.5 * A.x_prec * sc[SC_TRXX] + 0.5 * sc[SC_LDX]
                                  + N * (0.5 * K * log(A.x_prec) + 0.5 * K);
                const double Lt = gamma_term(A.a0t, A.b0t, st[L.off_tau], st[L.off_tau + 1], tau, ltau);
                double *Lo = st + L.off_L;
                Lo[0] = LY;
                Lo[1] = LX;
                Lo[2] = sw;
                Lo[3] = Lt;
                Lo[4] = sa;
                Lo[5] = LY + LX + sw + Lt + sa;
            }
        }
    }
}

__global__ void __launch_bounds__(NT)
mpca_init_state_kernel(vmp_mpca_layout L, int K, double a0t, double b0t, double a0a, double b0a,
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

__global__ void mpca_set_scalar_kernel(double *p, double v) { p[0] = v; }

// <xx>_n of plates [0, nplates) of the chunk, unpacked to (nplates, K, K) (inspection: X.u[1])
__global__ void __launch_bounds__(NT)
mpca_unpack_kernel(const double *__restrict__ XXf, int PT, int K, int64_t nplates,
                   double *__restrict__ out)
{
    const int64_t total = nplates * K * K;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * NT) {
        const int64_t n = e / (K * K);
        const int ij = (int)(e - n * K * K), i = ij / K, j = ij - i * K;
        const int a = i > j ? i : j, b = i > j ? j : i, p = tri(a, b);
        out[e] = XXf[xxf_base(n, PT) + xxf_off(p)];
    }
}

// -------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------
constexpr int TSMAX = 9;      // column tiles per slice of mpca_stats (accumulators: 2 x 9 tiles)

int stats_slices(const mpca_dims &m) { return (m.CT + TSMAX - 1) / TSMAX; }

int64_t grid_cap(vmp_ctx *ctx, int per_cu) { return (int64_t)ctx->num_cu * per_cu; }

int32_t check_dims(vmp_ctx *ctx, int D, int K)
{
    VMP_REQUIRE(ctx, D >= 1 && K >= 1, VMP_ERR_INVALID, "bad dims D=%d K=%d", D, K);
    VMP_REQUIRE(ctx, D <= 128 && K <= 32, VMP_ERR_UNSUPPORTED,
                "the fused missing-data PCA block supports D <= 128, K <= 32 (got D=%d, K=%d)", D, K);
    return VMP_OK;
}

#define MPCA_FOR_EACH(M) M(1, 1) M(2, 1) M(4, 1) M(1, 2) M(2, 2) M(4, 2)

}  // namespace

extern "C" {

int32_t vmp_mpca_get_layout(int32_t D, int32_t K, vmp_mpca_layout *out)
{
    if (!out || D < 1 || K < 1) return VMP_ERR_INVALID;
    if (D > 128 || K > 32) return VMP_ERR_UNSUPPORTED;
    fill_layout(D, K, out);
    return VMP_OK;
}

int32_t vmp_mpca_sizes(vmp_ctx *ctx, int32_t D, int32_t K, int64_t N, int64_t chunk,
                       vmp_mpca_sizes_t *out)
{
    VMP_REQUIRE(ctx, ctx && out, VMP_ERR_INVALID, "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    VMP_REQUIRE(ctx, N >= 0 && chunk >= 32 && chunk % 32 == 0, VMP_ERR_INVALID,
                "the chunk must be a positive multiple of 32 plates");
    const mpca_dims m = make_dims(D, K);
    const int64_t ntiles = (N + TN - 1) / TN;
    out->ymt_doubles = (ntiles > 0 ? ntiles : 1) * m.DP * TN;
    out->mask_words = (ntiles > 0 ? ntiles : 1) * 2 * 64;
    out->xm_doubles = (ntiles > 0 ? ntiles : 1) * TN * m.KP;
    out->lam_doubles = chunk * m.LR;
    out->xxf_doubles = ((chunk + 3) / 4) * ((m.PT + 1) / 2) * 128;
    // partial sums: mpca_stats (workgroups x DP x LR), sweep / prepare scalars
    const int64_t gst = grid_cap(ctx, 2) / stats_slices(m);
    int64_t p = gst * m.DP * m.LR;
    const int64_t p2 = grid_cap(ctx, 16) * 4;
    const int64_t p3 = grid_cap(ctx, 8) * m.KP * m.KP;     // sum_n <xx>_n per sweep workgroup
    out->workspace_doubles = p + p2 + p3 + 64;
    return VMP_OK;
}

int32_t vmp_mpca_init_state(vmp_ctx *ctx, int32_t D, int32_t K, double a0_tau, double b0_tau,
                            double a0_alpha, double b0_alpha, double *state)
{
    VMP_REQUIRE(ctx, ctx && state, VMP_ERR_INVALID, "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    VMP_REQUIRE(ctx, a0_tau > 0 && b0_tau > 0 && a0_alpha > 0 && b0_alpha > 0, VMP_ERR_INVALID,
                "Gamma prior parameters must be positive");
    vmp_mpca_layout L;
    fill_layout(D, K, &L);
    VMP_HIP_CHECK(ctx, hipMemsetAsync(state, 0, (size_t)L.total * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(mpca_init_state_kernel, dim3(1), dim3(NT), 0, ctx->stream, L, K, a0_tau,
                       b0_tau, a0_alpha, b0_alpha, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_mpca_prepare(vmp_ctx *ctx, const double *Y, int64_t ldy, const uint8_t *mask,
                         int64_t ldm, int64_t N, int32_t D, int32_t K, double *Ymt, uint32_t *Mb1,
                         uint32_t *Mb2, double *state, void *workspace)
{
    VMP_REQUIRE(ctx, ctx && Y && Ymt && Mb1 && Mb2 && state && workspace, VMP_ERR_INVALID,
                "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    VMP_REQUIRE(ctx, ldy >= N && (!mask || ldm >= N), VMP_ERR_INVALID, "leading dimension < N");
    const mpca_dims m = make_dims(D, K);
    vmp_mpca_layout L;
    fill_layout(D, K, &L);
    const int64_t ntiles = (N + TN - 1) / TN;
    double *partial = reinterpret_cast<double *>(workspace);
    int64_t g = ntiles < grid_cap(ctx, 8) ? ntiles : grid_cap(ctx, 8);
    if (g < 1) g = 1;
    hipLaunchKernelGGL(mpca_prepare_kernel, dim3((unsigned)g), dim3(NT), 0, ctx->stream, Y, ldy,
                       mask, ldm, N, D, m.DP, Ymt, Mb1, Mb2, partial, ntiles);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    launch_reduce(ctx->stream, partial, (int)g, 2, 2, state + L.off_scal + SC_SYY, 0);
    launch_reduce(ctx->stream, partial + 2 * g, (int)g, m.DP, m.DP, state + L.off_rowobs, 0);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // extern "C"

namespace {

struct chunk_streams {
    hipStream_t sL, sS, sG;          // precision GEMM, sweep, statistics GEMM
    hipEvent_t eL, eS, eG;           // recorded after each stage (null: one stream, in order)
    int wgs_lambda, wgs_sweep, wgs_stats;   // workgroups per CU each stage may occupy
    bool timed;
};

// One chunk of plates: the three stages on their streams.  With one stream (all three equal, no
// events) this is the in-order form; with three, the caller pipelines consecutive chunks so that
// the VALU-bound sweep of chunk c runs beside the MFMA-bound GEMMs of chunks c+1 and c-1.
int32_t run_chunk(vmp_ctx *ctx, const mpca_dims &m, const vmp_mpca_layout &L, int K, int64_t n0,
                  int64_t nplates, int flags, double x_prec, const double *Ymt,
                  const uint32_t *Mb1, const uint32_t *Mb2, double *Xm, double *Lam, double *XXf,
                  double *state, void *workspace, const chunk_streams &cs)
{
    const bool first = (flags & VMP_MPCA_FIRST) != 0, inspect = (flags & VMP_MPCA_INSPECT) != 0;
    const bool from_value = (flags & (VMP_MPCA_FROM_VALUE | VMP_MPCA_PRIOR)) != 0;
    const double xx_diag = (flags & VMP_MPCA_PRIOR) ? 1.0 / x_prec : 0.0;
    if (nplates <= 0) {
        // an empty plate (N = 0 on this rank): the statistics of the pass are zero
        if (first && !inspect) {
            VMP_HIP_CHECK(ctx, hipMemsetAsync(state + L.off_scal + SC_TRXX, 0, 2 * sizeof(double),
                                              cs.sG));
            VMP_HIP_CHECK(ctx, hipMemsetAsync(state + L.off_Sxx, 0,
                                              (size_t)(m.KP * m.KP) * sizeof(double), cs.sG));
            VMP_HIP_CHECK(ctx, hipMemsetAsync(state + L.off_M, 0,
                                              (size_t)m.DP * m.LR * sizeof(double), cs.sG));
        }
        return VMP_OK;
    }
    const int64_t sub0 = n0 / 16, nsub = (nplates + 15) / 16;
    double *partial = reinterpret_cast<double *>(workspace);
    const int ns = stats_slices(m);
    const int64_t gst_wg = grid_cap(ctx, 2) / ns;
    double *pscal = partial + gst_wg * m.DP * m.LR;
    double *psxx = pscal + grid_cap(ctx, 16) * 4;
    hipEvent_t *ev = (ctx->timing && cs.timed) ? vmp_next_events(ctx) : nullptr;
    // ---- stage 1: Lam~ = mask^T . panel ---------------------------------------------------------
    // the fused form (round 5): stage 1 runs inside the per-plate kernel
    const bool fused = !from_value && !inspect && vmp_tune_get("mpca_blk4", 1) != 0
                       && vmp_tune_get("mpca_fuse", 1) != 0 && (K == 16 || K == 32);
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[0], cs.sL));
    if (!from_value && !fused) {
        constexpr int NSUB = 2;
        int64_t g = (nsub + NSUB - 1) / NSUB;
        if (g > grid_cap(ctx, cs.wgs_lambda)) g = grid_cap(ctx, cs.wgs_lambda);
#define MPCA_CASE(db, kt)                                                                          \
    if (m.DP == 32 * db && m.KT == kt)                                                             \
        hipLaunchKernelGGL((mpca_lambda_kernel<db, kt, NSUB>), dim3((unsigned)g), dim3(NT), 0,     \
                           cs.sL, Ymt, Mb1, state + L.off_panel_x, sub0, nsub, nplates, Lam);      \
    else
        MPCA_FOR_EACH(MPCA_CASE) { return VMP_ERR_UNSUPPORTED; }
#undef MPCA_CASE
        VMP_HIP_CHECK(ctx, hipGetLastError());
    }
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[1], cs.sL));
    if (cs.eL) {
        VMP_HIP_CHECK(ctx, hipEventRecord(cs.eL, cs.sL));
        VMP_HIP_CHECK(ctx, hipStreamWaitEvent(cs.sS, cs.eL, 0));
    }
    // ---- stage 2: per-plate sweep ---------------------------------------------------------------
    hipStream_t s = cs.sS;
    int nm = vmp_tune_get("mpca_sweep_nm", 2);
    if (m.KT == 1 || from_value) nm = 1;
    int64_t gs = (nplates + 4 * nm - 1) / (4 * nm);
    const int64_t gs_cap = grid_cap(ctx, cs.wgs_sweep);
    if (gs > gs_cap) gs = gs_cap;
    if (gs < 1) gs = 1;
#define MPCA_SWEEP(kt, fv, nmm, oc)                                                                \
    hipLaunchKernelGGL((mpca_sweep_kernel<kt, fv, nmm, oc>), dim3((unsigned)gs), dim3(NT), 0, s,   \
                       Lam, m.LR, n0, nplates, K, x_prec, xx_diag, state + L.off_scal + SC_TAUX,   \
                       XXf, Xm, inspect ? 0 : 1, pscal, psxx)
    if (!from_value && vmp_tune_get("mpca_blk4", 1) != 0) {
        // default: four plates per wavefront on the 4x4x4 matrix instruction (mpca_blk4_kernel)
        gs = (nplates + 15) / 16;
        const int64_t cap = grid_cap(ctx, vmp_tune_get("mpca_blk4_wgs", 2));
        if (gs > cap) gs = cap;
        if (gs < 1) gs = 1;
        const int nb = (K + 3) / 4;
#define MPCA_BLK4(NBV)                                                                             \
    case NBV:                                                                                      \
        if (K == 4 * NBV)                                                                          \
            hipLaunchKernelGGL((mpca_blk4_kernel<NBV, true, 0>), dim3((unsigned)gs), dim3(NT), 0, s, \
                               Lam, n0, nplates, K, x_prec, state + L.off_scal + SC_TAUX, XXf, Xm, \
                               inspect ? 0 : 1, pscal, psxx, nullptr, nullptr, nullptr, 0);        \
        else                                                                                       \
            hipLaunchKernelGGL((mpca_blk4_kernel<NBV, false, 0>), dim3((unsigned)gs), dim3(NT), 0, s, \
                               Lam, n0, nplates, K, x_prec, state + L.off_scal + SC_TAUX, XXf, Xm, \
                               inspect ? 0 : 1, pscal, psxx, nullptr, nullptr, nullptr, 0);        \
        break;
        if (fused) {
            // the precision GEMM inside the per-plate kernel (no Lam~ in HBM): K = 16 or 32
            const int stg_n = vmp_tune_get("mpca_fuse_stagger", 2);
#define MPCA_FUSED(NBV, db)                                                                        \
    if (nb == NBV && m.DP == 32 * db)                                                              \
        hipLaunchKernelGGL((mpca_blk4_kernel<NBV, true, db>), dim3((unsigned)gs), dim3(NT), 0, s,  \
                           Lam, n0, nplates, K, x_prec, state + L.off_scal + SC_TAUX, XXf, Xm, 1,  \
                           pscal, psxx, Ymt, Mb1, state + L.off_panel_x, stg_n);                   \
    else
            MPCA_FUSED(4, 1) MPCA_FUSED(4, 2) MPCA_FUSED(4, 4)
            MPCA_FUSED(8, 1) MPCA_FUSED(8, 2) MPCA_FUSED(8, 4) { return VMP_ERR_UNSUPPORTED; }
#undef MPCA_FUSED
        } else
        switch (nb) {
            MPCA_BLK4(1) MPCA_BLK4(2) MPCA_BLK4(3) MPCA_BLK4(4)
            MPCA_BLK4(5) MPCA_BLK4(6) MPCA_BLK4(7) MPCA_BLK4(8)
        default:
            return VMP_ERR_UNSUPPORTED;
        }
#undef MPCA_BLK4
    } else if (m.KT == 1) {
        if (from_value) MPCA_SWEEP(1, true, 1, 2);
        else MPCA_SWEEP(1, false, 1, 2);
    } else if (from_value) {
        MPCA_SWEEP(2, true, 1, 2);
    } else if (vmp_tune_get("mpca_rows", 0) != 0) {
        // measured alternative, not the default: see the note above mpca_rows_kernel
        gs = (nplates + 3) / 4;
        const int64_t cap = grid_cap(ctx, 7);
        if (gs > cap) gs = cap;
        if (K == 32)
            hipLaunchKernelGGL((mpca_rows_kernel<true>), dim3((unsigned)gs), dim3(RW_NT), 0, s, Lam,
                               n0, nplates, K, x_prec, state + L.off_scal + SC_TAUX, XXf, Xm,
                               inspect ? 0 : 1, pscal, psxx);
        else
            hipLaunchKernelGGL((mpca_rows_kernel<false>), dim3((unsigned)gs), dim3(RW_NT), 0, s, Lam,
                               n0, nplates, K, x_prec, state + L.off_scal + SC_TAUX, XXf, Xm,
                               inspect ? 0 : 1, pscal, psxx);
    } else if (nm == 2) MPCA_SWEEP(2, false, 2, 2);
    else MPCA_SWEEP(2, false, 1, 2);
#undef MPCA_SWEEP
    VMP_HIP_CHECK(ctx, hipGetLastError());
    if (inspect) return VMP_OK;
    // tr<xx>, log|Cov|, status, sum_n <xx>_n
    launch_reduce(s, pscal, (int)gs, 3, 2, state + L.off_scal + SC_TRXX, first ? 0 : 1);
    launch_reduce(s, pscal + 2, (int)gs, 3, 1, state + L.off_scal + SC_STATUS, 1);
    launch_reduce(s, psxx, (int)gs, (int64_t)(m.KP * m.KP), (int64_t)(m.KP * m.KP),
                  state + L.off_Sxx, first ? 0 : 1);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[2], s));
    if (cs.eS) {
        VMP_HIP_CHECK(ctx, hipEventRecord(cs.eS, cs.sS));
        VMP_HIP_CHECK(ctx, hipStreamWaitEvent(cs.sG, cs.eS, 0));
    }
    // ---- stage 3: M_d, r_d --------------------------------------------------------------------------
    s = cs.sG;
    hipEvent_t *ev2 = (ctx->timing && cs.timed) ? vmp_next_events(ctx) : nullptr;
    if (ev2) VMP_HIP_CHECK(ctx, hipEventRecord(ev2[0], s));
    {
        // packed columns: column tiles split over the wavefronts; r_d: its own small kernel.
        // Both write disjoint columns of the same per-workgroup partial rows.
        // default: mpca_stats3 (4x4x4 instruction, a wavefront owns 32 rows x 3 tile pairs);
        // vmp_tune_set("mpca_stats3", 0): the 16x16x4 column-split form
        const int use3 = vmp_tune_get("mpca_stats3", 1);
        constexpr int NP3 = 3;
        const int pt2 = (m.PT + 1) / 2;
        const int ns2 = use3 ? (pt2 + NP3 - 1) / NP3 : (m.PT + 7) / 8;
        const int64_t npair = (nsub + 1) / 2;
        int64_t gw = grid_cap(ctx, cs.wgs_stats) / ns2;
        if (gw > gst_wg) gw = gst_wg;
        if (gw > npair) gw = npair;
        if (gw < 1) gw = 1;
        const int64_t len = (int64_t)m.DP * m.LR;
        const dim3 grid((unsigned)(gw * ns2));
#define MPCA_CASE(db, kt)                                                                       \
    if (m.DP == 32 * db && m.KT == kt) {                                                        \
        if (use3)                                                                               \
            hipLaunchKernelGGL((mpca_stats3_kernel<db, kt, NP3>), grid, dim3(NT), 0, s, Mb2,    \
                               XXf, sub0, nsub, ns2, partial);                                  \
        else                                                                                    \
            hipLaunchKernelGGL((mpca_stats2_kernel<db, kt>), grid, dim3(NT), 0, s, Mb2, XXf,    \
                               sub0, nsub, ns2, partial);                                       \
        hipLaunchKernelGGL((mpca_ryx_kernel<db, kt>), dim3((unsigned)gw), dim3(NT), 0, s, Ymt,  \
                           Xm, n0 / TN, (nplates + TN - 1) / TN, partial, m.LR, 16 * m.PT);     \
    } else
        MPCA_FOR_EACH(MPCA_CASE) { return VMP_ERR_UNSUPPORTED; }
#undef MPCA_CASE
        VMP_HIP_CHECK(ctx, hipGetLastError());
        if (ev2) VMP_HIP_CHECK(ctx, hipEventRecord(ev2[1], s));
        launch_reduce(s, partial, (int)gw, len, len, state + L.off_M, first ? 0 : 1);
        VMP_HIP_CHECK(ctx, hipGetLastError());
        if (ev2) VMP_HIP_CHECK(ctx, hipEventRecord(ev2[2], s));
    }
    if (cs.eG) VMP_HIP_CHECK(ctx, hipEventRecord(cs.eG, cs.sG));
    return VMP_OK;
}

}  // namespace

extern "C" {

// One chunk of X.update() or of the statistics of a given <x>: plates [n0, n0 + nplates), n0 a
// multiple of 32, in order on the context's stream.
int32_t vmp_mpca_x_chunk(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n0, int64_t nplates,
                         int32_t flags, double x_prec, const double *Ymt,
                         const uint32_t *Mb1, const uint32_t *Mb2, double *Xm, double *Lam,
                         double *XXf, double *state, void *workspace)
{
    VMP_REQUIRE(ctx, ctx && Ymt && Mb1 && Mb2 && Xm && Lam && XXf && state && workspace,
                VMP_ERR_INVALID, "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    VMP_REQUIRE(ctx, n0 >= 0 && n0 % 32 == 0 && nplates >= 0, VMP_ERR_INVALID,
                "a chunk starts at a multiple of 32 plates");
    if (nplates == 0) return VMP_OK;
    const mpca_dims m = make_dims(D, K);
    vmp_mpca_layout L;
    fill_layout(D, K, &L);
    chunk_streams cs = {ctx->stream, ctx->stream, ctx->stream, nullptr, nullptr, nullptr, 2, 8, 2, true};
    return run_chunk(ctx, m, L, K, n0, nplates, flags, x_prec, Ymt, Mb1, Mb2, Xm, Lam, XXf, state,
                     workspace, cs);
}

// All chunks of one pass over the plates [0, N).  nsets = 1: in order on the context's stream.
// nsets = 2 (Lam and XXf hold TWO chunks each): consecutive chunks are pipelined over three
// library streams -- the precision GEMM of chunk c+1 and the statistics GEMM of chunk c-1
// (matrix cores) run beside the sweep of chunk c (vector ALU); the statistics are accumulated in
// chunk order on one stream, so the result does not depend on the interleaving.
int32_t vmp_mpca_x_pass(vmp_ctx *ctx, int32_t D, int32_t K, int64_t N, int64_t chunk, int32_t nsets,
                        int32_t flags, double x_prec, const double *Ymt, const uint32_t *Mb1,
                        const uint32_t *Mb2, double *Xm, double *Lam, double *XXf, double *state,
                        void *workspace)
{
    VMP_REQUIRE(ctx, ctx && Ymt && Mb1 && Mb2 && Xm && Lam && XXf && state && workspace,
                VMP_ERR_INVALID, "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    VMP_REQUIRE(ctx, N >= 0 && chunk >= 32 && chunk % 32 == 0 && (nsets == 1 || nsets == 2),
                VMP_ERR_INVALID, "bad chunking");
    const mpca_dims m = make_dims(D, K);
    vmp_mpca_layout L;
    fill_layout(D, K, &L);
    const int64_t lam_n = chunk * m.LR, xxf_n = ((chunk + 3) / 4) * ((m.PT + 1) / 2) * 128;
    const bool pipelined = nsets == 2 && N > chunk && !(flags & VMP_MPCA_INSPECT)
                           && vmp_tune_get("mpca_streams", 1) != 0;
    if (!pipelined) {
        chunk_streams cs = {ctx->stream, ctx->stream, ctx->stream, nullptr, nullptr, nullptr, 2, 8, 2, true};
        int f = flags | VMP_MPCA_FIRST;
        for (int64_t n0 = 0; n0 < (N > 0 ? N : 1); n0 += chunk) {
            const int64_t npl = (N - n0) < chunk ? (N - n0) : chunk;
            rc = run_chunk(ctx, m, L, K, n0, npl, f, x_prec, Ymt, Mb1, Mb2, Xm, Lam, XXf, state,
                           workspace, cs);
            if (rc != VMP_OK) return rc;
            f = flags & ~VMP_MPCA_FIRST;
        }
        return VMP_OK;
    }
    // three streams, two scratch sets
    if (!ctx->ms[0]) {
        for (int i = 0; i < 3; ++i)
            VMP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->ms[i], hipStreamNonBlocking));
        for (int i = 0; i < VMP_NME; ++i)
            VMP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->me[i], hipEventDisableTiming));
    }
    hipEvent_t eStart = ctx->me[6], eEnd = ctx->me[7];
    VMP_HIP_CHECK(ctx, hipEventRecord(eStart, ctx->stream));
    for (int i = 0; i < 3; ++i) VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->ms[i], eStart, 0));
    int f = flags | VMP_MPCA_FIRST;
    int64_t c = 0;
    for (int64_t n0 = 0; n0 < N; n0 += chunk, ++c) {
        const int64_t npl = (N - n0) < chunk ? (N - n0) : chunk;
        const int set = (int)(c & 1);
        hipEvent_t eL = ctx->me[0 + set], eS = ctx->me[2 + set], eG = ctx->me[4 + set];
        if (c >= 2) {
            // Lam[set] is free once the sweep of chunk c-2 has read it; XXf[set] once the
            // statistics of chunk c-2 have
            VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->ms[0], eS, 0));
            VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->ms[1], eG, 0));
        }
        chunk_streams cs = {ctx->ms[0], ctx->ms[1], ctx->ms[2], eL, eS, eG,
                            vmp_tune_get("mpca_lambda_wgs", 1), vmp_tune_get("mpca_sweep_wgs", 4),
                            vmp_tune_get("mpca_stats_wgs", 1), false};
        rc = run_chunk(ctx, m, L, K, n0, npl, f, x_prec, Ymt, Mb1, Mb2, Xm, Lam + set * lam_n,
                       XXf + set * xxf_n, state, workspace, cs);
        if (rc != VMP_OK) return rc;
        f = flags & ~VMP_MPCA_FIRST;
    }
    // the caller's stream continues after the last sweep and the last statistics
    VMP_HIP_CHECK(ctx, hipEventRecord(eEnd, ctx->ms[1]));
    VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, eEnd, 0));
    VMP_HIP_CHECK(ctx, hipEventRecord(eEnd, ctx->ms[2]));
    VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, eEnd, 0));
    return VMP_OK;
}

// Snapshot what X.update() reads of its Markov blanket (the panel of <ww>, <w> and <tau>): the
// chunks of one pass must all see the same values, and X.u[1] is re-derived from it later.
int32_t vmp_mpca_x_begin(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_plates, double *state)
{
    VMP_REQUIRE(ctx, ctx && state, VMP_ERR_INVALID, "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    const mpca_dims m = make_dims(D, K);
    vmp_mpca_layout L;
    fill_layout(D, K, &L);
    const size_t pbytes = (size_t)m.CT * (m.DQ / 2) * 128 * sizeof(double);
    VMP_HIP_CHECK(ctx, hipMemcpyAsync(state + L.off_panel_x, state + L.off_panel, pbytes,
                                      hipMemcpyDeviceToDevice, ctx->stream));
    VMP_HIP_CHECK(ctx, hipMemcpyAsync(state + L.off_scal + SC_TAUX, state + L.off_tau + 2,
                                      sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(mpca_set_scalar_kernel, dim3(1), dim3(1), 0, ctx->stream,
                       state + L.off_scal + SC_N, (double)n_plates);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_mpca_update_w(vmp_ctx *ctx, int32_t D, int32_t K, int32_t mode, double *state)
{
    VMP_REQUIRE(ctx, ctx && state, VMP_ERR_INVALID, "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    VMP_REQUIRE(ctx, mode >= 0 && mode <= 2, VMP_ERR_INVALID, "bad mode %d", mode);
    const mpca_dims m = make_dims(D, K);
    vmp_mpca_layout L;
    fill_layout(D, K, &L);
    const dim3 grid((unsigned)((D + 3) / 4));
    if (m.KT == 1)
        hipLaunchKernelGGL(mpca_w_kernel<1>, grid, dim3(NT), 0, ctx->stream, L, D, K, m.DQ, mode, state);
    else
        hipLaunchKernelGGL(mpca_w_kernel<2>, grid, dim3(NT), 0, ctx->stream, L, D, K, m.DQ, mode, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_mpca_small_ops(vmp_ctx *ctx, int32_t D, int32_t K, double x_prec, double a0_tau,
                           double b0_tau, double a0_alpha, double b0_alpha, int32_t nops,
                           const int32_t *ops, double *state)
{
    VMP_REQUIRE(ctx, ctx && state && ops, VMP_ERR_INVALID, "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    VMP_REQUIRE(ctx, nops >= 1 && nops <= 8, VMP_ERR_INVALID, "1..8 operations per call");
    small_args A;
    fill_layout(D, K, &A.L);
    A.D = D;
    A.K = K;
    A.nops = nops;
    for (int i = 0; i < nops; ++i) {
        VMP_REQUIRE(ctx, ops[i] >= VMP_MPCA_OP_TAU && ops[i] <= VMP_MPCA_OP_ELBO, VMP_ERR_INVALID,
                    "unknown operation %d", ops[i]);
        A.ops[i] = ops[i];
    }
    A.x_prec = x_prec;
    A.a0t = a0_tau;
    A.b0t = b0_tau;
    A.a0a = a0_alpha;
    A.b0a = b0_alpha;
    hipLaunchKernelGGL(mpca_small_kernel, dim3(1), dim3(NT), 0, ctx->stream, A, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_mpca_unpack_xx(vmp_ctx *ctx, int32_t D, int32_t K, int64_t nplates, const double *XXf,
                           double *out)
{
    VMP_REQUIRE(ctx, ctx && XXf && out, VMP_ERR_INVALID, "null argument");
    int32_t rc = check_dims(ctx, D, K);
    if (rc != VMP_OK) return rc;
    if (nplates <= 0) return VMP_OK;
    const mpca_dims m = make_dims(D, K);
    int64_t g = (nplates * K * K + NT - 1) / NT;
    if (g > grid_cap(ctx, 8)) g = grid_cap(ctx, 8);
    hipLaunchKernelGGL(mpca_unpack_kernel, dim3((unsigned)g), dim3(NT), 0, ctx->stream, XXf, m.PT, K,
                       nplates, out);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // extern "C"
