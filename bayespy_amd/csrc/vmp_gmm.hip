// vmp_gmm.hip -- fused full-covariance Gaussian-mixture VB block for gfx950.
//
// Model block of bayespy/demos/mog.py:17-64:
//   Y = Mixture(z, Gaussian, mu, Lambda), z = Categorical(alpha), alpha = Dirichlet(a0),
//   mu = GaussianARD(0, beta0, shape=(D,), plates=(K,)), Lambda = Wishart(n0, V0, plates=(K,)),
// Y fully observed.  Replaces the NumPy call sites E12, E18-E24 of SURVEY.md 2.2:
//   mixture.py:53-293 (+ expfamily.py:45-61 compute_logpdf), multinomial.py:83-128,
//   utils/misc.py:1366-1401 (normalized_exp), gaussian.py:293-573, :2374-2527,
//   wishart.py:118-225, dirichlet.py:107-231, expfamily.py:400-480.
//
// One pass over Y per VB iteration (issued by z.update()):
//   phase 1  Phi(K x 16) = C (K x F2) * feat2(y)   (fp64 MFMA; symmetric pairs once, their
//                                                   coefficients folded: 12 k-steps, not 19)
//   softmax  r = normalized_exp(Phi) per column (reference recipe), r written (N, K)
//   phase 2  T (K x F2) += r * feat2(y)^T          feat2 = [y_a y_b (a<=b), y_d, 1]
// T = [R_k, sum r y, sum r y y^T] are the messages to mu / Lambda / alpha, i.e. the
// plate sums of mixture.py:126-158 + node.py:650 that the reference materialises
// as (N, K, D, D) arrays.  D <= 32, K <= 64 built (D <= 8: the tuned instances of config 3;
// 9 <= D <= 16: the same pass with the features formed twice instead of staged in LDS, the
// clusters split over a wavefront pair at K > 32; 17 <= D <= 32: vmp_gmm_wide.hip, coefficient
// fragments streamed from L2, one cluster tile per wavefront).
//
// Layout: Y (N, D) row-major as in the reference (plates (N,), dims (D,)); a wave reads
// 16 consecutive rows = one contiguous block.  r (N, K) row-major, written as whole rows.
#include "vmp_gmm_dev.h"

namespace {

constexpr int NT = 256;
constexpr int MAXK = 64, MAXD = 32;

inline int round_pow2(int x, int unit)
{
    int b = (x + unit - 1) / unit, p = 1;
    while (p < b) p <<= 1;
    return p * unit;
}

inline int n_feat2(int D) { return D * (D + 1) / 2 + D + 1; }

inline void fill_layout(int D, int K, vmp_gmm_layout *L)
{
    const int64_t DP = round_pow2(D, 4), KP = round_pow2(K, 16);
    const int64_t FS = 1 + D + (int64_t)D * D;
    // D > 16 (vmp_gmm_wide.hip): whole groups of four feature tiles
    const int64_t F2P = D > 16 ? (n_feat2(D) + 63) / 64 * 64 : (n_feat2(D) + 15) / 16 * 16;
    const int64_t FP = F2P;        // coefficient rows of C use the compact feature order
    int64_t o = 0;
    L->DP = DP; L->KP = KP; L->FS = FS; L->FP = FP;
    L->F2P = F2P;
    L->off_T = o;         L->len_T = KP * FS; o += L->len_T;
    L->off_zs = o;        o += 8;
    L->off_alpha = o;     o += 2 * KP;
    L->off_mu = o;        o += KP * D;
    L->off_Cmu = o;       o += KP * D * D;
    L->off_logdetLmu = o; o += KP;
    L->off_nk = o;        o += KP;
    L->off_Vk = o;        o += KP * D * D;
    L->off_Lam = o;       o += KP * D * D;
    L->off_logdetLam = o; o += KP;
    L->off_logdetV = o;   o += KP;
    L->off_C = o;         o += KP * FP;
    L->off_prior = o;     o += KP + 8 + (int64_t)D * D;   // alpha0[KP], beta0, n0, logdetV0, -, V0
    L->off_scal = o;      o += 8;
    L->off_L = o;         o += 8;
    L->total = (o + 7) / 8 * 8;
}

// ---------------------------------------------------------------------------
// The pass.  DPT = DP/4 in {1,2}, KT = KP/16 in {1,2,4}, FT2 = F2P/16 in {1,2,3}.
// FROM_LABELS: responsibilities are the one-hot of given labels
// (z.initialize_from_value, categorical.py:30-46) instead of the softmax.
//
// One workgroup of eight wavefronts per CU (two per SIMD, 256 VGPRs each: the phase-2
// accumulators alone are 96); a wavefront owns 16-column tiles.  Per tile:
//   features   lane (g, n) forms feat[4q + g] of column n from the y tile in LDS (per-lane
//              LDS addresses precomputed), feeds it to phase 1 as the B operand AND stores it
//              once in LDS in the operand order of phase 2
//   phase 1    Phi(K x 16) = C * feat        softmax: one table exponential per element,
//              p = e / sum (the reference's second renormalisation changes p by <= 2 ulp and
//              is not repeated), sum_n lse_n accumulated as sum max + log of a running product
//   r tile     k-major in LDS (stride 18: conflict-free for both the writes and the A-operand
//              reads), aliased with the y tile; rows written to HBM as whole (N, K) rows
//   phase 2    T += r * feat^T
// sum_nk r phi (bound term of z) is not accumulated here: it equals <C, T> exactly and is
// formed from the reduced statistics (gmm_rphi_kernel).
//
// D > 8 (REGEN): the feature tile (16 x F2P doubles per wavefront; 20 KB at D = 16) is not staged:
// phase 2 forms feature ft*16 + l15 of column 4q + g again from the y tile (two LDS reads and
// one multiply per KT matrix instructions), the y tile keeps its own storage beside the r
// tile, and the per-lane factor offsets (two 16-bit byte offsets per feature) live in an LDS
// table shared by the wavefronts instead of 2 KS1 address registers.  NW = 4 (one wavefront per
// SIMD, 512 registers) where the KTL x FT2 accumulator tiles do not leave room for two.
//
// KS = 2 (D > 8 and K > 32): the clusters are split between the two wavefronts of a pair that
// walk the same tiles (KTL = KT/2 cluster tiles each: the 4 x 10 accumulator tiles of D = 16,
// K = 64 are 320 registers, more than one wavefront can hold beside the rest).  Each wavefront
// runs both phases for its own clusters on its own copy of the y tile; the pair exchanges only
// the softmax normalisers -- the column maxima and the column sums, 16 doubles each, through
// LDS with two workgroup barriers per tile (every wavefront of the grid walks the same number
// of tiles; the surplus ones are empty).  Sums are combined as (half 0) + (half 1) in both
// wavefronts, so the two halves of a row of r carry the same normaliser bit for bit.
// ---------------------------------------------------------------------------
template <int DPT, int KT, int FT2, bool FROM_LABELS, int NW, bool REGEN, int KS, bool M4 = false>
__global__ void __launch_bounds__(64 * NW, NW / 4)
gmm_pass_kernel(const double *__restrict__ Y, int64_t N, int D, int K,
                const double *__restrict__ Cmat, const int64_t *__restrict__ labels,
                double *__restrict__ Rout, double *__restrict__ P, int64_t ntiles)
{
    constexpr int DP = 4 * DPT, KP = 16 * KT;
    constexpr int YS = DP + 3;                    // y tile row stride (ONE at DP, ZERO at DP+1)
    constexpr int F2P = 16 * FT2;                 // compact features: y_a y_b (a<=b), y_d, 1
    constexpr int FS = F2P + 2;                   // feature tile row stride
    constexpr int KS1 = F2P / 4;                  // phase-1 k-steps over the same features
    constexpr int NTP = 64 * NW;
    constexpr int WAVES = NW;
    constexpr int GROUPS = NW / KS;               // tiles in flight per workgroup
    constexpr int KTL = KT / KS, KPL = 16 * KTL;  // cluster tiles / clusters of one wavefront
    static_assert(KS == 1 || (KS == 2 && KTL == 2 && REGEN), "cluster split: K > 32, D > 8 only");
    // y tile / r tile: aliased when the features are staged, side by side when phase 2 re-reads y
    constexpr int YR = REGEN ? TNC * YS + KPL * RS : ((TNC * YS > KPL * RS) ? TNC * YS : KPL * RS);
    constexpr int FTL = REGEN ? 0 : TNC * FS;     // staged feature tile
    constexpr int NCF = FROM_LABELS ? 0 : KT * KS1 * 64;
    constexpr int NTAB = FROM_LABELS ? 0 : 256;
    constexpr int NXCH = (KS == 2 && !FROM_LABELS) ? 2 * NW * 16 : 0;   // softmax normalisers
    constexpr int NFTAB = REGEN ? (KS1 + FT2) * 32 : 0;                 // factor offsets (u32)

    extern __shared__ double lds[];
    double *Cf = lds;                                            // KT*KS1*64
    double *tab = lds + NCF;                                     // 256
    double *xch = tab + NTAB;                                    // [2][NW][16]  (KS == 2)
    uint32_t *ftab = reinterpret_cast<uint32_t *>(xch + NXCH);   // [KS1 + FT2][64]  (REGEN)
    double *wbase = xch + NXCH + NFTAB;
    const int tid = threadIdx.x;
    const int l = tid & 63, l15 = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = (KS == 2) ? (w & 1) : 0;                      // which half of the clusters
    const int grp = (KS == 2) ? (w >> 1) : w;                    // which tile of the workgroup
    double *ytile = wbase + w * (YR + FTL);                      // [16][YS]
    double *rtile = REGEN ? ytile + TNC * YS : ytile;            // [KP][RS]   (after phase 1)
    double *ftile = ytile + YR;                                  // [16][FS]   (!REGEN)

    if (!FROM_LABELS) {
        // A fragments of phase 1: lane holds C[it*16 + l15][4q + g]
        for (int e = tid; e < KT * KS1 * 64; e += NTP) {
            const int lane = e & 63, fq = e >> 6;
            const int it = fq / KS1, q = fq - it * KS1;
            Cf[e] = Cmat[(int64_t)(it * 16 + (lane & 15)) * F2P + 4 * q + (lane >> 4)];
        }
        for (int e = tid; e < 256; e += NTP) tab[e] = VMP_EXP2_TAB[e];
    }
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(lds_f64 *)tab;

    // feature f of a column n is ytile[n][fa(f)] * ytile[n][fb(f)] (slot DP holds 1, DP+1 holds 0)
    auto feature = [&](int f, int &a, int &b) {
        const int npair = D * (D + 1) / 2;
        a = DP + 1; b = DP + 1;                      // zero feature
        if (f < npair) {
            int rem = f, aa = 0;
            while (rem >= D - aa) { rem -= D - aa; ++aa; }
            a = aa; b = aa + rem;
        } else if (f < npair + D) {
            a = f - npair; b = DP;                   // linear: y_d * 1
        } else if (f == npair + D) {
            a = DP; b = DP;                          // constant
        }
    };
    // LDS byte addresses of the two factors of feature 4q + g of this lane's column
    const uint32_t yrow_addr = (uint32_t)(uintptr_t)(lds_f64 *)(ytile + l15 * YS);
    // staged form: full addresses in registers; REGEN: byte offsets of both factors packed in
    // the table row [q][lane] (phase 1: feature 4q + g) / [KS1 + ft][lane] (phase 2: feature
    // ft*16 + l15 of the rows 4q + g of the y tile)
    uint32_t fa[REGEN ? 1 : KS1], fb[REGEN ? 1 : KS1];
    if constexpr (REGEN) {
        for (int e = tid; e < (KS1 + FT2) * 64; e += NTP) {
            const int lane = e & 63, row = e >> 6;
            int a, b;
            feature(row < KS1 ? 4 * row + (lane >> 4) : (row - KS1) * 16 + (lane & 15), a, b);
            ftab[e] = 8u * (uint32_t)a | (8u * (uint32_t)b) << 16;
        }
    } else {
#pragma unroll
        for (int q = 0; q < KS1; ++q) {
            int a, b;
            feature(4 * q + g, a, b);
            fa[q] = yrow_addr + 8u * (uint32_t)a;
            fb[q] = yrow_addr + 8u * (uint32_t)b;
        }
    }
    const uint32_t ftab_addr = (uint32_t)(uintptr_t)(lds_u32 *)(ftab + l);
    const uint32_t ygrp_addr = (uint32_t)(uintptr_t)(lds_f64 *)(ytile + g * YS);
    __syncthreads();

    v4f64 acc2[KTL][FT2];
#pragma unroll
    for (int it = 0; it < KTL; ++it)
#pragma unroll
        for (int ft = 0; ft < FT2; ++ft) acc2[it][ft] = v4f64{0.0, 0.0, 0.0, 0.0};
    double s_mx = 0.0, s_log = 0.0, prod = 1.0;
    int since = 0;

    const int64_t stride = (int64_t)gridDim.x * GROUPS;
    const int64_t tile0 = (int64_t)blockIdx.x * GROUPS + grp;
    // KS == 2: the barriers of the exchange need the same trip count in every wavefront
    const int64_t tile_end = (KS == 2) ? tile0 + (ntiles + stride - 1) / stride * stride : ntiles;
    for (int64_t tile = tile0; tile < tile_end; tile += stride) {
        const int64_t n0 = tile * TNC;
        const int64_t n = n0 + l15;
        const bool nok = n < N;
        // ---- y: lane (g, l15) owns y[n][4j + g] ------------------------------------
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int d = 4 * j + g;
            ytile[l15 * YS + d] = (nok && d < D) ? Y[n * D + d] : 0.0;
        }
        if (g == 0) {
            // constant slots (the r tile of the previous round overwrote them)
            ytile[l15 * YS + DP] = 1.0;
            ytile[l15 * YS + DP + 1] = 0.0;
        }
        lds_fence();

        // ---- features (+ phase 1: Phi = C * feat over the compact features) -------------
        v4f64 acc1[KTL];
#pragma unroll
        for (int it = 0; it < KTL; ++it) acc1[it] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < KS1; ++q) {
            // keep the operand reads next to their use (hoisted together they spill)
            if ((q & 3) == 0) asm volatile("" ::: "memory");
            double b;
            if constexpr (REGEN) {
                const uint32_t pk = lds_read_u32(ftab_addr + 256u * q);
                b = lds_read(yrow_addr + (pk & 0xffffu)) * lds_read(yrow_addr + (pk >> 16));
            } else {
                b = lds_read(fa[q]) * lds_read(fb[q]);
                ftile[l15 * FS + 4 * q + g] = b;
            }
            if (!FROM_LABELS) {
#pragma unroll
                for (int it = 0; it < KTL; ++it)
                    acc1[it] = mfma_f64(Cf[((kh * KTL + it) * KS1 + q) * 64 + l], b, acc1[it]);
            }
        }
        lds_fence();             // every lane is done with the y tile: the r tile may overwrite it

        if (!FROM_LABELS) {
            // ---- softmax over k for column n (utils/misc.py:1388-1401) ------------------
            // lane holds Phi[k = (kh*KTL + it)*16 + g + 4r][n]
            mfma_settle<KTL>(acc1);
            double mx = max_raw(acc1[0][0], acc1[0][1]);
            mx = max_raw(mx, max_raw(acc1[0][2], acc1[0][3]));
#pragma unroll
            for (int it = 1; it < KTL; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = max_raw(mx, acc1[it][r]);
            mx = max_raw(mx, __shfl_xor(mx, 16, 64));
            mx = max_raw(mx, __shfl_xor(mx, 32, 64));
            if constexpr (KS == 2) {
                // column maxima of the other half of the clusters
                if (g == 0) xch[w * 16 + l15] = mx;
                __syncthreads();
                mx = fmax(mx, xch[(w ^ 1) * 16 + l15]);
            }
            if (!isfinite(mx)) mx = 0.0;
            double s = 0.0;
#pragma unroll
            for (int h = 0; h < KTL; h += 2) {
                constexpr int NV = KTL >= 2 ? 8 : 4;
                double v[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] = acc1[h + (i >> 2)][i & 3];
                exp_tab_batch<NV>(v, mx, tab_addr);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    acc1[h + (i >> 2)][i & 3] = v[i];
                    s += v[i];
                }
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if constexpr (KS == 2) {
                // column sums: (half 0) + (half 1) in both wavefronts
                if (g == 0) xch[NW * 16 + w * 16 + l15] = s;
                __syncthreads();
                const double so = xch[NW * 16 + (w ^ 1) * 16 + l15];
                s = kh == 0 ? s + so : so + s;
            }
            // lse_n = mx + log s; the logarithm is taken of a running product (1 <= s <= KP)
            s_mx += nok ? mx : 0.0;
            prod *= nok ? s : 1.0;
            if (++since == 16) {
                s_log += log(prod);
                prod = 1.0;
                since = 0;
            }
            const double is = nok ? recip_small(s) : 0.0;
#pragma unroll
            for (int it = 0; it < KTL; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    rtile[(it * 16 + g + 4 * r) * RS + l15] = acc1[it][r] * is;
        } else {
            const int64_t lab = nok ? labels[n] : -1;
#pragma unroll
            for (int it = 0; it < KTL; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = it * 16 + g + 4 * r;              // row of this wavefront's r tile
                    rtile[k * RS + l15] = (lab == kh * KPL + k) ? 1.0 : 0.0;
                }
        }
        lds_fence();

        // ---- r -> HBM, whole rows (N, K) ------------------------------------------------
        if constexpr (KS == 1) {
#pragma unroll 4
            for (int rr = 0; rr < TNC; ++rr) {
                if (n0 + rr < N) {
                    for (int k = l; k < K; k += 64) Rout[(n0 + rr) * K + k] = rtile[k * RS + rr];
                }
            }
        } else {
            // this wavefront's 32 columns of two rows per instruction
            const int kk = l & 31, k = kh * KPL + kk;
#pragma unroll 4
            for (int rr = 0; rr < TNC; rr += 2) {
                const int row = rr + (l >> 5);
                if (n0 + row < N && k < K) Rout[(n0 + row) * K + k] = rtile[kk * RS + row];
            }
        }

        // ---- phase 2: T += r * feat2(y)^T  (contraction over the 16 columns) ---------------
#pragma unroll
        for (int q = 0; q < TNC / 4; ++q) {
            const int nn = 4 * q + g;
            if constexpr (REGEN) {
                double a[KTL];
#pragma unroll
                for (int it = 0; it < KTL; ++it) a[it] = rtile[(it * 16 + l15) * RS + nn];
                const uint32_t yq = ygrp_addr + (uint32_t)(4 * q * YS * 8);
#pragma unroll
                for (int ft = 0; ft < FT2; ++ft) {
                    // operand reads stay next to their use (hoisted together they spill)
                    if ((ft & 1) == 0) asm volatile("" ::: "memory");
                    const uint32_t pk = lds_read_u32(ftab_addr + 256u * (KS1 + ft));
                    const double b = lds_read(yq + (pk & 0xffffu)) * lds_read(yq + (pk >> 16));
#pragma unroll
                    for (int it = 0; it < KTL; ++it)
                        acc2[it][ft] = mfma_f64(a[it], b, acc2[it][ft]);
                }
            } else if constexpr (M4) {
                // the SHORT matrix instruction (v_mfma_f64_4x4x4, four blocks; section 4.7 of
                // 62-71 flop/ns against 44-46 of the 16x16x4 form).  The B operand of
                // the long form IS the B operand of the short one -- lane (l15, g) holds feature
                // 16 ft + l15 of point 4q + g, i.e. B[b][k][j] with 4b + j = l15, k = g --, the
                // A operand is r[4Q + i][4q + k] for ALL four blocks (lane i + 4b + 16k: a read
                // that does not depend on b), and D[b][i][j] lands in lane (l15 = 4b + j, g = i)
                // = T[4Q + g][16 ft + l15]: component (Q & 3) of the long form's accumulator
                // tile Q >> 2.  Four times the matrix instructions, each less than a quarter of
                // the time; the feature operands stay in registers over the 4 KTL row blocks
                double bf[FT2];
#pragma unroll
                for (int ft = 0; ft < FT2; ++ft) bf[ft] = ftile[nn * FS + ft * 16 + l15];
#pragma unroll
                for (int Q = 0; Q < 4 * KTL; ++Q) {
                    if ((Q & 3) == 0) asm volatile("" ::: "memory");
                    const double a = rtile[(4 * Q + (l & 3)) * RS + nn];
#pragma unroll
                    for (int ft = 0; ft < FT2; ++ft)
                        acc2[Q >> 2][ft][Q & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                            a, bf[ft], acc2[Q >> 2][ft][Q & 3], 0, 0, 0);
                }
            } else {
                double bf[FT2];
#pragma unroll
                for (int ft = 0; ft < FT2; ++ft) bf[ft] = ftile[nn * FS + ft * 16 + l15];
#pragma unroll
                for (int it = 0; it < KTL; ++it) {
                    const double a = rtile[(it * 16 + l15) * RS + nn];
#pragma unroll
                    for (int ft = 0; ft < FT2; ++ft)
                        acc2[it][ft] = mfma_f64(a, bf[ft], acc2[it][ft]);
                }
            }
        }
        lds_fence();
    }

    // ---- per-WORKGROUP partials: [KP][F2P] + 2 scalars (waves combined in fixed order) ----
    __syncthreads();                       // every wave is done with the LDS tiles / fragments
    double *scr = lds;                     // KP*F2P + 2 doubles
    // the four lane groups of a column (and both wavefronts of a pair) hold identical (mx, s):
    // count group 0 (of half 0) only
    double s_lse = (g == 0 && kh == 0) ? s_mx + s_log + log(prod) : 0.0;
    s_lse = wave_sum(s_lse);
    for (int ww = 0; ww < WAVES; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int it = 0; it < KTL; ++it)
#pragma unroll
                for (int ft = 0; ft < FT2; ++ft)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = ((kh * KTL + it) * 16 + g + 4 * r) * F2P + ft * 16 + l15;
                        // the first KS wavefronts start the rows of their clusters
                        scr[idx] = (ww < KS ? 0.0 : scr[idx]) + acc2[it][ft][r];
                    }
            if (l == 0) {
                scr[KP * F2P + 0] = (ww == 0 ? 0.0 : scr[KP * F2P + 0]) + s_lse;
                scr[KP * F2P + 1] = 0.0;
            }
        }
        __syncthreads();
    }
    const int64_t plen = (int64_t)KP * F2P + 8;
    double *Pb = P + (int64_t)blockIdx.x * plen;
    for (int e2 = tid; e2 < KP * F2P + 2; e2 += NTP) Pb[e2] = scr[e2];
}

// T (natural layout) <- sum over wave partials (fixed order), compact -> natural.
__global__ void __launch_bounds__(NT)
gmm_reduce_kernel(vmp_gmm_layout L, int D, int K, const double *__restrict__ P, int nb,
                  int update_zs, double *__restrict__ Tc, double *__restrict__ st)
{
    __shared__ double part[4][64];
    const int KP = (int)L.KP, F2P = (int)L.F2P;
    const int64_t plen = (int64_t)KP * F2P + 8;
    const int total = KP * F2P + 2;
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + o;
    // slice sl sums partials [b0, b1) in order; slices combined in order 0..3
    const int per = (nb + 3) / 4;
    const int b0 = sl * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    double s0 = 0.0, s1 = 0.0;
    if (e < total) {
        int b = b0;
        for (; b + 1 < b1; b += 2) {
            s0 += P[(int64_t)b * plen + e];
            s1 += P[(int64_t)(b + 1) * plen + e];
        }
        if (b < b1) s0 += P[(int64_t)b * plen + e];
    }
    part[sl][o] = s0 + s1;
    __syncthreads();
    if (sl != 0 || e >= total) return;
    const double v = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    if (e >= KP * F2P) {
        if (update_zs) st[L.off_zs + (e - KP * F2P)] = v;
        return;
    }
    const int k = e / F2P, f = e - k * F2P;
    Tc[e] = (k < K) ? v : 0.0;               // compact order, for <C, T> (gmm_rphi_kernel)
    if (k >= K) return;
    double *T = st + L.off_T + (int64_t)k * L.FS;
    const int npair = D * (D + 1) / 2;
    if (f < npair) {
        int rem = f, a = 0;
        while (rem >= D - a) { rem -= D - a; ++a; }
        const int bb = a + rem;
        T[1 + D + a * D + bb] = v;
        T[1 + D + bb * D + a] = v;
    } else if (f < npair + D) {
        T[1 + (f - npair)] = v;
    } else if (f == npair + D) {
        T[0] = v;
    }
}

// sum_nk r_nk phi_nk = sum_kf C_kf T_kf (Phi = C feat and T = r feat^T are the same contraction
// summed in the other order); 0 * -inf (impossible clusters) counts as 0 like the p != 0 guard of
// the reference (expfamily.py:455-460 via the masked sum).
__global__ void __launch_bounds__(NT)
gmm_rphi_kernel(vmp_gmm_layout L, const double *__restrict__ Tc, double *__restrict__ st)
{
    __shared__ double red[NT / 64];
    const int total = (int)(L.KP * L.F2P);
    const double *C = st + L.off_C;
    double acc = 0.0;
    for (int e = threadIdx.x; e < total; e += NT) {
        const double t = Tc[e];
        if (t != 0.0) acc += C[e] * t;
    }
    acc = block_sum<NT>(acc, red);
    if (threadIdx.x == 0) st[L.off_zs + 1] = acc;
}

// ---------------------------------------------------------------------------
// Per-cluster small kernels: one matrix element per lane, SPD inverse by Gauss-Jordan sweeps.
// WPC = wavefronts per cluster: 1 for D*D <= 64 (four clusters per workgroup, wavefront-level
// ordering only), 4 for D*D <= 256 and 16 for D*D <= 1024 (one cluster per workgroup, workgroup
// barriers).
// ---------------------------------------------------------------------------
template <int WPC>
__device__ __forceinline__ void grp_sync()
{
    if constexpr (WPC == 1) lds_fence();
    else __syncthreads();
}

// in: v = element (i,j) of an SPD matrix (lanes >= D*D idle); out: element of the inverse
template <int WPC>
__device__ inline double grp_spd_inverse(double v, int D, int i, int j, bool act, int l, double *M,
                                         double *logdet, int *bad)
{
    if constexpr (WPC == 1) return wave_spd_inverse(v, D, i, j, act, M, logdet, bad);
    double ld = 0.0, prod = 1.0;
    for (int p = 0; p < D; ++p) {
        M[l] = v;
        __syncthreads();
        const double piv = M[p * D + p];
        const double ci = act ? M[i * D + p] : 0.0, rj = act ? M[p * D + j] : 0.0;
        if (!(piv > 0.0)) *bad = 1;
        logdet_accumulate(piv, prod, ld);
        const double d = fast_recip(piv);
        if (i == p) v = (j == p) ? d : rj * d;
        else if (j == p) v = -ci * d;
        else v = v - ci * rj * d;
        __syncthreads();
    }
    *logdet = logdet_finish(prod, ld);
    return v;
}

// threads per workgroup of the per-cluster kernels: 256 (four clusters of <= 64 elements or one of
// <= 256), 1024 for one cluster of <= 1024 elements
constexpr int grp_threads(int WPC) { return WPC <= 4 ? NT : 64 * WPC; }

template <int WPC>
__global__ void __launch_bounds__(grp_threads(WPC))
gmm_init_state_kernel(vmp_gmm_layout L, int D, int K, double beta0, double n0, double *st)
{
    // priors are already stored in st[off_prior..]; initialise every node from its prior
    // (ExponentialFamily.initialize_from_prior, expfamily.py:168-184)
    constexpr int TPC = 64 * WPC, CPB = grp_threads(WPC) / TPC;
    __shared__ double Ms[CPB][TPC];
    const int w = threadIdx.x / TPC, l = threadIdx.x % TPC;
    const int k = blockIdx.x * CPB + w;
    const bool act = (k < K) && (l < D * D);
    const int i = act ? l / D : 0, j = act ? l - i * D : 0;
    const double *pr = st + L.off_prior;
    const double *V0 = pr + L.KP + 8;
    // alpha: Dirichlet prior moments
    if (k < K && l == 0) {
        double sa = 0.0;
        for (int c = 0; c < K; ++c) sa += pr[c];
        st[L.off_alpha + k] = pr[k];
        st[L.off_alpha + L.KP + k] = vmp_digamma(pr[k]) - vmp_digamma(sa);
        st[L.off_nk + k] = n0;
        st[L.off_logdetLmu + k] = (double)D * log(beta0);
    }
    if (act) {
        st[L.off_Cmu + (int64_t)k * D * D + l] = (i == j) ? 1.0 / beta0 : 0.0;
        st[L.off_Vk + (int64_t)k * D * D + l] = V0[l];
    }
    if (act && j == 0) st[L.off_mu + (int64_t)k * D + i] = 0.0;
    double ld;
    int bad = 0;
    const double vinv = grp_spd_inverse<WPC>(act ? V0[l] : 0.0, D, i, j, act, l, Ms[w], &ld, &bad);
    if (act) st[L.off_Lam + (int64_t)k * D * D + l] = n0 * vinv;
    if (k < K && l == 0) {
        double md = 0.0;
        for (int c = 0; c < D; ++c) md += vmp_digamma(0.5 * n0 - 0.5 * c);
        st[L.off_logdetV + k] = ld;
        st[L.off_logdetLam + k] = md + (double)D * log(2.0) - ld;
        if (bad) st[L.off_scal + 3] = (double)VMP_ERR_NOT_POSDEF;
    }
    if (k == 0 && l == 0) st[L.off_prior + L.KP + 2] = ld;   // log|V0|
}

// mu.update(): Lambda_mu = beta0 I + R_k <Lambda_k>, Cov, mean = Cov <Lambda_k> S1_k
// (gaussian.py:649-706 with the messages gaussian.py:2451-2454 weighted by r, mixture.py:126-158)
template <int WPC>
__global__ void __launch_bounds__(grp_threads(WPC))
gmm_update_mu_kernel(vmp_gmm_layout L, int D, int K, double *st)
{
    constexpr int TPC = 64 * WPC, CPB = grp_threads(WPC) / TPC;
    __shared__ double Ms[CPB][TPC];
    __shared__ double Cs[CPB][TPC];
    __shared__ double bs[CPB][MAXD];
    const int w = threadIdx.x / TPC, l = threadIdx.x % TPC;
    const int k = blockIdx.x * CPB + w;
    const bool act = (k < K) && (l < D * D);
    const int i = act ? l / D : 0, j = act ? l - i * D : 0;
    const double beta0 = st[L.off_prior + L.KP + 0];
    const int kk = k < K ? k : 0;
    const double *T = st + L.off_T + (int64_t)kk * L.FS;
    const double *Lam = st + L.off_Lam + (int64_t)kk * D * D;
    const double R = T[0];
    double v = act ? R * Lam[l] + ((i == j) ? beta0 : 0.0) : 0.0;
    double ld;
    int bad = 0;
    v = grp_spd_inverse<WPC>(v, D, i, j, act, l, Ms[w], &ld, &bad);
    Cs[w][l] = v;
    if (l < D) {
        double s = 0.0;
        for (int c = 0; c < D; ++c) s += Lam[l * D + c] * T[1 + c];     // <Lambda> S1
        bs[w][l] = s;
    }
    grp_sync<WPC>();
    if (act) st[L.off_Cmu + (int64_t)k * D * D + l] = v;
    if (k < K && l < D) {
        double s = 0.0;
        for (int c = 0; c < D; ++c) s += Cs[w][l * D + c] * bs[w][c];
        st[L.off_mu + (int64_t)k * D + l] = s;
    }
    if (k < K && l == 0) {
        st[L.off_logdetLmu + k] = ld;
        if (bad) st[L.off_scal + 3] = (double)VMP_ERR_NOT_POSDEF;
    }
}

// Lambda.update(): n_k = n0 + R_k, V_k = V0 + S2 - S1 mu^T - mu S1^T + R <mu mu^T>
// (wishart.py:153-188 with the message gaussian.py:2516-2520 weighted by r)
template <int WPC>
__global__ void __launch_bounds__(grp_threads(WPC))
gmm_update_lambda_kernel(vmp_gmm_layout L, int D, int K, double *st)
{
    constexpr int TPC = 64 * WPC, CPB = grp_threads(WPC) / TPC;
    __shared__ double Ms[CPB][TPC];
    const int w = threadIdx.x / TPC, l = threadIdx.x % TPC;
    const int k = blockIdx.x * CPB + w;
    const bool act = (k < K) && (l < D * D);
    const int i = act ? l / D : 0, j = act ? l - i * D : 0;
    const int kk = k < K ? k : 0;
    const double n0 = st[L.off_prior + L.KP + 1];
    const double *V0 = st + L.off_prior + L.KP + 8;
    const double *T = st + L.off_T + (int64_t)kk * L.FS;
    const double *mu = st + L.off_mu + (int64_t)kk * D;
    const double *Cmu = st + L.off_Cmu + (int64_t)kk * D * D;
    const double R = T[0];
    const double nk = n0 + R;
    double v = 0.0;
    if (act) {
        const double mm = Cmu[l] + mu[i] * mu[j];
        v = V0[l] + T[1 + D + l] - T[1 + i] * mu[j] - mu[i] * T[1 + j] + R * mm;
        st[L.off_Vk + (int64_t)k * D * D + l] = v;
    }
    // symmetrise before factorising
    Ms[w][l] = v;
    grp_sync<WPC>();
    if (act) v = 0.5 * (Ms[w][i * D + j] + Ms[w][j * D + i]);
    grp_sync<WPC>();
    double ld;
    int bad = 0;
    v = grp_spd_inverse<WPC>(v, D, i, j, act, l, Ms[w], &ld, &bad);
    if (act) st[L.off_Lam + (int64_t)k * D * D + l] = nk * v;               // wishart.py:184
    if (k < K && l == 0) {
        double md = 0.0;
        for (int c = 0; c < D; ++c) md += vmp_digamma(0.5 * nk - 0.5 * c);  // utils/misc.py:1146
        st[L.off_nk + k] = nk;
        st[L.off_logdetV + k] = ld;
        st[L.off_logdetLam + k] = md + (double)D * log(2.0) - ld;           // wishart.py:185
        if (bad) st[L.off_scal + 3] = (double)VMP_ERR_NOT_POSDEF;
    }
}

// c_k of E[log N(y | mu_k, Lambda_k)] without the data terms
__device__ inline double gmm_ck(const double *st, const vmp_gmm_layout &L, int D, int k)
{
    const double *Lam = st + L.off_Lam + (int64_t)k * D * D;
    const double *mu = st + L.off_mu + (int64_t)k * D;
    const double *Cmu = st + L.off_Cmu + (int64_t)k * D * D;
    double tr = 0.0;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) tr += Lam[i * D + j] * (Cmu[i * D + j] + mu[i] * mu[j]);
    return 0.5 * st[L.off_logdetLam + k] - 0.5 * (double)D * log(2.0 * M_PI) - 0.5 * tr;
}

// z.update(), replicated half: the coefficient matrix C of the pass
// (phi of the Categorical = <log pi> + E[log p(y | k)], mixture.py:67-104)
__global__ void __launch_bounds__(NT)
gmm_prepare_z_kernel(vmp_gmm_layout L, int D, int K, int prior_only, double *st)
{
    // Row k of C holds the coefficients of  ell_nk = c_k + b_k . y - 1/2 y^T Lam_k y  in the
    // compact feature order of the pass kernel: pairs y_a y_b (a <= b), then y_d, then 1.
    const int F2P = (int)L.F2P, KP = (int)L.KP;
    const int npair = D * (D + 1) / 2;
    for (int e = blockIdx.x * NT + threadIdx.x; e < KP * F2P; e += gridDim.x * NT) {
        const int k = e / F2P, f = e - k * F2P;
        double v = 0.0;
        if (k < K && prior_only) {
            // q(z) from its prior: phi = <log pi> only (initialize_from_prior)
            if (f == npair + D) v = st[L.off_alpha + KP + k];
        } else if (k < K) {
            const double *Lam = st + L.off_Lam + (int64_t)k * D * D;
            if (f < npair) {
                int rem = f, a = 0;
                while (rem >= D - a) { rem -= D - a; ++a; }
                const int b = a + rem;
                v = (a == b) ? -0.5 * Lam[a * D + a] : -0.5 * (Lam[a * D + b] + Lam[b * D + a]);
            } else if (f < npair + D) {
                const int d = f - npair;
                const double *mu = st + L.off_mu + (int64_t)k * D;
                double s = 0.0;
                for (int c = 0; c < D; ++c) s += Lam[d * D + c] * mu[c];
                v = s;
            } else if (f == npair + D) {
                v = st[L.off_alpha + KP + k] + gmm_ck(st, L, D, k);
            }
        } else if (f == npair + D) {
            v = -INFINITY;                    // padded clusters get zero responsibility
        }
        st[L.off_C + e] = v;
    }
}

__global__ void __launch_bounds__(NT)
gmm_update_alpha_kernel(vmp_gmm_layout L, int D, int K, double *st)
{
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int k = tid; k < K; k += NT) {
        const double a = st[L.off_prior + k] + st[L.off_T + (int64_t)k * L.FS];   // alpha0 + R_k
        st[L.off_alpha + k] = a;
        s += a;
    }
    s = block_sum<NT>(s, red);
    const double ps = vmp_digamma(s);
    for (int k = tid; k < K; k += NT)
        st[L.off_alpha + L.KP + k] = vmp_digamma(st[L.off_alpha + k]) - ps;      // dirichlet.py:150-152
}

// out-of-line copies keep the register footprint of the bookkeeping kernels small
__device__ __noinline__ double lgamma_ni(double x) { return vmp_lgamma(x); }

__device__ __noinline__ double multigammaln_dev(double a, int d)
{
    double s = (double)d * (d - 1) / 4.0 * log(M_PI);
    for (int i = 0; i < d; ++i) s += lgamma_ni(a - 0.5 * i);
    return s;
}

// expfamily.py:400-480 for Y, z, alpha, mu, Lambda.  One workgroup of 16
// wavefronts; a wavefront owns clusters k = w, w+8, ... with the (i,j) matrix elements dealt to its lanes.
constexpr int NTLB = 512;
__global__ void __launch_bounds__(NTLB)
gmm_lower_bound_kernel(vmp_gmm_layout L, int D, int K, double *st)
{
    __shared__ double red[NTLB / 64];
    __shared__ double lgs[MAXK][2 + 2 * MAXD];     // all log-Gamma values, ONE call site
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int KP = (int)L.KP;
    const double beta0 = st[L.off_prior + KP + 0], n0 = st[L.off_prior + KP + 1];
    const double ldV0 = st[L.off_prior + KP + 2];
    const double *V0 = st + L.off_prior + KP + 8;
    const int per = 2 + 2 * D;
    for (int e = tid; e < K * per; e += NTLB) {
        const int k = e / per, c = e - k * per;
        double x;
        if (c == 0) x = st[L.off_prior + k];                          // alpha0_k
        else if (c == 1) x = st[L.off_alpha + k];                     // alpha_k
        else if (c < 2 + D) x = 0.5 * n0 - 0.5 * (c - 2);             // Gamma_D(n0/2) terms
        else x = 0.5 * st[L.off_nk + k] - 0.5 * (c - 2 - D);          // Gamma_D(n_k/2) terms
        lgs[k][c] = vmp_lgamma(x);
    }
    __syncthreads();
    const double mgc = (double)D * (D - 1) / 4.0 * log(M_PI);
    double LY = 0.0, Lz = 0.0, La = 0.0, Lmu = 0.0, LL = 0.0, sa0 = 0.0, sa = 0.0;
    for (int k = w; k < K; k += NTLB / 64) {
        const double *T = st + L.off_T + (int64_t)k * L.FS;
        const double *Lam = st + L.off_Lam + (int64_t)k * D * D;
        const double *mu = st + L.off_mu + (int64_t)k * D;
        const double *Cmu = st + L.off_Cmu + (int64_t)k * D * D;
        const double *Vk = st + L.off_Vk + (int64_t)k * D * D;
        double bs = 0.0, ls2 = 0.0, trLmm = 0.0, trmm = 0.0, trV0 = 0.0, trVk = 0.0;
        // one (i,j) element per lane and round (a single round while D*D <= 64)
        for (int e = l; e < D * D; e += 64) {
            const int i = e / D, j = e - i * D;
            const double lam = Lam[e];
            bs += lam * mu[j] * T[1 + i];                      // (Lambda mu) . S1
            ls2 += lam * T[1 + D + e];                         // tr(Lambda S2)
            trLmm += lam * (Cmu[e] + mu[i] * mu[j]);           // tr(Lambda <mu mu^T>)
            trV0 += V0[e] * lam;
            trVk += 0.5 * (Vk[e] + Vk[j * D + i]) * lam;
            if (i == j) trmm += Cmu[e] + mu[i] * mu[i];
        }
        bs = wave_sum(bs); ls2 = wave_sum(ls2); trLmm = wave_sum(trLmm);
        trmm = wave_sum(trmm); trV0 = wave_sum(trV0); trVk = wave_sum(trVk);
        if (l == 0) {
            const double R = T[0];
            const double ldL = st[L.off_logdetLam + k];
            const double ck = 0.5 * ldL - 0.5 * (double)D * log(2.0 * M_PI) - 0.5 * trLmm;
            LY += R * ck + bs - 0.5 * ls2;
            const double a0 = st[L.off_prior + k], a = st[L.off_alpha + k];
            const double lp = st[L.off_alpha + KP + k];
            Lz += R * lp;
            La += -lgs[k][0] + lgs[k][1] + (a0 - a) * lp;
            sa0 += a0;
            sa += a;
            Lmu += -0.5 * beta0 * trmm + 0.5 * (double)D * log(beta0)
                   - 0.5 * st[L.off_logdetLmu + k] + 0.5 * (double)D;
            const double nk = st[L.off_nk + k];
            double mg0 = mgc, mgk = mgc;
            for (int c = 0; c < D; ++c) { mg0 += lgs[k][2 + c]; mgk += lgs[k][2 + D + c]; }
            const double gp = 0.5 * n0 * ldV0 - 0.5 * D * n0 * log(2.0) - mg0;
            const double gq = 0.5 * nk * st[L.off_logdetV + k] - 0.5 * D * nk * log(2.0) - mgk;
            LL += gp - gq - 0.5 * trV0 + 0.5 * n0 * ldL + 0.5 * trVk - 0.5 * nk * ldL;
        }
    }
    LY = block_sum<NTLB>(LY, red);
    Lz = block_sum<NTLB>(Lz, red);
    La = block_sum<NTLB>(La, red);
    Lmu = block_sum<NTLB>(Lmu, red);
    LL = block_sum<NTLB>(LL, red);
    sa0 = block_sum<NTLB>(sa0, red);
    sa = block_sum<NTLB>(sa, red);
    if (tid < 2) lgs[0][tid] = vmp_lgamma(tid == 0 ? sa0 : sa);
    __syncthreads();
    if (tid == 0) {
        Lz += st[L.off_zs + 0] - st[L.off_zs + 1];
        La += lgs[0][0] - lgs[0][1];
        st[L.off_L + 0] = LY;
        st[L.off_L + 1] = Lz;
        st[L.off_L + 2] = La;
        st[L.off_L + 3] = Lmu;
        st[L.off_L + 4] = LL;
        st[L.off_L + 5] = LY + Lz + La + Lmu + LL;
    }
}

int gmm_wgs_per_cu()
{
    static int v = -1;
    if (v < 0) {
        // one 8-wavefront workgroup per CU = two wavefronts per SIMD (the register budget of
        // the pass); the LDS tiles of the K = 64 instance (148 KB) admit no second one
        const char *e = getenv("VMP_GMM_WGS_PER_CU");
        v = e ? atoi(e) : 1;
        if (v < 1) v = 1;
        if (v > 4) v = 4;
    }
    return v;
}

int64_t gmm_max_grid(vmp_ctx *ctx) { return (int64_t)ctx->num_cu * gmm_wgs_per_cu(); }

// wavefronts per workgroup of an instance: eight (two per SIMD, 256 registers) while the
// KT x FT2 accumulator tiles of phase 2 fit in 96 registers, else four (one per SIMD, 512)
constexpr int gmm_waves(int KTL, int FT2) { return KTL * FT2 <= 20 ? 8 : 4; }

template <int DPT, int KT, int FT2, bool FROM_LABELS>
int32_t launch_gmm_pass_as(vmp_ctx *ctx, const double *Y, int64_t N, int D, int K,
                           const double *C, const int64_t *labels, double *R, double *P,
                           int64_t ntiles)
{
    constexpr int DP = 4 * DPT, KP = 16 * KT;
    constexpr int KS1 = 4 * FT2;                  // k-steps of phase 1 (compact features)
    constexpr int F2P = 16 * FT2;
    constexpr bool REGEN = DPT > 2;
    constexpr int KS = (REGEN && KT == 4) ? 2 : 1;             // cluster split over a wavefront pair
    constexpr int KTL = KT / KS;
    constexpr int NW = gmm_waves(KTL, FT2);
    constexpr size_t ytile = (size_t)TNC * (DP + 3), rtile = (size_t)KTL * 16 * RS;
    constexpr size_t per_wave = REGEN ? ytile + rtile
                                      : (ytile > rtile ? ytile : rtile) + (size_t)TNC * (F2P + 2);
    size_t ldsd = (FROM_LABELS ? 0 : (size_t)KT * KS1 * 64 + 256 + (KS == 2 ? 2 * NW * 16 : 0))
                  + (REGEN ? (size_t)(KS1 + FT2) * 32 : 0) + NW * per_wave;
    if (ldsd < (size_t)KP * F2P + 8) ldsd = (size_t)KP * F2P + 8;   // partials of the workgroup
    const size_t lds = ldsd * sizeof(double);
    int64_t g = (ntiles + NW / KS - 1) / (NW / KS);
    if (g > gmm_max_grid(ctx)) g = gmm_max_grid(ctx);
    if (g < 1) g = 1;
    auto kern = gmm_pass_kernel<DPT, KT, FT2, FROM_LABELS, NW, REGEN, KS>;
    if constexpr (!REGEN) {
        // phase 2 on the short matrix instruction (the staged-feature instances: D <= 8): an
        // experiment kept behind the tune key -- measured 2.84-2.86 against 2.81 ms at config 3
        // (tools/gmm_lab.py, profiles/r05/pmc_gmm_mfma4.txt)
        if (vmp_tune_get("gmm_mfma4", 0) != 0)
            kern = gmm_pass_kernel<DPT, KT, FT2, FROM_LABELS, NW, REGEN, KS, true>;
    }
    static const void *attr_done[2] = {nullptr, nullptr};
    if (attr_done[0] != (const void *)kern && attr_done[1] != (const void *)kern) {
        VMP_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)kern,
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024));
        attr_done[attr_done[0] ? 1 : 0] = (const void *)kern;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(64 * NW), lds, ctx->stream, Y, N, D, K, C,
                       labels, R, P, ntiles);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return (int32_t)g + 1000;                     // > 0: number of workgroup partials + 1000
}

template <int DPT, int KT, int FT2>
int32_t launch_gmm_pass(vmp_ctx *ctx, bool from_labels, const double *Y, int64_t N,
                        int D, int K, const double *C, const int64_t *labels, double *R,
                        double *P, int64_t ntiles)
{
    return from_labels
        ? launch_gmm_pass_as<DPT, KT, FT2, true>(ctx, Y, N, D, K, C, labels, R, P, ntiles)
        : launch_gmm_pass_as<DPT, KT, FT2, false>(ctx, Y, N, D, K, C, labels, R, P, ntiles);
}

int32_t run_gmm_pass(vmp_ctx *ctx, bool from_labels, const double *Y, int64_t N, int D, int K,
                     const int64_t *labels, double *R, double *state, void *workspace)
{
    VMP_REQUIRE(ctx, ctx && Y && R && state && workspace, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, !from_labels || labels, VMP_ERR_INVALID, "null labels");
    VMP_REQUIRE(ctx, D >= 1 && K >= 1 && N >= 0, VMP_ERR_INVALID, "bad dims");
    VMP_REQUIRE(ctx, D <= MAXD && K <= MAXK, VMP_ERR_UNSUPPORTED,
                "fused GMM block supports D <= %d, K <= %d (got D=%d, K=%d)", MAXD, MAXK, D, K);
    vmp_gmm_layout L;
    fill_layout(D, K, &L);
    const int DPT = (int)(L.DP / 4), KT = (int)(L.KP / 16), FT2 = (int)(L.F2P / 16);
    const int64_t ntiles = (N + TNC - 1) / TNC;
    double *P = reinterpret_cast<double *>(workspace);
    const double *C = state + L.off_C;
    int32_t rc = VMP_ERR_UNSUPPORTED;
    hipEvent_t *ev = ctx->timing ? vmp_next_events(ctx) : nullptr;
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[0], ctx->stream));
    if (D > 16) {
        // workspace: [partials | compact T | coefficient fragments]
        int64_t pd = 0, fd = 0;
        vmp_gmm_wide_workspace_doubles(ctx, D, K, L.F2P, L.KP, &pd, &fd);
        double *Tcw = P + pd, *Cfrag = Tcw + L.KP * L.F2P;
        int nb = 0;
        rc = vmp_gmm_wide_pass(ctx, Y, N, D, K, L.F2P, L.KP, C, from_labels ? labels : nullptr, R,
                               P, Cfrag, &nb);
        if (rc != VMP_OK) return rc;
        if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[1], ctx->stream));
        const int totalw = (int)(L.KP * L.F2P + 2);
        hipLaunchKernelGGL(gmm_reduce_kernel, dim3((totalw + 63) / 64), dim3(NT), 0, ctx->stream,
                           L, D, K, P, nb, from_labels ? 0 : 1, Tcw, state);
        if (!from_labels)
            hipLaunchKernelGGL(gmm_rphi_kernel, dim3(1), dim3(NT), 0, ctx->stream, L, Tcw, state);
        VMP_HIP_CHECK(ctx, hipGetLastError());
        if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[2], ctx->stream));
        return VMP_OK;
    }
#define VMP_GCASE(dpt, kt, ft2)                                                                 \
    if (DPT == dpt && KT == kt && FT2 == ft2)                                                   \
        rc = launch_gmm_pass<dpt, kt, ft2>(ctx, from_labels, Y, N, D, K, C, labels, R, P, ntiles);
#define VMP_GCASE_K(dpt, ft2) VMP_GCASE(dpt, 1, ft2) VMP_GCASE(dpt, 2, ft2) VMP_GCASE(dpt, 4, ft2)
    VMP_GCASE_K(1, 1)
    VMP_GCASE_K(2, 1) VMP_GCASE_K(2, 2) VMP_GCASE_K(2, 3)
    // 9 <= D <= 16: DP = 16, F2P = 64 ... 160
    VMP_GCASE_K(4, 4) VMP_GCASE_K(4, 5) VMP_GCASE_K(4, 6) VMP_GCASE_K(4, 7)
    VMP_GCASE_K(4, 8) VMP_GCASE_K(4, 9) VMP_GCASE_K(4, 10)
#undef VMP_GCASE_K
#undef VMP_GCASE
    if (rc < 1000) {
        if (rc == VMP_ERR_UNSUPPORTED)
            VMP_SET_ERR(ctx, "no GMM kernel instance for DPT=%d KT=%d FT2=%d", DPT, KT, FT2);
        return rc;
    }
    const int g = rc - 1000;
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[1], ctx->stream));
    const int total = (int)(L.KP * L.F2P + 2);
    double *Tc = P + gmm_max_grid(ctx) * (L.KP * L.F2P + 8);
    hipLaunchKernelGGL(gmm_reduce_kernel, dim3((total + 63) / 64), dim3(NT), 0, ctx->stream,
                       L, D, K, P, g, from_labels ? 0 : 1, Tc, state);
    if (!from_labels)
        hipLaunchKernelGGL(gmm_rphi_kernel, dim3(1), dim3(NT), 0, ctx->stream, L, Tc, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[2], ctx->stream));
    return VMP_OK;
}

}  // namespace

extern "C" {

int32_t vmp_gmm_get_layout(int32_t D, int32_t K, vmp_gmm_layout *out)
{
    if (!out || D < 1 || K < 1) return VMP_ERR_INVALID;
    if (D > MAXD || K > MAXK) return VMP_ERR_UNSUPPORTED;
    fill_layout(D, K, out);
    return VMP_OK;
}

int32_t vmp_gmm_workspace_bytes(vmp_ctx *ctx, int32_t D, int32_t K, size_t *bytes)
{
    VMP_REQUIRE(ctx, ctx && bytes, VMP_ERR_INVALID, "null argument");
    vmp_gmm_layout L;
    int32_t rc = vmp_gmm_get_layout(D, K, &L);
    VMP_REQUIRE(ctx, rc == VMP_OK, rc, "fused GMM block supports D <= %d, K <= %d", MAXD, MAXK);
    *bytes = (size_t)(gmm_max_grid(ctx) * (L.KP * L.F2P + 8) + L.KP * L.F2P + 64) * sizeof(double);
    if (D > 16) {
        int64_t pd = 0, fd = 0;
        vmp_gmm_wide_workspace_doubles(ctx, D, K, L.F2P, L.KP, &pd, &fd);
        *bytes = (size_t)(pd + L.KP * L.F2P + fd + 64) * sizeof(double);
    }
    return VMP_OK;
}

int32_t vmp_gmm_init_state(vmp_ctx *ctx, int32_t D, int32_t K, const double *alpha0_host,
                           double beta0, double n0, const double *V0_host, double *state)
{
    VMP_REQUIRE(ctx, ctx && alpha0_host && V0_host && state, VMP_ERR_INVALID, "null argument");
    vmp_gmm_layout L;
    int32_t rc = vmp_gmm_get_layout(D, K, &L);
    VMP_REQUIRE(ctx, rc == VMP_OK, rc, "fused GMM block supports D <= %d, K <= %d", MAXD, MAXK);
    VMP_REQUIRE(ctx, beta0 > 0 && n0 > D - 1, VMP_ERR_INVALID, "bad prior parameters");
    for (int k = 0; k < K; ++k)
        VMP_REQUIRE(ctx, alpha0_host[k] > 0, VMP_ERR_NOT_POSITIVE,
                    "Natural parameters should be positive");
    hipStream_t s = ctx->stream;
    VMP_HIP_CHECK(ctx, hipMemsetAsync(state, 0, (size_t)L.total * sizeof(double), s));
    double hdr[8] = {beta0, n0, 0, 0, 0, 0, 0, 0};
    VMP_HIP_CHECK(ctx, hipMemcpyAsync(state + L.off_prior, alpha0_host, K * sizeof(double),
                                      hipMemcpyHostToDevice, s));
    VMP_HIP_CHECK(ctx, hipMemcpyAsync(state + L.off_prior + L.KP, hdr, sizeof(hdr),
                                      hipMemcpyHostToDevice, s));
    VMP_HIP_CHECK(ctx, hipMemcpyAsync(state + L.off_prior + L.KP + 8, V0_host,
                                      (size_t)D * D * sizeof(double), hipMemcpyHostToDevice, s));
    VMP_HIP_CHECK(ctx, hipStreamSynchronize(s));   // host buffers may be temporaries
    if (D * D <= 64)
        hipLaunchKernelGGL(gmm_init_state_kernel<1>, dim3((K + 3) / 4), dim3(NT), 0, s, L, D, K,
                           beta0, n0, state);
    else if (D * D <= 256)
        hipLaunchKernelGGL(gmm_init_state_kernel<4>, dim3(K), dim3(NT), 0, s, L, D, K, beta0, n0,
                           state);
    else
        hipLaunchKernelGGL(gmm_init_state_kernel<16>, dim3(K), dim3(1024), 0, s, L, D, K, beta0,
                           n0, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_gmm_stats_from_labels(vmp_ctx *ctx, const double *Y, int64_t N, int32_t D, int32_t K,
                                  const int64_t *labels, double *R, double *state,
                                  void *workspace)
{
    return run_gmm_pass(ctx, true, Y, N, D, K, labels, R, state, workspace);
}

int32_t vmp_gmm_pass(vmp_ctx *ctx, const double *Y, int64_t N, int32_t D, int32_t K, double *R,
                     double *state, void *workspace)
{
    return run_gmm_pass(ctx, false, Y, N, D, K, nullptr, R, state, workspace);
}

#define VMP_GMM_PROLOGUE()                                                           \
    VMP_REQUIRE(ctx, ctx && state, VMP_ERR_INVALID, "null argument");                \
    vmp_gmm_layout L;                                                                \
    {                                                                                \
        int32_t rc__ = vmp_gmm_get_layout(D, K, &L);                                 \
        VMP_REQUIRE(ctx, rc__ == VMP_OK, rc__, "unsupported dims D=%d K=%d", D, K);  \
    }

int32_t vmp_gmm_update_mu(vmp_ctx *ctx, int32_t D, int32_t K, double *state)
{
    VMP_GMM_PROLOGUE();
    if (D * D <= 64)
        hipLaunchKernelGGL(gmm_update_mu_kernel<1>, dim3((K + 3) / 4), dim3(NT), 0, ctx->stream, L,
                           D, K, state);
    else if (D * D <= 256)
        hipLaunchKernelGGL(gmm_update_mu_kernel<4>, dim3(K), dim3(NT), 0, ctx->stream, L, D, K,
                           state);
    else
        hipLaunchKernelGGL(gmm_update_mu_kernel<16>, dim3(K), dim3(1024), 0, ctx->stream, L, D, K,
                           state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_gmm_update_lambda(vmp_ctx *ctx, int32_t D, int32_t K, double *state)
{
    VMP_GMM_PROLOGUE();
    if (D * D <= 64)
        hipLaunchKernelGGL(gmm_update_lambda_kernel<1>, dim3((K + 3) / 4), dim3(NT), 0,
                           ctx->stream, L, D, K, state);
    else if (D * D <= 256)
        hipLaunchKernelGGL(gmm_update_lambda_kernel<4>, dim3(K), dim3(NT), 0, ctx->stream, L, D, K,
                           state);
    else
        hipLaunchKernelGGL(gmm_update_lambda_kernel<16>, dim3(K), dim3(1024), 0, ctx->stream, L, D,
                           K, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_gmm_prepare_z(vmp_ctx *ctx, int32_t D, int32_t K, int32_t prior_only, double *state)
{
    VMP_GMM_PROLOGUE();
    const int n = (int)(L.KP * L.F2P);
    hipLaunchKernelGGL(gmm_prepare_z_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, ctx->stream, L,
                       D, K, (int)prior_only, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_gmm_update_alpha(vmp_ctx *ctx, int32_t D, int32_t K, double *state)
{
    VMP_GMM_PROLOGUE();
    hipLaunchKernelGGL(gmm_update_alpha_kernel, dim3(1), dim3(NT), 0, ctx->stream, L, D, K, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_gmm_lower_bound(vmp_ctx *ctx, int32_t D, int32_t K, double *state)
{
    VMP_GMM_PROLOGUE();
    hipLaunchKernelGGL(gmm_lower_bound_kernel, dim3(1), dim3(NTLB), 0, ctx->stream, L, D, K,
                       state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // extern "C"
