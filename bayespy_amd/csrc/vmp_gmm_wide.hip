// vmp_gmm_wide.hip -- the mixture pass for 16 < D <= 32 (K <= 64).
//
// Same pass as vmp_gmm.hip (phase 1 Phi = C feat2(y), softmax, r, phase 2 T += r feat2(y)^T over
// the compact feature list [y_a y_b (a <= b), y_d, 1]; mixture.py:53-293, expfamily.py:45-61),
// re-cut for F2 up to 561 features (F2P = 576):
//   * the coefficient fragments of ONE 16-cluster tile are 73.7 KB at D = 32 -- the four of
//     K = 64 do not fit the LDS.  They are laid out in fragment order once per pass
//     (gmm_cfrag_kernel, 295 KB: L2-resident) and phase 1 streams them: eight 512-byte loads per
//     wavefront in flight ahead of the eight matrix instructions that use them (sixteen: +5 %
//     at D = 32 but spills where two wavefronts per SIMD fit); the next tile's y values are
//     fetched during the current tile.  Phi is summed in FOUR interleaved accumulators (k-steps
//     q = 0, 1, 2, 3 mod 4), added at the end: one accumulator is a chain of 48 ... 144 dependent
//     matrix instructions, which issue at about half the rate of independent ones;
//   * one 16-cluster tile per wavefront (KS = KT wavefronts share a 16-point tile; the T
//     accumulators of a tile are FT2 x 8 = 288 registers at D = 32: one wavefront per SIMD); the
//     wavefronts of a tile exchange the softmax normalisers through LDS as in the pair-split form
//     of vmp_gmm.hip (maxima, then sums added in the fixed order 0 .. KS-1 by every wavefront);
//   * features are formed twice from the y tile (factor offsets in an LDS table);
//   * every wavefront writes its rows of the partial statistics itself (K x F2P doubles are
//     295 KB: no workgroup staging); a tile group = one partial block for gmm_reduce_kernel.
// FT2 is rounded up to a multiple of 4 (F2P to 64) to bound the number of instances; the pad
// features are zero.  Given labels (z.initialize_from_value) skip phase 1 at run time.
#include "vmp_gmm_dev.h"

namespace {

constexpr int WNW = 4;            // wavefronts per workgroup
constexpr int WDP = 32, WYS = WDP + 3;
constexpr int PF = 8;             // coefficient fragments in flight per wavefront

__global__ void __launch_bounds__(256)
gmm_cfrag_kernel(const double *__restrict__ Cmat, int KT, int KS1, int F2P,
                 double *__restrict__ Cfrag)
{
    // lane (g, l15) of fragment (it, q) holds C[it*16 + l15][4q + g]
    const int total = KT * KS1 * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int lane = e & 63, fq = e >> 6;
        const int it = fq / KS1, q = fq - it * KS1;
        Cfrag[e] = Cmat[(int64_t)(it * 16 + (lane & 15)) * F2P + 4 * q + (lane >> 4)];
    }
}

template <int KT, int FT2>
__global__ void __launch_bounds__(64 * WNW, FT2 <= 16 ? 2 : 1)
gmm_wide_kernel(const double *__restrict__ Y, int64_t N, int D, int K,
                const double *__restrict__ Cfrag, const int64_t *__restrict__ labels,
                double *__restrict__ Rout, double *__restrict__ P, int64_t ntiles)
{
    constexpr int KS = KT, GROUPS = WNW / KS;
    constexpr int KP = 16 * KT, F2P = 16 * FT2, KS1 = F2P / 4;
    static_assert(KS1 % PF == 0, "FT2 is a multiple of 4");
    constexpr int NTP = 64 * WNW;
    const bool from_labels = labels != nullptr;

    extern __shared__ double lds[];
    double *tab = lds;                                           // 256
    double *xch = tab + 256;                                     // [2][WNW][16]
    uint32_t *ftab = reinterpret_cast<uint32_t *>(xch + 2 * WNW * 16);   // [KS1 + FT2][64]
    double *wbase = xch + 2 * WNW * 16 + (KS1 + FT2) * 32;
    const int tid = threadIdx.x;
    const int l = tid & 63, l15 = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = w % KS, grp = w / KS;
    double *ytile = wbase + w * (TNC * WYS + 16 * RS);           // [16][WYS]
    double *rtile = ytile + TNC * WYS;                           // [16][RS]

    for (int e = tid; e < 256; e += NTP) tab[e] = VMP_EXP2_TAB[e];
    // factors of feature f: ytile[n][a] * ytile[n][b] (slot DP holds 1, DP + 1 holds 0)
    const int npair = D * (D + 1) / 2;
    for (int e = tid; e < (KS1 + FT2) * 64; e += NTP) {
        const int lane = e & 63, row = e >> 6;
        const int f = row < KS1 ? 4 * row + (lane >> 4) : (row - KS1) * 16 + (lane & 15);
        int a = WDP + 1, b = WDP + 1;                // zero feature
        if (f < npair) {
            int rem = f, aa = 0;
            while (rem >= D - aa) { rem -= D - aa; ++aa; }
            a = aa; b = aa + rem;
        } else if (f < npair + D) {
            a = f - npair; b = WDP;                  // linear: y_d * 1
        } else if (f == npair + D) {
            a = WDP; b = WDP;                        // constant
        }
        ftab[e] = 8u * (uint32_t)a | (8u * (uint32_t)b) << 16;
    }
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(lds_f64 *)tab;
    const uint32_t yrow_addr = (uint32_t)(uintptr_t)(lds_f64 *)(ytile + l15 * WYS);
    const uint32_t ygrp_addr = (uint32_t)(uintptr_t)(lds_f64 *)(ytile + g * WYS);
    const uint32_t ftab_addr = (uint32_t)(uintptr_t)(lds_u32 *)(ftab + l);
    __syncthreads();

    v4f64 acc2[FT2];
#pragma unroll
    for (int ft = 0; ft < FT2; ++ft) acc2[ft] = v4f64{0.0, 0.0, 0.0, 0.0};
    double s_mx = 0.0, s_log = 0.0, prod = 1.0;
    int since = 0;
    const double *Cw = Cfrag + (int64_t)kh * KS1 * 64 + l;       // this wavefront's cluster tile

    const int64_t stride = (int64_t)gridDim.x * GROUPS;
    const int64_t tile0 = (int64_t)blockIdx.x * GROUPS + grp;
    // the barriers of the exchange need the same trip count in every wavefront
    const int64_t tile_end = tile0 + (ntiles + stride - 1) / stride * stride;
    // y of the first tile; inside the loop the next tile's values are fetched a tile ahead
    double ynx[WDP / 4];
#pragma unroll
    for (int j = 0; j < WDP / 4; ++j) {
        const int d = 4 * j + g;
        const int64_t nf = tile0 * TNC + l15;
        ynx[j] = (nf < N && d < D) ? Y[nf * D + d] : 0.0;
    }
    for (int64_t tile = tile0; tile < tile_end; tile += stride) {
        const int64_t n0 = tile * TNC;
        const int64_t n = n0 + l15;
        const bool nok = n < N;
        // ---- y: lane (g, l15) owns y[n][4j + g] ------------------------------------
#pragma unroll
        for (int j = 0; j < WDP / 4; ++j) ytile[l15 * WYS + 4 * j + g] = ynx[j];
        {
            const int64_t nn2 = (tile + stride) * TNC + l15;
#pragma unroll
            for (int j = 0; j < WDP / 4; ++j) {
                const int d = 4 * j + g;
                ynx[j] = (nn2 < N && d < D) ? Y[nn2 * D + d] : 0.0;
            }
        }
        if (g == 0) {
            ytile[l15 * WYS + WDP] = 1.0;
            ytile[l15 * WYS + WDP + 1] = 0.0;
        }
        lds_fence();

        if (!from_labels) {
            // ---- phase 1: Phi(16 x 16) = C_tile * feat, coefficients streamed from L2 --------
            v4f64 acc1[1], accp[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) accp[c] = v4f64{0.0, 0.0, 0.0, 0.0};
            double cc[PF], cn[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) cc[u] = Cw[(int64_t)u * 64];
            // (a real loop: unrolled over the 48 ... 144 k-steps it costs ~100 registers)
#pragma unroll 1
            for (int q0 = 0; q0 < KS1; q0 += PF) {
                if (q0 + PF < KS1) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) cn[u] = Cw[(int64_t)(q0 + PF + u) * 64];
                }
                asm volatile("" ::: "memory");       // the next batch is in flight from here on
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const uint32_t pk = lds_read_u32(ftab_addr + 256u * (q0 + u));
                    const double b = lds_read(yrow_addr + (pk & 0xffffu))
                                     * lds_read(yrow_addr + (pk >> 16));
                    accp[u & 3] = mfma_f64(cc[u], b, accp[u & 3]);
                }
                if (q0 + PF < KS1) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) cc[u] = cn[u];
                }
            }
            // ---- softmax over k for column n (utils/misc.py:1388-1401) ------------------
            // lane holds Phi[k = kh*16 + g + 4r][n]
            mfma_settle<4>(accp);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc1[0][r] = (accp[0][r] + accp[1][r]) + (accp[2][r] + accp[3][r]);
            double mx = max_raw(acc1[0][0], acc1[0][1]);
            mx = max_raw(mx, max_raw(acc1[0][2], acc1[0][3]));
            mx = max_raw(mx, __shfl_xor(mx, 16, 64));
            mx = max_raw(mx, __shfl_xor(mx, 32, 64));
            if constexpr (KS > 1) {
                if (g == 0) xch[w * 16 + l15] = mx;
                __syncthreads();
#pragma unroll
                for (int p = 0; p < KS; ++p) mx = fmax(mx, xch[(grp * KS + p) * 16 + l15]);
            }
            if (!isfinite(mx)) mx = 0.0;
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc1[0][i];
            exp_tab_batch<4>(v, mx, tab_addr);
            double s = (v[0] + v[1]) + (v[2] + v[3]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if constexpr (KS > 1) {
                // column sums of the cluster tiles in the order 0 .. KS-1, in every wavefront
                if (g == 0) xch[WNW * 16 + w * 16 + l15] = s;
                __syncthreads();
                s = 0.0;
#pragma unroll
                for (int p = 0; p < KS; ++p) s += xch[WNW * 16 + (grp * KS + p) * 16 + l15];
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
            for (int r = 0; r < 4; ++r) rtile[(g + 4 * r) * RS + l15] = v[r] * is;
        } else {
            const int64_t lab = nok ? labels[n] : -1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rtile[(g + 4 * r) * RS + l15] = (lab == kh * 16 + g + 4 * r) ? 1.0 : 0.0;
        }
        lds_fence();

        // ---- r -> HBM: this wavefront's 16 columns of four rows per instruction ----------
        {
            const int kk = l & 15, k = kh * 16 + kk;
#pragma unroll
            for (int rr = 0; rr < TNC; rr += 4) {
                const int row = rr + (l >> 4);
                if (n0 + row < N && k < K) Rout[(n0 + row) * K + k] = rtile[kk * RS + row];
            }
        }

        // ---- phase 2: T_tile += r * feat2(y)^T  (contraction over the 16 columns) ---------
#pragma unroll 1
        for (int q = 0; q < TNC / 4; ++q) {
            const int nn = 4 * q + g;
            const double a = rtile[l15 * RS + nn];
            const uint32_t yq = ygrp_addr + (uint32_t)(4 * q * WYS * 8);
#pragma unroll
            for (int ft = 0; ft < FT2; ++ft) {
                // operand reads of four feature tiles ahead of their matrix instructions (one
                // wavefront per SIMD: nothing else hides the LDS latency)
                if ((ft & 3) == 0) asm volatile("" ::: "memory");
                const uint32_t pk = lds_read_u32(ftab_addr + 256u * (KS1 + ft));
                const double b = lds_read(yq + (pk & 0xffffu)) * lds_read(yq + (pk >> 16));
                acc2[ft] = mfma_f64(a, b, acc2[ft]);
            }
        }
        lds_fence();
    }

    // ---- partial statistics of the tile group: [KP][F2P] + 2 scalars, rows of this wavefront ----
    const int64_t plen = (int64_t)KP * F2P + 8;
    double *Pb = P + ((int64_t)blockIdx.x * GROUPS + grp) * plen;
#pragma unroll
    for (int ft = 0; ft < FT2; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Pb[(int64_t)(kh * 16 + g + 4 * r) * F2P + ft * 16 + l15] = acc2[ft][r];
    // the four lane groups of a column and the wavefronts of a tile hold identical (mx, s)
    double s_lse = (g == 0 && kh == 0) ? s_mx + s_log + log(prod) : 0.0;
    s_lse = wave_sum(s_lse);
    if (kh == 0 && l == 0) {
        Pb[(int64_t)KP * F2P + 0] = s_lse;
        Pb[(int64_t)KP * F2P + 1] = 0.0;
    }
}

template <int KT, int FT2>
int32_t launch_wide(vmp_ctx *ctx, int64_t g, const double *Y, int64_t N, int D, int K,
                    const double *Cfrag, const int64_t *labels, double *R, double *P,
                    int64_t ntiles)
{
    constexpr int KS1 = 4 * FT2;
    const size_t lds = (256 + 2 * WNW * 16 + (size_t)(KS1 + FT2) * 32
                        + (size_t)WNW * (TNC * WYS + 16 * RS)) * sizeof(double);
    auto kern = gmm_wide_kernel<KT, FT2>;
    // function attributes are per device: one flag per device of this process (a context on a
    // second GPU must set it again); the worst a race between two host threads does is set it twice
    static unsigned long long attr_devs = 0;
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(attr_devs & bit)) {
        VMP_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)kern,
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024));
        attr_devs |= bit;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(64 * WNW), lds, ctx->stream, Y, N, D, K, Cfrag,
                       labels, R, P, ntiles);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // namespace

int32_t vmp_gmm_wide_workspace_doubles(vmp_ctx *ctx, int D, int K, int64_t F2P, int64_t KP,
                                       int64_t *partial_doubles, int64_t *frag_doubles)
{
    (void)D; (void)K;
    const int KT = (int)(KP / 16);
    // one partial block per tile group, WNW / KT groups per workgroup, one workgroup per CU
    // (two per CU where the accumulators leave room for a second wavefront per SIMD)
    *partial_doubles = (int64_t)ctx->num_cu * 2 * (WNW / KT) * (KP * F2P + 8);
    *frag_doubles = KP * F2P;
    return VMP_OK;
}

int32_t vmp_gmm_wide_pass(vmp_ctx *ctx, const double *Y, int64_t N, int D, int K, int64_t F2P,
                          int64_t KP, const double *Cmat, const int64_t *labels, double *R,
                          double *P, double *Cfrag, int *nb_out)
{
    const int KT = (int)(KP / 16), FT2 = (int)(F2P / 16), KS1 = (int)(F2P / 4);
    const int GROUPS = WNW / KT;
    const int64_t ntiles = (N + TNC - 1) / TNC;
    int64_t g = (ntiles + GROUPS - 1) / GROUPS;
    // F2P <= 256: 136 accumulator registers + ~104 others, two wavefronts per SIMD fit
    const int64_t gmax = (int64_t)ctx->num_cu * (FT2 <= 16 ? 2 : 1);
    if (g > gmax) g = gmax;
    if (g < 1) g = 1;
    if (!labels) {
        hipLaunchKernelGGL(gmm_cfrag_kernel, dim3((unsigned)((KT * KS1 * 64 + 255) / 256)),
                           dim3(256), 0, ctx->stream, Cmat, KT, KS1, (int)F2P, Cfrag);
        VMP_HIP_CHECK(ctx, hipGetLastError());
    }
    int32_t rc = VMP_ERR_UNSUPPORTED;
#define VMP_WCASE(kt, ft2)                                                                    \
    if (KT == kt && FT2 == ft2)                                                               \
        rc = launch_wide<kt, ft2>(ctx, g, Y, N, D, K, Cfrag, labels, R, P, ntiles);
#define VMP_WCASE_K(ft2) VMP_WCASE(1, ft2) VMP_WCASE(2, ft2) VMP_WCASE(4, ft2)
    VMP_WCASE_K(12) VMP_WCASE_K(16) VMP_WCASE_K(20) VMP_WCASE_K(24) VMP_WCASE_K(28)
    VMP_WCASE_K(32) VMP_WCASE_K(36)
#undef VMP_WCASE_K
#undef VMP_WCASE
    if (rc == VMP_ERR_UNSUPPORTED)
        VMP_SET_ERR(ctx, "no wide GMM kernel instance for KT=%d FT2=%d", KT, FT2);
    *nb_out = (int)(g * GROUPS);
    return rc;
}
