// vmp_hmm.hip -- forward-backward ("alpha-beta") recursion of categorical Markov chains:
//
//   vmp_alpha_beta_recursion   random.alpha_beta_recursion (utils/random.py:357-422), called by
//                              CategoricalMarkovChainDistribution.compute_moments_and_cgf
//                              (categorical_markov_chain.py:107-117)
//
// The reference loops over time in Python with (plates, K, K) logsumexp temporaries per step.
// Here a group of KP lanes owns one chain, lane j owns column j of every K x K slice:
//   forward   v_ij = logalpha_n[i] + logP_n[i,j];  c = lse_ij v;  logalpha_{n+1}[j] = lse_i(v_ij - c)
//             (column sums stay inside a lane; the row vector logalpha is exchanged through LDS)
//   backward  zz_n[i,j] = softmax_ij(logalpha_n[i] + logbeta_n[j] + logP_n[i,j])   (lane-local
//             columns again), then logbeta_{n-1}[i] = lse_j(logbeta_n[j] + logP_n[i,j] - c):
//             a row reduction, done by transposing the exponentials through a padded LDS tile.
// 64 / KP chains share a wavefront, chains are independent (grid over chains), time is
// sequential.  Per chain and step: 3 K^2 exponentials and 3 K^2 fp64 words of HBM traffic
// (logP read twice, zz written once; logalpha goes through a K-vector workspace) -- the two
// are balanced on MI355X, so the kernel is bound by whichever the chain count favours; single
// chains are bound by the latency of one step.
#include "vmp_common.h"

namespace {

template <int KP>
__device__ inline double group_max(double v)
{
#pragma unroll
    for (int m = 1; m < KP; m <<= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
}

template <int KP>
__device__ inline double group_sum(double v)
{
#pragma unroll
    for (int m = 1; m < KP; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// exp(x - m) with the conventions of a masked softmax: -inf (and padded lanes) give 0
__device__ inline double exp_shift(double x, double m)
{
    return (x == -INFINITY) ? 0.0 : exp(x - m);
}

template <int KP>
__global__ __launch_bounds__(64) void alpha_beta_kernel(
    int32_t N, int32_t K, int64_t nchains, const double *__restrict__ logp0, int64_t p0_bs,
    const double *__restrict__ logP, int64_t P_bs, int64_t P_ts, double *__restrict__ z0,
    double *__restrict__ zz, double *__restrict__ g, double *__restrict__ alpha_ws)
{
    constexpr int GROUPS = 64 / KP;
    constexpr int LD = KP + 1;
    constexpr bool REG = KP <= 16;                // a K-column of fp64 values fits the registers
    __shared__ double s_vec[GROUPS][KP];          // logalpha / logbeta row vector of the group
    __shared__ double s_tile[GROUPS][KP * LD];    // transposition tile
    __shared__ double s_tile2[GROUPS][REG ? 1 : KP * LD];
    const int lane = threadIdx.x;
    const int grp = lane / KP, j = lane % KP;
    const bool act = j < K;
    double *vec = s_vec[grp];
    double *tile = s_tile[grp];
    double *tile2 = s_tile2[grp];
    // chains are handed out in rounds of GROUPS per wavefront; every lane of the wavefront runs
    // the same number of steps (N is common), idle groups compute on chain 0 and do not store
    for (int64_t c0 = (int64_t)blockIdx.x * GROUPS; c0 < nchains;
         c0 += (int64_t)gridDim.x * GROUPS) {
        const int64_t c = c0 + grp;
        const bool live = c < nchains;
        const int64_t cc = live ? c : 0;
        const double *P = logP + cc * P_bs;
        double *aw = alpha_ws + cc * (int64_t)N * K;
        double *zzc = zz + cc * (int64_t)N * K * K;
        // ---- forward -------------------------------------------------------------------------
        double la = act ? logp0[cc * p0_bs + j] : -INFINITY;     // logalpha_n[j]
        double gsum = 0.0;
        // the slices do not depend on the recursion: the column of the next instance is
        // fetched while the current one is reduced (a step is otherwise one HBM latency long)
        double pre[REG ? KP : 1];
        if constexpr (REG) {
#pragma unroll
            for (int i = 0; i < KP; ++i) pre[i] = (act && i < K) ? P[i * K + j] : -INFINITY;
        }
        for (int n = 0; n < N; ++n) {
            if (act && live) aw[(int64_t)n * K + j] = la;
            vec[j] = la;
            lds_fence();
            const double *Pn = P + (int64_t)n * P_ts;
            double col[REG ? KP : 1];
            double m = -INFINITY;
            if constexpr (REG) {
                double cur[KP];
#pragma unroll
                for (int i = 0; i < KP; ++i) cur[i] = pre[i];
                if (n + 1 < N) {
                    const double *Pq = Pn + P_ts;
#pragma unroll
                    for (int i = 0; i < KP; ++i)
                        pre[i] = (act && i < K) ? Pq[i * K + j] : -INFINITY;
                }
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    col[i] = (act && i < K) ? vec[i] + cur[i] : -INFINITY;
                    m = fmax(m, col[i]);
                }
            } else {
                // many states: the column does not fit the register file, evaluate it twice
                if (act)
                    for (int i = 0; i < K; ++i) m = fmax(m, vec[i] + Pn[i * K + j]);
            }
            m = group_max<KP>(m);
            double s = 0.0;
            if constexpr (REG) {
#pragma unroll
                for (int i = 0; i < KP; ++i) s += exp_shift(col[i], m);
            } else {
                if (act)
                    for (int i = 0; i < K; ++i) s += exp_shift(vec[i] + Pn[i * K + j], m);
            }
            const double tot = group_sum<KP>(s);
            gsum -= m + log(tot);
            la = log(s) - log(tot);              // lse_i(v_ij - c)
            lds_fence();
        }
        if (live && j == 0) g[c] = gsum;
        // the backward sweep reads logalpha values stored by OTHER lanes of this wavefront
        __threadfence_block();
        // ---- backward: zz_n and logbeta_{n-1} from the same slice ----------------------------
        double lb = 0.0;                          // logbeta_n[j]
        double prea[REG ? KP : 1];
        if constexpr (REG) {
            const double *Pl = P + (int64_t)(N - 1) * P_ts;
            const double *al = aw + (int64_t)(N - 1) * K;
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                pre[i] = (act && i < K) ? Pl[i * K + j] : -INFINITY;
                prea[i] = (act && i < K) ? al[i] : 0.0;
            }
        }
        for (int n = N - 1; n >= 0; --n) {
            const double *Pn = P + (int64_t)n * P_ts;
            const double *an = aw + (int64_t)n * K;
            double col[REG ? KP : 1], w[REG ? KP : 1];
            double mz = -INFINITY, mb = -INFINITY;
            if constexpr (REG) {
                double cur[KP], cura[KP];
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    cur[i] = pre[i];
                    cura[i] = prea[i];
                }
                if (n > 0) {
                    const double *Pq = Pn - P_ts;
                    const double *aq = an - K;
#pragma unroll
                    for (int i = 0; i < KP; ++i) {
                        pre[i] = (act && i < K) ? Pq[i * K + j] : -INFINITY;
                        prea[i] = (act && i < K) ? aq[i] : 0.0;
                    }
                }
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const bool ok = act && i < K;
                    col[i] = ok ? lb + cur[i] : -INFINITY;
                    w[i] = ok ? cura[i] + col[i] : -INFINITY;
                    mz = fmax(mz, w[i]);
                    mb = fmax(mb, col[i]);
                }
            } else if (act) {
                for (int i = 0; i < K; ++i) {
                    const double cv = lb + Pn[i * K + j];
                    mz = fmax(mz, an[i] + cv);
                    mb = fmax(mb, cv);
                }
            }
            mz = group_max<KP>(mz);
            mb = group_max<KP>(mb);
            double sz = 0.0, sb = 0.0;
            if constexpr (REG) {
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    w[i] = exp_shift(w[i], mz);
                    sz += w[i];
                    col[i] = exp_shift(col[i], mb);
                    sb += col[i];
                    tile[i * LD + j] = col[i];
                }
            } else {
                for (int i = 0; i < KP; ++i) {
                    const bool ok = act && i < K;
                    const double cv = ok ? lb + Pn[i * K + j] : -INFINITY;
                    const double ez = ok ? exp_shift(an[i] + cv, mz) : 0.0;
                    const double eb = exp_shift(cv, mb);
                    sz += ez;
                    sb += eb;
                    tile[i * LD + j] = eb;
                    tile2[i * LD + j] = ez;
                }
            }
            const double rz = 1.0 / group_sum<KP>(sz);
            const double totb = group_sum<KP>(sb);
            if (act && live) {
                if constexpr (REG) {
#pragma unroll
                    for (int i = 0; i < KP; ++i)
                        if (i < K) zzc[((int64_t)n * K + i) * K + j] = w[i] * rz;
                } else {
                    for (int i = 0; i < K; ++i)
                        zzc[((int64_t)n * K + i) * K + j] = tile2[i * LD + j] * rz;
                }
            }
            if (n == 0) {
                // z0[i] = sum_j zz_0[i,j], normalised (utils/random.py:419-420)
                lds_fence();
                if constexpr (REG) {
#pragma unroll
                    for (int i = 0; i < KP; ++i) tile[i * LD + j] = w[i] * rz;
                } else {
                    for (int i = 0; i < KP; ++i) tile[i * LD + j] = tile2[i * LD + j] * rz;
                }
                lds_fence();
                double r = 0.0;
                for (int t = 0; t < KP; ++t) r += tile[j * LD + t];
                const double rt = group_sum<KP>(act ? r : 0.0);
                if (act && live) z0[c * K + j] = r / rt;
            } else {
                lds_fence();
                double r = 0.0;                   // row j of the exponentials
                for (int t = 0; t < KP; ++t) r += tile[j * LD + t];
                lb = log(r) - log(totb);          // lse_j(v_ij - c), now indexed by this lane
                lds_fence();
            }
        }
    }
}

template <int KP>
int32_t launch_alpha_beta(vmp_ctx *ctx, int32_t N, int32_t K, int64_t nchains,
                          const double *logp0, int64_t p0_bs, const double *logP, int64_t P_bs,
                          int64_t P_ts, double *z0, double *zz, double *g, double *ws)
{
    constexpr int GROUPS = 64 / KP;
    int64_t blocks = (nchains + GROUPS - 1) / GROUPS;
    const int64_t cap = (int64_t)ctx->num_cu * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(alpha_beta_kernel<KP>, dim3((unsigned)blocks), dim3(64), 0, ctx->stream, N,
                       K, nchains, logp0, p0_bs, logP, P_bs, P_ts, z0, zz, g, ws);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // namespace

extern "C" {

int32_t vmp_alpha_beta_recursion(vmp_ctx *ctx, int32_t N, int32_t K, int64_t nchains,
                                 const double *logp0, int64_t p0_bstride, const double *logP,
                                 int64_t P_bstride, int64_t P_tstride, double *z0, double *zz,
                                 double *g, void *workspace, size_t workspace_bytes)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && logp0 && logP && z0 && zz && g, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, N >= 1 && K >= 1 && nchains >= 0, VMP_ERR_INVALID, "bad dims");
    VMP_REQUIRE(ctx, K <= 64, VMP_ERR_UNSUPPORTED,
                "the alpha-beta recursion is built for K <= 64 states (got %d)", K);
    if (nchains == 0) return VMP_OK;
    const size_t need = (size_t)nchains * (size_t)N * (size_t)K * sizeof(double);
    VMP_REQUIRE(ctx, workspace && workspace_bytes >= need, VMP_ERR_INVALID,
                "workspace too small: %zu bytes needed", need);
    double *ws = (double *)workspace;
#define GO(KP) return launch_alpha_beta<KP>(ctx, N, K, nchains, logp0, p0_bstride, logP,     \
                                            P_bstride, P_tstride, z0, zz, g, ws)
    if (K <= 2) GO(2);
    if (K <= 4) GO(4);
    if (K <= 8) GO(8);
    if (K <= 16) GO(16);
    if (K <= 32) GO(32);
    GO(64);
#undef GO
}

}  // extern "C"
