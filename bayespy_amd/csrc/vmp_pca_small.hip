// vmp_pca_small.hip -- replicated-node updates of the fused PCA block (gfx950).
//
// W.update(), the replicated half of X.update(), tau.update(), alpha.update() and the
// lower bound work on K x K / D x K state only.  They are latency-, not throughput-bound,
// and at shard sizes of a million columns they decide the step time.  So the sequences a VB
// iteration produces -- (W, X-replicated) and (tau, alpha, lower bound) -- are ONE launch
// each: a single 256-thread workgroup runs the operations back to back (no launch gaps,
// state re-read from L2 by the same CU), small enough (< 64 KB LDS, one wavefront per SIMD)
// to run beside the streaming grid of vmp_pca_xpass, i.e. concurrently with the plate pass
// of the previous iteration.
//
// Reference formulas: gaussian.py:649-706 (GaussianARD phi / moments / cgf),
// dot.py:581 (message E4), gaussian.py:2344-2369 (WrapToGaussianGamma), gamma.py:116-160,
// expfamily.py:400-480 (lower bound), utils/linalg.py:31-223 (chol, chol_inv, chol_logdet).
#include "vmp_common.h"
#include "vmp_sweep.h"

#include <stdlib.h>

namespace {


struct small_args {
    vmp_pca_layout L;
    int D, K;
    double n_total, x_prec, a0t, b0t, a0a, b0a;
    int has_mean;       // W has a constant non-zero prior mean mu (state[off_mu]); see op_update_w
};

// sum_d <(w_dk - mu_dk)^2> = Sww_kk - 2 sum_d mu_dk <w_dk> + sum_d mu_dk^2: what the Gamma message
// of the ARD prior (gaussian.py:2344-2369) and the bound term of W see of a non-zero prior mean;
// the two sums are kept in state[off_mstat] by op_update_w
__device__ inline double ww_centred(const small_args &a, const double *st, int k)
{
    const vmp_pca_layout &L = a.L;
    double v = st[L.off_Sww + k * L.KP + k];
    if (a.has_mean) v += st[L.off_mstat + L.KP + k] - 2.0 * st[L.off_mstat + k];
    return v;
}

// <x x^T> total statistic: n_total * Cov_X + sum_n <x><x>^T, symmetrised.
__device__ inline double sxx_total(const double *st, const vmp_pca_layout &L, double n_total,
                                   int i, int j)
{
    const double *S = st + L.off_S;
    const double m = 0.5 * (S[(L.DP + i) * L.KP + j] + S[(L.DP + j) * L.KP + i]);
    return n_total * st[L.off_CX + i * L.KP + j] + m;
}

// One shared copy of each special function (the fused kernels are instruction-fetch
// sensitive when they run beside the streaming pass: 21 KB of inlined code took longer to
// fetch than to execute).
__device__ __noinline__ double sf_lgamma(double x) { return vmp_lgamma(x); }
__device__ __noinline__ double sf_digamma(double x) { return vmp_digamma(x); }
__device__ __noinline__ double sf_log(double x) { return log(x); }

// In-place inverse of the SPD matrix in M (KP x (KP+1) in LDS, entries beyond n x n hold
// the identity) by pivot-free Gauss-Jordan sweeps, all KP^2 elements updated in parallel
// per pivot.  Replaces chol / chol_inv / chol_logdet (utils/linalg.py:31-63, :174-223).
template <int KP, int NTQ>
__device__ void gj_inverse(double *M, int n, double *logdet, int *bad)
{
    constexpr int LDM = KP + 1;
    constexpr int E = KP * KP / NTQ;
    const int tid = threadIdx.x;
    double ld = 0.0, prod = 1.0;
    int isbad = 0;
#pragma unroll 1
    for (int p = 0; p < n; ++p) {
        __syncthreads();
        const double piv = M[p * LDM + p];
        double ci[E], rj[E], me[E];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTQ;
            const int i = e / KP, j = e % KP;
            ci[m] = M[i * LDM + p];
            rj[m] = M[p * LDM + j];
            me[m] = M[i * LDM + j];
        }
        if (!(piv > 0.0)) isbad = 1;
        // running log-determinant: fold the pivot product only when it leaves a safe range
        prod *= piv;
        if (!(prod < 1e120 && prod > 1e-120)) {
            ld += sf_log(prod);
            prod = 1.0;
        }
        const double d = fast_recip(piv);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTQ;
            const int i = e / KP, j = e % KP;
            double v;
            if (i == p) v = (j == p) ? d : rj[m] * d;
            else if (j == p) v = -ci[m] * d;
            else v = me[m] - ci[m] * rj[m] * d;
            M[i * LDM + j] = v;
        }
    }
    __syncthreads();
    if (tid == 0) {
        *logdet = ld + sf_log(prod);
        if (isbad) *bad = 1;
    }
    __syncthreads();
}

// The same inverse for KP <= 32 by ONE wavefront in registers (vmp_sweep.h: symmetric sweep with
// 4 x 4 pivot blocks on the matrix cores): no LDS traffic and no barriers between the pivots.
// gj_inverse above takes 2 barriers and 16 LDS accesses per pivot -- 14 of the 41 us of the fused
// W / X-prepare kernel per inverse when it runs alone, and 70+ us each beside the streaming grid of
// the plate pass, whose wavefronts keep the CU's LDS pipeline busy (measured with wall_clock64
// stamps at N = 1.25e6, the shard one of 8 ranks holds at the headline size).
template <int KP>
__device__ __forceinline__ void sweep_inverse(double *M, int n, double *logdet, int *bad)
{
    static_assert(KP == 16 || KP == 32, "one 32 x 32 register tile set");
    constexpr int LDM = KP + 1;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int l = threadIdx.x, l15 = l & 15, l4 = l >> 4;
        v4f64 T[2][2];
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * tr + l4 + 4 * r, col = 16 * tc + l15;
                    T[tr][tc][r] = (row < KP && col < KP) ? M[row * LDM + col]
                                                          : (row == col ? 1.0 : 0.0);
                }
        double prod = 1.0, ld = 0.0;
        int isbad = 0;
        vmp_sweep::sweep_upto<0>(T, (n + 3) / 4, l15, l4, prod, ld, isbad);
#pragma unroll
        for (int tr = 0; tr < KP / 16; ++tr)
#pragma unroll
            for (int tc = 0; tc < KP / 16; ++tc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * tr + l4 + 4 * r, col = 16 * tc + l15;
                    // rows / columns beyond n were not swept: they keep the identity
                    M[row * LDM + col] = (row < 4 * ((n + 3) / 4) && col < 4 * ((n + 3) / 4))
                                             ? -T[tr][tc][r] : T[tr][tc][r];
                }
        if (l == 0) {
            *logdet = vmp_sweep::sweep_logdet(prod, ld);
            if (isbad) *bad = 1;
        }
    }
    __syncthreads();
}

template <int KP, int NTQ>
struct small_lds {
    static constexpr int LDM = KP + 1;
    static constexpr int RB = (KP == 64) ? 16 : 32;   // row block of the D x K products
    double M[KP * LDM];
    double Sb[RB * LDM];
    double Wb[RB * LDM];
    double red[NTQ / 64 + 1];
    double logdet;
    int bad;
};

// W.update(): Lambda_W = diag<alpha> + <tau> Sxx ; Cov_W ; <W> = <tau> Syx Cov_W ;
// Sww = D Cov_W + W^T W.
template <int KP, int NTQ>
__device__ void op_update_w(const small_args &a, double *st, small_lds<KP, NTQ> &s)
{
    constexpr int LDM = KP + 1, RB = small_lds<KP, NTQ>::RB, E = KP * KP / NTQ;
    const vmp_pca_layout &L = a.L;
    const int tid = threadIdx.x, D = a.D, K = a.K;
    const double tau = st[L.off_tau + 2];
    // gaussian.py:656-670 (prior phi) + dot.py:581 (message E4)
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTQ;
        const int i = e / KP, j = e % KP;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < K && j < K) {
            v = tau * sxx_total(st, L, a.n_total, i, j);
            if (i == j) v += st[L.off_alpha + 2 * KP + i];
        }
        s.M[i * LDM + j] = v;
    }
    gj_inverse<KP, NTQ>(s.M, K, &s.logdet, &s.bad);
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTQ;
        const int i = e / KP, j = e % KP;
        if (i < K && j < K) st[L.off_CW + i * KP + j] = s.M[i * LDM + j];
    }
    const double *Syx = st + L.off_S;
    double *W = st + L.off_W;
    double sw[E];
#pragma unroll
    for (int m = 0; m < E; ++m) sw[m] = 0.0;
    for (int r0 = 0; r0 < D; r0 += RB) {
        const int nr = (D - r0) < RB ? (D - r0) : RB;
        __syncthreads();
#pragma unroll 1
        for (int e = tid; e < RB * KP; e += NTQ) {
            const int r = e / KP, k = e % KP;
            double v = (r < nr && k < K) ? Syx[(r0 + r) * KP + k] : 0.0;
            // a non-zero prior mean: phi0_d = <tau> Syx[d] + <alpha> * mu_d (gaussian.py:656-670)
            if (a.has_mean && r < nr && k < K)
                v = tau * v + st[L.off_alpha + 2 * KP + k] * st[L.off_mu + (r0 + r) * KP + k];
            s.Sb[r * LDM + k] = v;
        }
        __syncthreads();
        // <w_d> = Cov_W phi0_d,  phi0_d = <tau> Syx[d]        (gaussian.py:694)
#pragma unroll 1
        for (int e = tid; e < RB * KP; e += NTQ) {
            const int r = e / KP, k = e % KP;
            double acc = 0.0;
            if (k < K)
#pragma unroll 4
                for (int j = 0; j < K; ++j) acc += s.Sb[r * LDM + j] * s.M[j * LDM + k];
            if (!a.has_mean) acc *= tau;
            s.Wb[r * LDM + k] = acc;
            if (r < nr && k < K) W[(r0 + r) * KP + k] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTQ;
            const int i = e / KP, j = e % KP;
            double acc = 0.0;
#pragma unroll 4
            for (int r = 0; r < RB; ++r) acc += s.Wb[r * LDM + i] * s.Wb[r * LDM + j];
            sw[m] += acc;
        }
    }
    // Sww = sum_d <w_d w_d^T> = D Cov_W + W^T W                (gaussian.py:695)
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTQ;
        const int i = e / KP, j = e % KP;
        if (i < K && j < K)
            st[L.off_Sww + i * KP + j] = (double)D * s.M[i * LDM + j] + sw[m];
    }
    if (tid == 0) st[L.off_scal + 0] = s.logdet;
    if (a.has_mean) {
        // sum_d mu_dk <w_dk> and sum_d mu_dk^2 for the Gamma message and the bound term of W
        __threadfence_block();
        __syncthreads();
        for (int k = tid; k < K; k += NTQ) {
            double m1 = 0.0, m2 = 0.0;
            for (int d = 0; d < D; ++d) {
                const double mu = st[L.off_mu + d * KP + k];
                m1 += mu * W[d * KP + k];
                m2 += mu * mu;
            }
            st[L.off_mstat + k] = m1;
            st[L.off_mstat + KP + k] = m2;
        }
    }
}

// X.update(), replicated half: Lambda_X = x_prec I + <tau> Sww ; Cov_X ; A = <tau> Cov_X W^T.
template <int KP, int NTQ>
__device__ void op_prepare_x(const small_args &a, double *st, small_lds<KP, NTQ> &s)
{
    constexpr int LDM = KP + 1, RB = small_lds<KP, NTQ>::RB, E = KP * KP / NTQ;
    const vmp_pca_layout &L = a.L;
    const int tid = threadIdx.x, D = a.D, K = a.K;
    const int DP = (int)L.DP;
    const double tau = st[L.off_tau + 2];
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTQ;
        const int i = e / KP, j = e % KP;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < K && j < K) {
            v = tau * 0.5 * (st[L.off_Sww + i * KP + j] + st[L.off_Sww + j * KP + i]);
            if (i == j) v += a.x_prec;
        }
        s.M[i * LDM + j] = v;
    }
    gj_inverse<KP, NTQ>(s.M, K, &s.logdet, &s.bad);
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTQ;
        const int i = e / KP, j = e % KP;
        if (i < K && j < K) st[L.off_CX + i * KP + j] = s.M[i * LDM + j];
    }
    const double *W = st + L.off_W;
    for (int r0 = 0; r0 < D; r0 += RB) {
        const int nr = (D - r0) < RB ? (D - r0) : RB;
        __syncthreads();
#pragma unroll 1
        for (int e = tid; e < RB * KP; e += NTQ) {
            const int r = e / KP, k = e % KP;
            s.Wb[r * LDM + k] = (r < nr && k < K) ? W[(r0 + r) * KP + k] : 0.0;
        }
        __syncthreads();
        // A[k][d] = <tau> sum_j Cov_X[k][j] W[d][j]; r fastest => coalesced stores
#pragma unroll 1
        for (int e = tid; e < RB * KP; e += NTQ) {
            const int k = e / RB, r = e % RB;
            if (k < K && r < nr) {
                double acc = 0.0;
#pragma unroll 4
                for (int j = 0; j < K; ++j) acc += s.M[k * LDM + j] * s.Wb[r * LDM + j];
                st[L.off_A + (int64_t)k * DP + r0 + r] = tau * acc;
            }
        }
    }
    if (tid == 0) st[L.off_scal + 1] = s.logdet;
}

// sum_dn <(y_dn - f_dn)^2> = Syy - 2 sum(W o Syx) + sum(Sww o Sxx)
// (dot.py:355 E1 and dot.py:403 E2 collapsed to traces).
template <int NTQ>
__device__ double pca_residual(const small_args &a, const double *st, double *red)
{
    const vmp_pca_layout &L = a.L;
    const int tid = threadIdx.x, D = a.D, K = a.K;
    const int KP = (int)L.KP;
    double t1 = 0.0, t2 = 0.0;
    // padded entries of W are zero, so the D x KP blocks can be walked linearly
#pragma unroll 4
    for (int e = tid; e < D * KP; e += NTQ) t1 += st[L.off_W + e] * st[L.off_S + e];
    for (int e = tid; e < K * K; e += NTQ) {
        const int i = e / K, j = e - i * K;
        t2 += st[L.off_Sww + i * KP + j] * sxx_total(st, L, a.n_total, i, j);
    }
    t1 = block_sum<NTQ>(t1, red);
    t2 = block_sum<NTQ>(t2, red);
    return st[L.off_Syy] - 2.0 * t1 + t2;
}

template <int NTQ>
__device__ void op_update_tau(const small_args &a, double *st, double *red)
{
    const vmp_pca_layout &L = a.L;
    const double resid = pca_residual<NTQ>(a, st, red);
    if (threadIdx.x == 0) {
        // gamma.py:116-122 phi = [-b, a] with the message gaussian.py:2363-2369
        const double al = a.a0t + 0.5 * (double)a.D * a.n_total;
        const double be = a.b0t + 0.5 * resid;
        st[L.off_tau + 0] = al;
        st[L.off_tau + 1] = be;
        st[L.off_tau + 2] = al / be;                       // gamma.py:144
        st[L.off_tau + 3] = sf_digamma(al) - sf_log(be);     // gamma.py:145
        st[L.off_scal + 2] = resid;
        if (!(be > 0.0)) st[L.off_scal + 3] = (double)VMP_ERR_FLOATING;
    }
}

template <int NTQ>
__device__ void op_update_alpha(const small_args &a, double *st)
{
    const vmp_pca_layout &L = a.L;
    const int KP = (int)L.KP;
    for (int k = threadIdx.x; k < a.K; k += NTQ) {
        const double al = a.a0a + 0.5 * (double)a.D;
        const double be = a.b0a + 0.5 * ww_centred(a, st, k);
        st[L.off_alpha + 0 * KP + k] = al;
        st[L.off_alpha + 1 * KP + k] = be;
        st[L.off_alpha + 2 * KP + k] = al / be;
        st[L.off_alpha + 3 * KP + k] = sf_digamma(al) - sf_log(be);
    }
}

// E[log p - log q] of a Gamma(a,b) node with prior Gamma(a0,b0)
// (expfamily.py:400-480 with gamma.py:147,160).
__device__ inline double gamma_elbo(double a0, double b0, double a, double b, double x,
                                    double logx)
{
    const double g_p = a0 * sf_log(b0) - sf_lgamma(a0);
    const double g_q = a * sf_log(b) - sf_lgamma(a);
    return g_p - g_q + (b - b0) * x + (a0 - a) * logx;
}

template <int NTQ>
__device__ void op_lower_bound(const small_args &a, double *st, double *red)
{
    const vmp_pca_layout &L = a.L;
    const int tid = threadIdx.x, D = a.D, K = a.K;
    const int KP = (int)L.KP;
    const double resid = pca_residual<NTQ>(a, st, red);
    double trx = 0.0, sla = 0.0, saw = 0.0, lal = 0.0;
    // index K stands for the tau node: every Gamma term goes through one copy of the
    // special-function code (register pressure)
#pragma unroll 1
    for (int k = tid; k <= K; k += NTQ) {
        const bool is_tau = (k == K);
        const int64_t oa = is_tau ? L.off_tau : L.off_alpha + k;
        const int64_t sa = is_tau ? 1 : KP;
        const double al = st[oa], be = st[oa + sa], x = st[oa + 2 * sa], lx = st[oa + 3 * sa];
        const double g = gamma_elbo(is_tau ? a.a0t : a.a0a, is_tau ? a.b0t : a.b0a, al, be, x, lx);
        if (is_tau) {
            red[NTQ / 64] = g;
        } else {
            trx += sxx_total(st, L, a.n_total, k, k);
            sla += lx;
            saw += x * ww_centred(a, st, k);
            lal += g;
        }
    }
    trx = block_sum<NTQ>(trx, red);
    sla = block_sum<NTQ>(sla, red);
    saw = block_sum<NTQ>(saw, red);
    lal = block_sum<NTQ>(lal, red);
    if (tid == 0) {
        const double tau = st[L.off_tau + 2], logtau = st[L.off_tau + 3];
        const double Dd = (double)D, Kd = (double)K;
        const double LY = Dd * a.n_total * (-0.5 * sf_log(2.0 * M_PI) + 0.5 * logtau)
                          - 0.5 * tau * resid;
        const double LX = -0.5 * a.x_prec * trx
                          + a.n_total * (0.5 * Kd * sf_log(a.x_prec) - 0.5 * st[L.off_scal + 1]
                                         + 0.5 * Kd);
        const double LW = 0.5 * Dd * sla - 0.5 * saw + Dd * (-0.5 * st[L.off_scal + 0] + 0.5 * Kd);
        const double Lt = red[NTQ / 64];
        st[L.off_L + 0] = LY;
        st[L.off_L + 1] = LX;
        st[L.off_L + 2] = LW;
        st[L.off_L + 3] = Lt;
        st[L.off_L + 4] = lal;
        st[L.off_L + 5] = LY + LX + LW + Lt + lal;
        st[L.off_scal + 2] = resid;
    }
}

template <int KP, int NTQ, int OP>
__device__ __forceinline__ void run_op(const small_args &a, double *st, small_lds<KP, NTQ> &s)
{
    if constexpr (OP != 0) {
        // the previous operation's global writes are read back by this workgroup only
        __threadfence_block();
        __syncthreads();
    }
    if constexpr (OP == VMP_PCA_OP_W) op_update_w<KP, NTQ>(a, st, s);
    if constexpr (OP == VMP_PCA_OP_XPREP) op_prepare_x<KP, NTQ>(a, st, s);
    if constexpr (OP == VMP_PCA_OP_TAU) op_update_tau<NTQ>(a, st, s.red);
    if constexpr (OP == VMP_PCA_OP_ALPHA) op_update_alpha<NTQ>(a, st);
    if constexpr (OP == VMP_PCA_OP_ELBO) op_lower_bound<NTQ>(a, st, s.red);
}

// The operation sequence is a template parameter: straight-line code keeps the register
// footprint at the maximum over the operations instead of their union.
template <int KP, int NTQ, int O0, int O1, int O2>
__global__ void __launch_bounds__(NTQ)
pca_small_kernel(small_args a, double *st)
{
    __shared__ small_lds<KP, NTQ> s;
    __builtin_amdgcn_s_setprio(3);   // latency-critical: win issue arbitration on a shared CU
    if (threadIdx.x == 0) s.bad = 0;
    run_op<KP, NTQ, O0>(a, st, s);
    run_op<KP, NTQ, O1>(a, st, s);
    run_op<KP, NTQ, O2>(a, st, s);
    __syncthreads();
    if (threadIdx.x == 0 && s.bad) st[a.L.off_scal + 3] = (double)VMP_ERR_NOT_POSDEF;
}

// ---------------------------------------------------------------------------
// LDS-resident forms of the two sequences of a VB iteration, for K <= 32 and D <= 128.
//
// Beside the streaming grid every dependent global access of this workgroup queues behind
// the pass kernel's outstanding loads on the same CU (~10 us per round trip, measured), so
// the generic operations above -- a dozen round trips each -- take 100-200 us there.  These
// kernels fetch their whole working set in one batch, compute from LDS, and store at the end.
// ---------------------------------------------------------------------------
constexpr int FAST_D = 128;
constexpr int NTF = 256;

// the state block is a few hundred KB: 32-bit element offsets keep the index arithmetic
// of these kernels off the 64-bit VALU paths (registers)
struct lay32 {
    int DP, KP, off_S, off_Syy, off_tau, off_alpha, off_W, off_CW, off_Sww, off_CX, off_A,
        off_scal, off_L;
    __device__ explicit lay32(const vmp_pca_layout &l)
        : DP((int)l.DP), KP((int)l.KP), off_S((int)l.off_S), off_Syy((int)l.off_Syy),
          off_tau((int)l.off_tau), off_alpha((int)l.off_alpha), off_W((int)l.off_W),
          off_CW((int)l.off_CW), off_Sww((int)l.off_Sww), off_CX((int)l.off_CX),
          off_A((int)l.off_A), off_scal((int)l.off_scal), off_L((int)l.off_L) {}
};

// One wavefront, one 16 x 16 tile of C = A B from LDS operands with arbitrary strides
// (v_mfma_f64_16x16x4_f64; A(m,k) = As[m*am + k*ak], B(k,n) = Bs[k*bk + n*bn], kc % 4 == 0).
// Result layout: element (row = (lane>>4) + 4*reg, col = lane&15).
__device__ __forceinline__ v4f64 wave_tile_mma(const double *As, int am, int ak, const double *Bs,
                                               int bk, int bn, int kc)
{
    const int l = threadIdx.x & 63, l15 = l & 15, l4 = l >> 4;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int q = 0; q < kc; q += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[l15 * am + (q + l4) * ak],
                                                   Bs[(q + l4) * bk + l15 * bn], acc, 0, 0, 0);
    return acc;
}

// (W.update(), X.update() replicated half)
// accumulate form of wave_tile_mma
__device__ __forceinline__ v4f64 wave_tile_mma_acc(v4f64 acc, const double *As, int am, int ak,
                                                   const double *Bs, int bk, int bn, int kc)
{
    const int l = threadIdx.x & 63, l15 = l & 15, l4 = l >> 4;
#pragma unroll 4
    for (int q = 0; q < kc; q += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[l15 * am + (q + l4) * ak],
                                                   Bs[(q + l4) * bk + l15 * bn], acc, 0, 0, 0);
    return acc;
}

// (W.update(), X.update() replicated half)
//
// Placement beside the plate pass decides this kernel's latency at the multi-GPU shard size, and
// what decides the placement was measured (tools/timeline.sh, tools/experiments/
// placement_microbench.hip): beside three resident pass workgroups per CU (32 KB of
// LDS, 122 VGPRs each) a workgroup asking for 52 KB of LDS is not placed until the persistent grid
// drains, whatever its register count; one asking for 26 KB and <= 96 registers is placed at once.
// So the D x K operand is staged through ONE 32-row block (8.4 KB) instead of a D-row array
// (33.8 KB), the rows of sum y<x>^T are loaded only after the first sweep and the rows of <W> come
// back from global memory for the last product instead of waiting in registers, the inverses are
// inlined (no call, no scratch) and the file is compiled with the VGPR form of the MFMAs (Makefile):
// 25.6 KB, 96 VGPRs, no AGPRs, no scratch.
template <int KP>
__global__ void __launch_bounds__(NTF, 4)
pca_head_fast_kernel(small_args a, double *st)
{
    constexpr int LDM = KP + 1, E = KP * KP / NTF, RB = 32;
    constexpr int KT = KP / 16;
    constexpr int NB = FAST_D / RB;               // row blocks
    constexpr int NLB = RB * KP / NTF;            // loads per thread and row block
    constexpr int TPB = 2 * KT;                   // 16 x 16 tiles of a row block
    constexpr int WT = (TPB + 3) / 4;             // ... per wavefront
    constexpr int ST = (KT * KT + 3) / 4;         // tiles of Sww per wavefront
    __shared__ double M[KP * LDM];          // Lambda_W -> Cov_W -> Lambda_X -> Cov_X
    __shared__ double T[KP * LDM];          // sum <x x^T> (shard sum) -> Sww
    __shared__ double SWb[RB * LDM];        // one row block: Syx rows -> <W> rows
    __shared__ double alm[KP];
    __shared__ double logdet;
    __shared__ int bad;
    const lay32 L(a.L);
    int tid = threadIdx.x;
    const int D = a.D, K = a.K;
    const int DP = (int)L.DP;
    const int nb = (D + RB - 1) / RB;
    __builtin_amdgcn_s_setprio(3);   // latency-critical: win issue arbitration on a shared CU
    if (tid == 0) bad = 0;
    // ---- one batch of loads ------------------------------------------------------------
    const double tau = st[L.off_tau + 2];
    {
        double cx[E], sx[E];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTF;
            const int i = e / KP, j = e % KP;
            const bool in = (i < K && j < K);
            cx[m] = in ? st[L.off_CX + i * KP + j] : 0.0;
            sx[m] = in ? st[L.off_S + (L.DP + i) * KP + j] : 0.0;
        }
        if (tid < KP) alm[tid] = tid < K ? st[L.off_alpha + 2 * KP + tid] : 0.0;
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTF;
            M[(e / KP) * LDM + (e % KP)] = cx[m];
            T[(e / KP) * LDM + (e % KP)] = sx[m];
        }
    }
    __syncthreads();
    // ---- W: Lambda_W = diag<alpha> + <tau> Sxx  (gaussian.py:656-670, dot.py:581) -------
    {
        double lam[E];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTF;
            const int i = e / KP, j = e % KP;
            double x = (i == j) ? 1.0 : 0.0;
            if (i < K && j < K) {
                x = tau * (a.n_total * M[i * LDM + j] + 0.5 * (T[i * LDM + j] + T[j * LDM + i]));
                if (i == j) x += alm[i];
            }
            lam[m] = x;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTF;
            M[(e / KP) * LDM + (e % KP)] = lam[m];
        }
    }
    asm volatile("" : "+v"(tid));     // nothing derived from the thread index stays live across the sweep
    sweep_inverse<KP>(M, K, &logdet, &bad);
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTF;
        const int i = e / KP, j = e % KP;
        if (i < K && j < K) st[L.off_CW + i * KP + j] = M[i * LDM + j];
    }
    if (tid == 0) st[L.off_scal + 0] = logdet;
    int w = tid >> 6, l15 = tid & 15, l4 = (tid & 63) >> 4;
    // <w_d> = <tau> Cov_W Syx[d]  (gaussian.py:694), one 32-row block at a time on the matrix cores;
    // Sww = D Cov_W + W^T W  (gaussian.py:695) accumulated over the blocks
    // rows of sum y<x>^T: one batch of loads, issued only now so that they do not occupy registers
    // during the sweep (the kernel must stay within ~90 VGPRs to be placed beside the plate pass)
    double v[NB][NLB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int m = 0; m < NLB; ++m) {
            const int e = tid + m * NTF;
            const int r = b * RB + e / KP, k = e % KP;
            v[b][m] = (r < D && k < K) ? st[L.off_S + r * KP + k] : 0.0;
        }
    v4f64 wt[WT];
    v4f64 sacc[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t) sacc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b < nb) {
#pragma unroll
            for (int m = 0; m < NLB; ++m) {
                const int e = tid + m * NTF;
                SWb[(e / KP) * LDM + (e % KP)] = v[b][m];
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                const int tile = w + 4 * t;
                if (tile < TPB)
                    wt[t] = wave_tile_mma(SWb + (tile / KT) * 16 * LDM, LDM, 1,
                                             M + (tile % KT) * 16, LDM, 1, KP);
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                const int tile = w + 4 * t;
                if (tile < TPB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lr = (tile / KT) * 16 + l4 + 4 * r, k = (tile % KT) * 16 + l15;
                        const double x = tau * wt[t][r];
                        SWb[lr * LDM + k] = x;
                        const int row = b * RB + lr;
                        if (row < D && k < K) st[L.off_W + row * KP + k] = x;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                const int tile = w + 4 * t;
                if (tile < KT * KT)
                    sacc[t] = wave_tile_mma_acc(sacc[t], SWb + (tile / KT) * 16, 1, LDM,
                                                SWb + (tile % KT) * 16, LDM, 1, RB);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) {
        const int tile = w + 4 * t;
        if (tile < KT * KT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = (tile / KT) * 16 + l4 + 4 * r, j = (tile % KT) * 16 + l15;
                const double sww = (double)D * M[i * LDM + j] + sacc[t][r];
                T[i * LDM + j] = sww;
                if (i < K && j < K) st[L.off_Sww + i * KP + j] = sww;
            }
        }
    }
    __syncthreads();
    // ---- X, replicated half: Lambda_X = x_prec I + <tau> Sww ; A = <tau> Cov_X W^T --------
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTF;
        const int i = e / KP, j = e % KP;
        double x = (i == j) ? 1.0 : 0.0;
        if (i < K && j < K) {
            x = tau * 0.5 * (T[i * LDM + j] + T[j * LDM + i]);
            if (i == j) x += a.x_prec;
        }
        M[i * LDM + j] = x;
    }
    asm volatile("" : "+v"(tid));
    sweep_inverse<KP>(M, K, &logdet, &bad);
    asm volatile("" : "+v"(tid));
    w = tid >> 6, l15 = tid & 15, l4 = (tid & 63) >> 4;
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTF;
        const int i = e / KP, j = e % KP;
        if (i < K && j < K) st[L.off_CX + i * KP + j] = M[i * LDM + j];
    }
    // A = <tau> Cov_X W^T : per row block (KP x 32) = (KP x KP)(KP x 32); the rows of <W> come back
    // from global memory (written above by this workgroup, ordered by the barriers since; these
    // addresses were not read before, so no stale line can be hit)
    double wv[NB][NLB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int m = 0; m < NLB; ++m) {
            const int e = tid + m * NTF;
            const int r = b * RB + e / KP, k = e % KP;
            wv[b][m] = (r < D && k < K) ? st[L.off_W + r * KP + k] : 0.0;
        }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b < nb) {
#pragma unroll
            for (int m = 0; m < NLB; ++m) {
                const int e = tid + m * NTF;
                SWb[(e / KP) * LDM + (e % KP)] = wv[b][m];
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                const int tile = w + 4 * t;
                if (tile < TPB) {
                    const int tk = tile % KT, td = tile / KT;
                    const v4f64 acc = wave_tile_mma(M + tk * 16 * LDM, LDM, 1, SWb + td * 16 * LDM, 1,
                                                    LDM, KP);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = tk * 16 + l4 + 4 * r, d = b * RB + td * 16 + l15;
                        if (k < K && d < D) st[L.off_A + k * DP + d] = tau * acc[r];
                    }
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        st[L.off_scal + 1] = logdet;
        if (bad) st[L.off_scal + 3] = (double)VMP_ERR_NOT_POSDEF;
    }
}

// (tau.update(), alpha.update(), lower bound)
// GRAM: the messages to W of the Gram form, S = [G A^T ; A G A^T] (sum y <x>^T and
// sum <x><x>^T collapsed onto the constant Gram matrix), are formed HERE first -- G streamed through
// LDS in batches of GB rows, A resident -- instead of by pca_gram_stats_kernel + reduce_partials_kernel
// in front of this kernel: one launch of the replicated-node chain instead of three beside the plate
// pass (BASELINE config 2: the chain, not the pass, set the iteration time).
constexpr int GB = 32;            // rows of G per batch
template <int KP, bool GRAM>
__global__ void __launch_bounds__(NTF)
pca_tail_fast_kernel(small_args a, double *st)
{
    constexpr int LDM = KP + 1, E = KP * KP / NTF;
    extern __shared__ double gl[];          // GRAM: A (KP x (DP+1)) | G batch (GB x (DP+1)) | S batch (GB x LDM)
    __shared__ double Sw[KP * LDM];         // Sww
    __shared__ double Cx[KP * LDM];         // Cov_X
    __shared__ double Sx[KP * LDM];         // sum <x><x>^T (shard sum)
    __shared__ double al[4 * KP];           // alpha: a, b, <alpha>, <log alpha>
    __shared__ double sc[8];                // tau a, b, mean, logmean ; Syy ; log|LW| ; log|LX|
    __shared__ double red[NTF / 64 + 1];
    const lay32 L(a.L);
    const int tid = threadIdx.x, D = a.D, K = a.K;
    __builtin_amdgcn_s_setprio(3);   // latency-critical: win issue arbitration on a shared CU
    double t1 = 0.0;
    double sxx[E];
#pragma unroll
    for (int m = 0; m < E; ++m) sxx[m] = 0.0;
    if constexpr (GRAM) {
        const int DP = L.DP, LA = DP + 1;
        double *As = gl, *Gs = As + KP * LA, *Ss = Gs + GB * LA;
        for (int e = tid; e < KP * DP; e += NTF) {
            const int k = e / DP, d = e - k * DP;
            As[k * LA + d] = (k < K && d < D) ? st[L.off_A + k * DP + d] : 0.0;
        }
        for (int r0 = 0; r0 < D; r0 += GB) {
            for (int e = tid; e < GB * DP; e += NTF) {
                const int r = e / DP, d = e - r * DP;
                Gs[r * LA + d] = (r0 + r < D && d < D) ? st[(int)a.L.off_G + (r0 + r) * DP + d] : 0.0;
            }
            __syncthreads();
            // Syx[r][k] = sum_d G[r][d] A[k][d]  (and sum W o Syx on the way)
#pragma unroll
            for (int m = 0; m < GB * KP / NTF; ++m) {
                const int e = tid + m * NTF;
                const int r = e / KP, k = e % KP;
                const double *gr = Gs + r * LA, *ar = As + k * LA;
                double s0 = 0.0, s1 = 0.0;
                for (int d = 0; d < D; d += 2) {
                    s0 += gr[d] * ar[d];
                    s1 += gr[d + 1] * ar[d + 1];        // (d + 1 <= DP - 1: zero padded)
                }
                const double sv = s0 + s1;
                Ss[r * LDM + k] = sv;
                if (r0 + r < D) {
                    st[L.off_S + (r0 + r) * KP + k] = sv;
                    if (k < K) t1 += st[L.off_W + (r0 + r) * KP + k] * sv;
                }
            }
            __syncthreads();
            // Sxx[k][k'] += sum_{r in batch} A[k][r] Syx[r][k']
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const int e = tid + m * NTF;
                const int i = e / KP, j = e % KP;
                double s0 = 0.0;
#pragma unroll 8
                for (int r = 0; r < GB; ++r) s0 += As[i * LA + r0 + r] * Ss[r * LDM + j];
                sxx[m] += s0;
            }
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTF;
            const int i = e / KP, j = e % KP;
            st[L.off_S + (L.DP + i) * KP + j] = (i < K && j < K) ? sxx[m] : 0.0;
        }
    }
    // ---- one batch of loads ------------------------------------------------------------
    {
        double v0[E], v1[E], v2[E];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTF;
            const int i = e / KP, j = e % KP;
            const bool in = (i < K && j < K);
            v0[m] = in ? st[L.off_Sww + i * KP + j] : 0.0;
            v1[m] = in ? st[L.off_CX + i * KP + j] : 0.0;
            if constexpr (GRAM) v2[m] = in ? sxx[m] : 0.0;
            else v2[m] = in ? st[L.off_S + (L.DP + i) * KP + j] : 0.0;
        }
        if (tid == 0) {
            sc[4] = st[L.off_Syy];
            sc[5] = st[L.off_scal + 0];
            sc[6] = st[L.off_scal + 1];
        }
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int e = tid + m * NTF;
            const int i = e / KP, j = e % KP;
            Sw[i * LDM + j] = v0[m];
            Cx[i * LDM + j] = v1[m];
            Sx[i * LDM + j] = v2[m];
        }
    }
    // sum(W o Syx): padded entries of W are zero, so the D x KP blocks are walked linearly
    if constexpr (!GRAM) {
        const int n = D * KP;
#pragma unroll 1
        for (int e0 = 0; e0 < n; e0 += 8 * NTF) {
            double w[8], y[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int e = e0 + tid + m * NTF;
                w[m] = e < n ? st[L.off_W + e] : 0.0;
                y[m] = e < n ? st[L.off_S + e] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) t1 += w[m] * y[m];
        }
    }
    __syncthreads();
    double t2 = 0.0, trx = 0.0;
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int e = tid + m * NTF;
        const int i = e / KP, j = e % KP;
        const double sxx = a.n_total * Cx[i * LDM + j] + 0.5 * (Sx[i * LDM + j] + Sx[j * LDM + i]);
        t2 += Sw[i * LDM + j] * sxx;
        if (i == j) trx += sxx;
    }
    t1 = block_sum<NTF>(t1, red);
    t2 = block_sum<NTF>(t2, red);
    trx = block_sum<NTF>(trx, red);
    // (dot.py:355 E1 and dot.py:403 E2 collapsed to traces)
    const double resid = sc[4] - 2.0 * t1 + t2;
    // ---- tau (gamma.py:116-148 with the message gaussian.py:2363-2369), alpha -------------
    if (tid == 64) {
        const double ta = a.a0t + 0.5 * (double)D * a.n_total;
        const double tb = a.b0t + 0.5 * resid;
        sc[0] = ta;
        sc[1] = tb;
        sc[2] = ta / tb;
        sc[3] = sf_digamma(ta) - sf_log(tb);
        st[L.off_tau + 0] = ta;
        st[L.off_tau + 1] = tb;
        st[L.off_tau + 2] = sc[2];
        st[L.off_tau + 3] = sc[3];
        st[L.off_scal + 2] = resid;
        if (!(tb > 0.0)) st[L.off_scal + 3] = (double)VMP_ERR_FLOATING;
    }
    if (tid < K) {
        const double aa = a.a0a + 0.5 * (double)D;
        const double ab = a.b0a + 0.5 * Sw[tid * LDM + tid];
        const double am = aa / ab, alg = sf_digamma(aa) - sf_log(ab);
        al[0 * KP + tid] = aa;
        al[1 * KP + tid] = ab;
        al[2 * KP + tid] = am;
        al[3 * KP + tid] = alg;
        st[L.off_alpha + 0 * KP + tid] = aa;
        st[L.off_alpha + 1 * KP + tid] = ab;
        st[L.off_alpha + 2 * KP + tid] = am;
        st[L.off_alpha + 3 * KP + tid] = alg;
    }
    __syncthreads();
    // ---- lower bound (expfamily.py:400-480) ----------------------------------------------
    double sla = 0.0, saw = 0.0, lal = 0.0;
    if (tid <= K) {
        const bool is_tau = (tid == K);
        const double *p = is_tau ? sc : al + tid;
        const int sp = is_tau ? 1 : KP;
        const double g = gamma_elbo(is_tau ? a.a0t : a.a0a, is_tau ? a.b0t : a.b0a, p[0], p[sp],
                                    p[2 * sp], p[3 * sp]);
        if (is_tau) {
            red[NTF / 64] = g;
        } else {
            sla = p[3 * sp];
            saw = p[2 * sp] * Sw[tid * LDM + tid];
            lal = g;
        }
    }
    sla = block_sum<NTF>(sla, red);
    saw = block_sum<NTF>(saw, red);
    lal = block_sum<NTF>(lal, red);
    if (tid == 0) {
        const double tau = sc[2], logtau = sc[3];
        const double Dd = (double)D, Kd = (double)K;
        const double LY = Dd * a.n_total * (-0.5 * sf_log(2.0 * M_PI) + 0.5 * logtau)
                          - 0.5 * tau * resid;
        const double LX = -0.5 * a.x_prec * trx
                          + a.n_total * (0.5 * Kd * sf_log(a.x_prec) - 0.5 * sc[6] + 0.5 * Kd);
        const double LW = 0.5 * Dd * sla - 0.5 * saw + Dd * (-0.5 * sc[5] + 0.5 * Kd);
        const double Lt = red[NTF / 64];
        st[L.off_L + 0] = LY;
        st[L.off_L + 1] = LX;
        st[L.off_L + 2] = LW;
        st[L.off_L + 3] = Lt;
        st[L.off_L + 4] = lal;
        st[L.off_L + 5] = LY + LX + LW + Lt + lal;
    }
}

template <int O0, int O1, int O2>
void launch_sequence(vmp_ctx *ctx, const small_args &a, double *state)
{
    // 256 threads (fits beside the streaming grid), except the K x K inverses of K > 32
    constexpr bool has_inverse = (O0 == VMP_PCA_OP_W || O0 == VMP_PCA_OP_XPREP ||
                                  O1 == VMP_PCA_OP_W || O1 == VMP_PCA_OP_XPREP);
    constexpr int NT64 = has_inverse ? 1024 : 256;
    if (a.L.KP == 16)
        hipLaunchKernelGGL((pca_small_kernel<16, 256, O0, O1, O2>), dim3(1), dim3(256), 0,
                           ctx->stream, a, state);
    else if (a.L.KP == 32)
        hipLaunchKernelGGL((pca_small_kernel<32, 256, O0, O1, O2>), dim3(1), dim3(256), 0,
                           ctx->stream, a, state);
    else
        hipLaunchKernelGGL((pca_small_kernel<64, NT64, O0, O1, O2>), dim3(1), dim3(NT64), 0,
                           ctx->stream, a, state);
}

int32_t launch_small(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total, double x_prec,
                     double a0t, double b0t, double a0a, double b0a, int32_t nops,
                     const int32_t *ops, double *state, int32_t has_mean = 0)
{
    VMP_REQUIRE(ctx, ctx && state && ops, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, nops >= 1 && nops <= VMP_PCA_MAX_OPS, VMP_ERR_INVALID,
                "between 1 and %d operations per call", VMP_PCA_MAX_OPS);
    small_args a;
    int32_t rc = vmp_pca_get_layout(D, K, &a.L);
    VMP_REQUIRE(ctx, rc == VMP_OK, rc, "unsupported dims D=%d K=%d", D, K);
    a.D = D;
    a.K = K;
    a.n_total = (double)n_total;
    a.x_prec = x_prec;
    a.a0t = a0t; a.b0t = b0t; a.a0a = a0a; a.b0a = b0a;
    a.has_mean = has_mean ? 1 : 0;
    for (int i = 0; i < nops; ++i) {
        const int op = ops[i];
        VMP_REQUIRE(ctx, op >= VMP_PCA_OP_W && op <= VMP_PCA_OP_ELBO, VMP_ERR_INVALID,
                    "unknown operation code %d", op);
        if (op == VMP_PCA_OP_XPREP || op == VMP_PCA_OP_ELBO)
            VMP_REQUIRE(ctx, x_prec > 0, VMP_ERR_INVALID, "x_prec must be positive");
        if (op == VMP_PCA_OP_TAU || op == VMP_PCA_OP_ELBO)
            VMP_REQUIRE(ctx, a0t > 0 && b0t > 0, VMP_ERR_INVALID,
                        "Gamma prior parameters must be positive");
        if (op == VMP_PCA_OP_ALPHA || op == VMP_PCA_OP_ELBO)
            VMP_REQUIRE(ctx, a0a > 0 && b0a > 0, VMP_ERR_INVALID,
                        "Gamma prior parameters must be positive");
    }
    // greedy: the two sequences a VB iteration produces are single launches (LDS-resident
    // forms when K <= 32 and D <= 128; VMP_PCA_FAST_SMALL=0 disables them), anything else
    // falls back to one launch per operation
    static const int fast_enabled = getenv("VMP_PCA_FAST_SMALL") ? atoi(getenv("VMP_PCA_FAST_SMALL")) : 1;
    // (the LDS-resident forms are built for the zero prior mean of the demo model)
    const bool fast = fast_enabled && a.L.KP <= 32 && D <= FAST_D && !a.has_mean;
    // the Gram-form messages to W of the latest latent pass may still be pending (vmp_pca.hip
    // run_xpass): the LDS-resident tail kernel forms them itself when it comes first, anything
    // else has them formed now
    bool gram_here = false;
    if (ctx->gram_pending) {
        const bool tail_first = nops >= 3 && ops[0] == VMP_PCA_OP_TAU && ops[1] == VMP_PCA_OP_ALPHA
                                && ops[2] == VMP_PCA_OP_ELBO;
        if (fast && tail_first && ctx->gram_state == state && ctx->gram_D == D && ctx->gram_K == K) {
            gram_here = true;
            ctx->gram_pending = 0;
        } else {
            rc = vmp_pca_ensure_gram(ctx);
            if (rc != VMP_OK) return rc;
        }
    }
    int i = 0;
    while (i < nops) {
        const int o0 = ops[i], o1 = i + 1 < nops ? ops[i + 1] : 0, o2 = i + 2 < nops ? ops[i + 2] : 0;
        if (o0 == VMP_PCA_OP_W && o1 == VMP_PCA_OP_XPREP) {
            if (fast && a.L.KP == 16)
                hipLaunchKernelGGL(pca_head_fast_kernel<16>, dim3(1), dim3(NTF), 0, ctx->stream, a,
                                   state);
            else if (fast)
                hipLaunchKernelGGL(pca_head_fast_kernel<32>, dim3(1), dim3(NTF), 0, ctx->stream, a,
                                   state);
            else
                launch_sequence<VMP_PCA_OP_W, VMP_PCA_OP_XPREP, 0>(ctx, a, state);
            i += 2;
        } else if (o0 == VMP_PCA_OP_TAU && o1 == VMP_PCA_OP_ALPHA && o2 == VMP_PCA_OP_ELBO) {
            if (fast && gram_here) {
                // S = [G A^T; A G A^T] of the latest pass is formed by the tail kernel itself
                const size_t lds = ((size_t)(a.L.KP + GB) * (a.L.DP + 1) + (size_t)GB * (a.L.KP + 1))
                                   * sizeof(double);
                static bool raised[64] = {false};
                const int dev = ctx->device & 63;
                if (!raised[dev]) {
                    VMP_HIP_CHECK(ctx, hipFuncSetAttribute(
                        (const void *)pca_tail_fast_kernel<16, true>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    VMP_HIP_CHECK(ctx, hipFuncSetAttribute(
                        (const void *)pca_tail_fast_kernel<32, true>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    raised[dev] = true;
                }
                if (a.L.KP == 16)
                    hipLaunchKernelGGL((pca_tail_fast_kernel<16, true>), dim3(1), dim3(NTF), lds,
                                       ctx->stream, a, state);
                else
                    hipLaunchKernelGGL((pca_tail_fast_kernel<32, true>), dim3(1), dim3(NTF), lds,
                                       ctx->stream, a, state);
                gram_here = false;
            } else if (fast && a.L.KP == 16)
                hipLaunchKernelGGL((pca_tail_fast_kernel<16, false>), dim3(1), dim3(NTF), 0,
                                   ctx->stream, a, state);
            else if (fast)
                hipLaunchKernelGGL((pca_tail_fast_kernel<32, false>), dim3(1), dim3(NTF), 0,
                                   ctx->stream, a, state);
            else
                launch_sequence<VMP_PCA_OP_TAU, VMP_PCA_OP_ALPHA, VMP_PCA_OP_ELBO>(ctx, a, state);
            i += 3;
        } else {
            switch (o0) {
            case VMP_PCA_OP_W: launch_sequence<VMP_PCA_OP_W, 0, 0>(ctx, a, state); break;
            case VMP_PCA_OP_XPREP: launch_sequence<VMP_PCA_OP_XPREP, 0, 0>(ctx, a, state); break;
            case VMP_PCA_OP_TAU: launch_sequence<VMP_PCA_OP_TAU, 0, 0>(ctx, a, state); break;
            case VMP_PCA_OP_ALPHA: launch_sequence<VMP_PCA_OP_ALPHA, 0, 0>(ctx, a, state); break;
            default: launch_sequence<VMP_PCA_OP_ELBO, 0, 0>(ctx, a, state); break;
            }
            i += 1;
        }
        VMP_HIP_CHECK(ctx, hipGetLastError());
    }
    return VMP_OK;
}

}  // namespace

extern "C" {

int32_t vmp_pca_small_ops(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total, double x_prec,
                          double a0_tau, double b0_tau, double a0_alpha, double b0_alpha,
                          int32_t nops, const int32_t *ops, double *state)
{
    return launch_small(ctx, D, K, n_total, x_prec, a0_tau, b0_tau, a0_alpha, b0_alpha, nops, ops,
                        state);
}

int32_t vmp_pca_small_ops_mean(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total, double x_prec,
                               double a0_tau, double b0_tau, double a0_alpha, double b0_alpha,
                               int32_t nops, const int32_t *ops, int32_t has_mean, double *state)
{
    return launch_small(ctx, D, K, n_total, x_prec, a0_tau, b0_tau, a0_alpha, b0_alpha, nops, ops,
                        state, has_mean);
}

int32_t vmp_pca_update_w(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total, double *state)
{
    const int32_t op = VMP_PCA_OP_W;
    return launch_small(ctx, D, K, n_total, 0.0, 0.0, 0.0, 0.0, 0.0, 1, &op, state);
}

int32_t vmp_pca_prepare_x(vmp_ctx *ctx, int32_t D, int32_t K, double x_prec, double *state)
{
    const int32_t op = VMP_PCA_OP_XPREP;
    return launch_small(ctx, D, K, 0, x_prec, 0.0, 0.0, 0.0, 0.0, 1, &op, state);
}

int32_t vmp_pca_update_tau(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total, double a0,
                           double b0, double *state)
{
    const int32_t op = VMP_PCA_OP_TAU;
    return launch_small(ctx, D, K, n_total, 0.0, a0, b0, 0.0, 0.0, 1, &op, state);
}

int32_t vmp_pca_update_alpha(vmp_ctx *ctx, int32_t D, int32_t K, double a0, double b0,
                             double *state)
{
    const int32_t op = VMP_PCA_OP_ALPHA;
    return launch_small(ctx, D, K, 0, 0.0, 0.0, 0.0, a0, b0, 1, &op, state);
}

int32_t vmp_pca_lower_bound(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total, double x_prec,
                            double a0_tau, double b0_tau, double a0_alpha, double b0_alpha,
                            double *state)
{
    const int32_t op = VMP_PCA_OP_ELBO;
    return launch_small(ctx, D, K, n_total, x_prec, a0_tau, b0_tau, a0_alpha, b0_alpha, 1, &op,
                        state);
}

}  // extern "C"
