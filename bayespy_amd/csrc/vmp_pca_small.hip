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

namespace {


struct small_args {
    vmp_pca_layout L;
    int D, K;
    double n_total, x_prec, a0t, b0t, a0a, b0a;
};

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
        logdet_accumulate(piv, prod, ld);
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
        *logdet = logdet_finish(prod, ld);
        if (isbad) *bad = 1;
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
            s.Sb[r * LDM + k] = (r < nr && k < K) ? Syx[(r0 + r) * KP + k] : 0.0;
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
            acc *= tau;
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
        const double be = a.b0a + 0.5 * st[L.off_Sww + k * KP + k];
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
            saw += x * st[L.off_Sww + k * KP + k];
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
    if (threadIdx.x == 0) s.bad = 0;
    run_op<KP, NTQ, O0>(a, st, s);
    run_op<KP, NTQ, O1>(a, st, s);
    run_op<KP, NTQ, O2>(a, st, s);
    __syncthreads();
    if (threadIdx.x == 0 && s.bad) st[a.L.off_scal + 3] = (double)VMP_ERR_NOT_POSDEF;
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
                     const int32_t *ops, double *state)
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
    // greedy: the two sequences a VB iteration produces are single launches, anything else
    // falls back to one launch per operation
    int i = 0;
    while (i < nops) {
        const int o0 = ops[i], o1 = i + 1 < nops ? ops[i + 1] : 0, o2 = i + 2 < nops ? ops[i + 2] : 0;
        if (o0 == VMP_PCA_OP_W && o1 == VMP_PCA_OP_XPREP) {
            launch_sequence<VMP_PCA_OP_W, VMP_PCA_OP_XPREP, 0>(ctx, a, state);
            i += 2;
        } else if (o0 == VMP_PCA_OP_TAU && o1 == VMP_PCA_OP_ALPHA && o2 == VMP_PCA_OP_ELBO) {
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
