// vmp_chain.hip -- block-tridiagonal SPD solve = Kalman filter + RTS smoother of the
// Gaussian Markov chain node (linear state-space models, BASELINE.json config 5).
//
// Replaces linalg.block_banded_solve (bayespy/utils/linalg.py:468-575), which loops in
// Python over the T time instances calling SciPy chol / chol_solve / chol_inv per block:
//   forward:  V_0 = A_0;  C_n = V_n^-1 B_n;  V_{n+1} = A_{n+1} - B_n^T C_n;
//             x_{n+1} = y_{n+1} - C_n^T x_n;   logdet = sum log|V_n|
//   backward: S_{T-1} = V_{T-1}^-1;  S_n = V_n^-1 + C_n S_{n+1} C_n^T;  X_n = -C_n S_{n+1};
//             x_n = V_n^-1 x_n - C_n x_{n+1}
// The matrix recursions are sequential in T and run once per *distinct* (A, B) sequence
// (one wavefront each, one matrix element per lane, K <= 8); with shared dynamics and a
// scalar mask that is ONE sequence.  The vector recursions are batched over all
// right-hand sides (one thread per sequence, the state vector in registers).
#include "vmp_common.h"

namespace {

constexpr int NT = 256;
constexpr int CHAIN_MAXK = 16;      // K <= 8: a wavefront per matrix sequence; 8 < K <= 16: a workgroup

// C = P * Q for K x K matrices held one element per lane (through the wave's LDS buffers)
__device__ inline double wave_matmul(double p, double q, int K, int i, int j, bool act,
                                     double *Pb, double *Qb, bool transpose_p, bool transpose_q)
{
    const int l = threadIdx.x & 63;
    Pb[l] = p;
    Qb[l] = q;
    lds_fence();
    double s = 0.0;
    if (act)
        for (int k = 0; k < K; ++k)
            s += (transpose_p ? Pb[k * K + i] : Pb[i * K + k])
                 * (transpose_q ? Qb[j * K + k] : Qb[k * K + j]);
    lds_fence();
    return s;
}

__device__ inline double wave_symmetrize(double v, int K, int i, int j, bool act, double *Pb)
{
    const int l = threadIdx.x & 63;
    Pb[l] = v;
    lds_fence();
    const double r = act ? 0.5 * (Pb[i * K + j] + Pb[j * K + i]) : 0.0;
    lds_fence();
    return r;
}

// forward matrix recursion: V <- V_n^-1, C <- V_n^-1 B_n, ldet
__global__ void __launch_bounds__(NT)
bbs_forward_kernel(int T, int K, int64_t nm, const double *__restrict__ A,
                   const double *__restrict__ B, double *__restrict__ V, double *__restrict__ C,
                   double *__restrict__ ldet, int32_t *__restrict__ info)
{
    __shared__ double Ms[4][64], Ps[4][64], Qs[4][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + w;
    const bool act = (b < nm) && (l < K * K);
    const int i = act ? l / K : 0, j = act ? l - i * K : 0;
    const int64_t bb = b < nm ? b : 0;
    const int KK = K * K;
    const double *Ab = A + bb * (int64_t)T * KK;
    const double *Bb = B + bb * (int64_t)(T - 1) * KK;
    double *Vb = V + bb * (int64_t)T * KK;
    double *Cb = C + bb * (int64_t)(T - 1) * KK;
    double v = wave_symmetrize(act ? Ab[l] : 0.0, K, i, j, act, Ps[w]);
    double total = 0.0;
    int bad = 0;
    for (int n = 0; n < T; ++n) {
        double ld;
        const double vi = wave_spd_inverse(v, K, i, j, act, Ms[w], &ld, &bad);
        total += ld;
        if (act) Vb[(int64_t)n * KK + l] = vi;
        if (n < T - 1) {
            const double bn = act ? Bb[(int64_t)n * KK + l] : 0.0;
            const double c = wave_matmul(vi, bn, K, i, j, act, Ps[w], Qs[w], false, false);
            if (act) Cb[(int64_t)n * KK + l] = c;
            // B_n^T C_n
            const double s = wave_matmul(bn, c, K, i, j, act, Ps[w], Qs[w], true, false);
            const double vn = (act ? Ab[(int64_t)(n + 1) * KK + l] : 0.0) - s;
            v = wave_symmetrize(vn, K, i, j, act, Ps[w]);
        }
    }
    if (b < nm && l == 0) {
        ldet[b] = total;
        if (bad) info[b] = 1;
    }
}

// backward matrix recursion: V <- diagonal blocks of the inverse, C <- super-diagonal blocks
__global__ void __launch_bounds__(NT)
bbs_backward_kernel(int T, int K, int64_t nm, double *__restrict__ V, double *__restrict__ C)
{
    __shared__ double Ps[4][64], Qs[4][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + w;
    const bool act = (b < nm) && (l < K * K);
    const int i = act ? l / K : 0, j = act ? l - i * K : 0;
    const int64_t bb = b < nm ? b : 0;
    const int KK = K * K;
    double *Vb = V + bb * (int64_t)T * KK;
    double *Cb = C + bb * (int64_t)(T - 1) * KK;
    double S = act ? Vb[(int64_t)(T - 1) * KK + l] : 0.0;
    for (int n = T - 2; n >= 0; --n) {
        const double c = act ? Cb[(int64_t)n * KK + l] : 0.0;
        const double vi = act ? Vb[(int64_t)n * KK + l] : 0.0;
        const double t1 = wave_matmul(c, S, K, i, j, act, Ps[w], Qs[w], false, false);   // C S
        const double s2 = wave_matmul(t1, c, K, i, j, act, Ps[w], Qs[w], false, true);   // (C S) C^T
        S = wave_symmetrize(vi + s2, K, i, j, act, Ps[w]);
        if (act) {
            Vb[(int64_t)n * KK + l] = S;
            Cb[(int64_t)n * KK + l] = -t1;
        }
    }
}

// ---- 8 < K <= 16: the same recursions with one matrix element per THREAD of a 256-thread
// workgroup (one matrix sequence per workgroup), exchanges through LDS behind workgroup barriers
struct BlockMat {
    double *Ms, *Ps, *Qs;
    int K, i, j;
    bool act;

    __device__ double symmetrize(double v) const
    {
        Ps[threadIdx.x] = v;
        __syncthreads();
        const double r = act ? 0.5 * (Ps[i * K + j] + Ps[j * K + i]) : 0.0;
        __syncthreads();
        return r;
    }

    __device__ double matmul(double p, double q, bool tp, bool tq) const
    {
        Ps[threadIdx.x] = p;
        Qs[threadIdx.x] = q;
        __syncthreads();
        double s = 0.0;
        if (act)
            for (int k = 0; k < K; ++k)
                s += (tp ? Ps[k * K + i] : Ps[i * K + k]) * (tq ? Qs[j * K + k] : Qs[k * K + j]);
        __syncthreads();
        return s;
    }

    // in-place Gauss-Jordan inverse of an SPD matrix, log-determinant, definiteness flag
    __device__ double spd_inverse(double v, double *logdet, int *bad) const
    {
        double ld = 0.0, prod = 1.0;
        for (int p = 0; p < K; ++p) {
            Ms[threadIdx.x] = v;
            __syncthreads();
            const double piv = Ms[p * K + p];
            const double ci = act ? Ms[i * K + p] : 0.0, rj = act ? Ms[p * K + j] : 0.0;
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
};

__global__ void __launch_bounds__(NT)
bbs_forward_block_kernel(int T, int K, int64_t nm, const double *__restrict__ A,
                         const double *__restrict__ B, double *__restrict__ V,
                         double *__restrict__ C, double *__restrict__ ldet,
                         int32_t *__restrict__ info)
{
    __shared__ double Ms[NT], Ps[NT], Qs[NT];
    const int t = threadIdx.x;
    const int64_t b = blockIdx.x;
    const bool act = t < K * K;
    const int i = act ? t / K : 0, j = act ? t - i * K : 0;
    const BlockMat m{Ms, Ps, Qs, K, i, j, act};
    const int KK = K * K;
    const double *Ab = A + b * (int64_t)T * KK;
    const double *Bb = B + b * (int64_t)(T - 1) * KK;
    double *Vb = V + b * (int64_t)T * KK;
    double *Cb = C + b * (int64_t)(T - 1) * KK;
    double v = m.symmetrize(act ? Ab[t] : 0.0);
    double total = 0.0;
    int bad = 0;
    for (int n = 0; n < T; ++n) {
        double ld;
        const double vi = m.spd_inverse(v, &ld, &bad);
        total += ld;
        if (act) Vb[(int64_t)n * KK + t] = vi;
        if (n < T - 1) {
            const double bn = act ? Bb[(int64_t)n * KK + t] : 0.0;
            const double c = m.matmul(vi, bn, false, false);
            if (act) Cb[(int64_t)n * KK + t] = c;
            const double sden = m.matmul(bn, c, true, false);          // B_n^T C_n
            v = m.symmetrize((act ? Ab[(int64_t)(n + 1) * KK + t] : 0.0) - sden);
        }
    }
    if (t == 0) {
        ldet[b] = total;
        if (bad) info[b] = 1;
    }
}

__global__ void __launch_bounds__(NT)
bbs_backward_block_kernel(int T, int K, int64_t nm, double *__restrict__ V, double *__restrict__ C)
{
    __shared__ double Ps[NT], Qs[NT];
    const int t = threadIdx.x;
    const int64_t b = blockIdx.x;
    const bool act = t < K * K;
    const int i = act ? t / K : 0, j = act ? t - i * K : 0;
    const BlockMat m{nullptr, Ps, Qs, K, i, j, act};
    const int KK = K * K;
    double *Vb = V + b * (int64_t)T * KK;
    double *Cb = C + b * (int64_t)(T - 1) * KK;
    double S = act ? Vb[(int64_t)(T - 1) * KK + t] : 0.0;
    for (int n = T - 2; n >= 0; --n) {
        const double c = act ? Cb[(int64_t)n * KK + t] : 0.0;
        const double vi = act ? Vb[(int64_t)n * KK + t] : 0.0;
        const double t1 = m.matmul(c, S, false, false);          // C S
        const double s2 = m.matmul(t1, c, false, true);          // (C S) C^T
        S = m.symmetrize(vi + s2);
        if (act) {
            Vb[(int64_t)n * KK + t] = S;
            Cb[(int64_t)n * KK + t] = -t1;
        }
    }
}

// vector recursions, one thread per right-hand side; Vinv / Cg are the outputs of the
// forward kernel (shared when nm == 1)
template <int K>
__global__ void __launch_bounds__(NT)
bbs_vector_kernel(int T, int64_t nm, int64_t ny, const double *__restrict__ Vinv,
                  const double *__restrict__ Cg, const double *__restrict__ y,
                  double *__restrict__ x)
{
    const int64_t s = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (s >= ny) return;
    constexpr int KK = K * K;
    const int64_t bm = (nm == 1) ? 0 : s;
    const double *Vb = Vinv + bm * (int64_t)T * KK;
    const double *Cb = Cg + bm * (int64_t)(T - 1) * KK;
    const double *ys = y + s * (int64_t)T * K;
    double *xs = x + s * (int64_t)T * K;
    double xv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { xv[k] = ys[k]; xs[k] = xv[k]; }
    for (int n = 0; n < T - 1; ++n) {
        const double *c = Cb + (int64_t)n * KK;
        double xn[K];
#pragma unroll
        for (int jx = 0; jx < K; ++jx) {
            double a = ys[(int64_t)(n + 1) * K + jx];
#pragma unroll
            for (int ix = 0; ix < K; ++ix) a -= c[ix * K + jx] * xv[ix];
            xn[jx] = a;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { xv[k] = xn[k]; xs[(int64_t)(n + 1) * K + k] = xn[k]; }
    }
    {
        const double *vi = Vb + (int64_t)(T - 1) * KK;
        double xn[K];
#pragma unroll
        for (int ix = 0; ix < K; ++ix) {
            double a = 0.0;
#pragma unroll
            for (int jx = 0; jx < K; ++jx) a += vi[ix * K + jx] * xv[jx];
            xn[ix] = a;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { xv[k] = xn[k]; xs[(int64_t)(T - 1) * K + k] = xn[k]; }
    }
    for (int n = T - 2; n >= 0; --n) {
        const double *vi = Vb + (int64_t)n * KK;
        const double *c = Cb + (int64_t)n * KK;
        double xf[K], xn[K];
#pragma unroll
        for (int k = 0; k < K; ++k) xf[k] = xs[(int64_t)n * K + k];
#pragma unroll
        for (int ix = 0; ix < K; ++ix) {
            double a = 0.0;
#pragma unroll
            for (int jx = 0; jx < K; ++jx) a += vi[ix * K + jx] * xf[jx] - c[ix * K + jx] * xv[jx];
            xn[ix] = a;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { xv[k] = xn[k]; xs[(int64_t)n * K + k] = xn[k]; }
    }
}

}  // namespace

extern "C" {

int32_t vmp_block_banded_solve(vmp_ctx *ctx, int32_t T, int32_t K, int64_t nm, int64_t ny,
                               const double *A, const double *B, const double *y, double *V,
                               double *C, double *x, double *ldet, int32_t *info)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && A && y && V && x && ldet && info && (T == 1 || (B && C)),
                VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, T >= 1 && K >= 1 && nm >= 1 && ny >= 0, VMP_ERR_INVALID, "bad dims");
    VMP_REQUIRE(ctx, nm == 1 || nm == ny, VMP_ERR_INVALID,
                "matrix sequences must be shared (1) or one per right-hand side");
    VMP_REQUIRE(ctx, K <= CHAIN_MAXK, VMP_ERR_UNSUPPORTED,
                "block_banded_solve kernels support state dimension <= %d", CHAIN_MAXK);
    hipStream_t s = ctx->stream;
    VMP_HIP_CHECK(ctx, hipMemsetAsync(info, 0, (size_t)nm * sizeof(int32_t), s));
    const bool wide = K > 8;           // a workgroup per matrix sequence instead of a wavefront
    const dim3 gm((unsigned)(wide ? nm : (nm + 3) / 4));
    if (wide)
        hipLaunchKernelGGL(bbs_forward_block_kernel, gm, dim3(NT), 0, s, T, K, nm, A, B, V, C,
                           ldet, info);
    else
        hipLaunchKernelGGL(bbs_forward_kernel, gm, dim3(NT), 0, s, T, K, nm, A, B, V, C, ldet,
                           info);
    if (ny > 0) {
        const dim3 gv((unsigned)((ny + NT - 1) / NT));
#define VMP_VCASE(k)                                                                          \
    case k:                                                                                   \
        hipLaunchKernelGGL(bbs_vector_kernel<k>, gv, dim3(NT), 0, s, T, nm, ny, V, C, y, x);  \
        break;
        switch (K) {
            VMP_VCASE(1) VMP_VCASE(2) VMP_VCASE(3) VMP_VCASE(4)
            VMP_VCASE(5) VMP_VCASE(6) VMP_VCASE(7) VMP_VCASE(8)
            VMP_VCASE(9) VMP_VCASE(10) VMP_VCASE(11) VMP_VCASE(12)
            VMP_VCASE(13) VMP_VCASE(14) VMP_VCASE(15) VMP_VCASE(16)
        default: break;
        }
#undef VMP_VCASE
    }
    if (T > 1) {
        if (wide)
            hipLaunchKernelGGL(bbs_backward_block_kernel, gm, dim3(NT), 0, s, T, K, nm, V, C);
        else
            hipLaunchKernelGGL(bbs_backward_kernel, gm, dim3(NT), 0, s, T, K, nm, V, C);
    }
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // extern "C"
