// vmp_common.h -- shared host/device helpers of libvmp_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "../../include/vmp_hip.h"

#define VMP_EV_RING 64

constexpr int VMP_NME = 16;     // side-stream events of a context (vmp_mpca / vmp_lssm pipelines)

struct vmp_ctx {
    int device;
    hipStream_t stream;
    int num_cu;
    int timing;
    // ring of (start, pass end, reduce end) event triples, one per timed plate pass, so that
    // the host never has to block inside an iteration to read a duration
    hipEvent_t ev[3 * VMP_EV_RING];
    int64_t ev_n;
    // plate stream: the PCA latent pass runs here (on all but a few reserved CUs) while the
    // replicated-node kernels of the next iteration proceed on `stream`
    hipStream_t xs;
    hipEvent_t ev_xfork, ev_xdone;
    int x_pending;
    // the pass reads A from one of two private copies in turn (vmp_pca.hip run_xpass)
    hipEvent_t ev_xbuf[2];
    int x_buf_pending[2];
    int64_t x_count;
    int xs_cus;                // compute units the plate stream may use
    // streams / events of the pipelined plate pass of the missing-data PCA block (vmp_mpca.hip)
    hipStream_t ms[3];
    hipEvent_t me[VMP_NME];
    // Gram-form PCA block: the messages to W, S = [G A^T; A G A^T], of the latest latent pass are
    // still to be formed from (G, A) -- by the fused tail kernel (vmp_pca_small.hip) when the next
    // small operation is tau / alpha / bound, by vmp_pca_ensure_gram before anything else reads S
    int gram_pending;
    int gram_D, gram_K;
    double *gram_state;
    double *gram_P;
    // RCCL communicator (vmp_comm.hip); null = a world of one rank
    void *comm;
    int comm_rank, comm_world;
    // queue of SMALL generic operations (vmp_generic.hip): while it is open, vmp_ewise /
    // vmp_sum_multiply calls on a few thousand elements are recorded on the host and run later by
    // ONE launch of an interpreter kernel (vmp_queue_begin / _flush / _end)
    void *queue;
    char err[512];
};

// forms the pending S = [G A^T; A G A^T] of the Gram-form PCA block, if any (vmp_pca.hip)
int32_t vmp_pca_ensure_gram(vmp_ctx *ctx);

// every entry point that puts work on ctx->stream behind the caller's back of the queue calls this
// first: queued small operations run before anything that may read what they write
int32_t vmp_queue_flush(vmp_ctx *ctx);
int32_t vmp_queue_commit(vmp_ctx *ctx);
int32_t destroy_small_queue(vmp_ctx *ctx);       // internal: frees the queue with the context
extern const char *vmp_flush_cause;               // diagnostic: who asked for the flush (VMP_QUEUE_TRACE=1)
#define VMP_FLUSH_SMALL(ctx)                                   \
    do {                                                       \
        if ((ctx) && (ctx)->queue) {                           \
            vmp_flush_cause = __func__;                        \
            const int32_t rc__ = vmp_queue_flush(ctx);         \
            if (rc__ != VMP_OK) return rc__;                   \
        }                                                      \
    } while (0)

// event triple of the plate pass being issued (timing enabled); advances the ring
static inline hipEvent_t *vmp_next_events(vmp_ctx *ctx)
{
    hipEvent_t *e = ctx->ev + 3 * (ctx->ev_n % VMP_EV_RING);
    ctx->ev_n += 1;
    return e;
}

#define VMP_SET_ERR(ctx, ...)                                         \
    do {                                                              \
        if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
    } while (0)

#define VMP_HIP_CHECK(ctx, expr)                                                   \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            VMP_SET_ERR(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                       \
            return VMP_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define VMP_REQUIRE(ctx, cond, code, ...)   \
    do {                                    \
        if (!(cond)) {                      \
            VMP_SET_ERR(ctx, __VA_ARGS__);  \
            return (code);                  \
        }                                   \
    } while (0)

// Measurement knobs (vmp_tune_set, include/vmp_hip.h): `dflt` unless a value was set for `key`.
int vmp_tune_get(const char *key, int dflt);

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

// vmp_spd_mfma.hip: batched SPD inverse / Gaussian moments for 16 < n <= 32 on the matrix cores
int32_t vmp_launch_spd_mfma(vmp_ctx *ctx, bool moments, int32_t n, int64_t batch, const double *A,
                            const double *rhs, double *Ainv, double *vec_out, double *logdet,
                            int32_t *info);

// ---------------------------------------------------------------------------
// Special functions (E17-E19 of SURVEY.md 2.2: scipy.special.digamma / gammaln
// call sites gamma.py:145-147, dirichlet.py:150-158, utils/misc.py:1146-1151).
// Host+device so the CPU test-suite can check them against SciPy.
// ---------------------------------------------------------------------------

// psi(x) for x > 0: upward recurrence to x >= 10, then the asymptotic series
// ln x - 1/(2x) - sum_n B_2n / (2n x^2n)  (truncation error < 1e-17 at x >= 10).
__host__ __device__ inline double vmp_digamma(double x)
{
    if (!(x > 0.0)) {
        if (x == 0.0) return -INFINITY;
        // reflection psi(1-x) - psi(x) = pi cot(pi x); not on the hot path
        double r = 1.0 - x;
        double s = 0.0;
        while (r < 10.0) { s -= 1.0 / r; r += 1.0; }
        double inv = 1.0 / r, inv2 = inv * inv;
        double ser = inv2 * (1.0 / 12 - inv2 * (1.0 / 120 - inv2 * (1.0 / 252 - inv2 * (1.0 / 240 - inv2 * (1.0 / 132 - inv2 * (691.0 / 32760 - inv2 * (1.0 / 12)))))));
        double psi1mx = s + log(r) - 0.5 * inv - ser;
        return psi1mx - M_PI / tan(M_PI * x);
    }
    double s = 0.0;
    while (x < 10.0) { s -= 1.0 / x; x += 1.0; }
    double inv = 1.0 / x, inv2 = inv * inv;
    double ser = inv2 * (1.0 / 12 - inv2 * (1.0 / 120 - inv2 * (1.0 / 252 - inv2 * (1.0 / 240 - inv2 * (1.0 / 132 - inv2 * (691.0 / 32760 - inv2 * (1.0 / 12)))))));
    return s + log(x) - 0.5 * inv - ser;
}

// ln Gamma(x) for x > 0: downward-recurrence-free Stirling series at x >= 10,
// product shift below (keeps ~1e-15 relative accuracy; matches scipy.special.gammaln).
__host__ __device__ inline double vmp_lgamma(double x)
{
    if (!(x > 0.0)) return INFINITY;
    double shift = 0.0;
    // accumulate log of the product in pieces to avoid overflow
    double prod = 1.0;
    while (x < 10.0) {
        prod *= x;
        x += 1.0;
        if (prod > 1e150) { shift += log(prod); prod = 1.0; }
    }
    shift += log(prod);
    double inv = 1.0 / x, inv2 = inv * inv;
    // sum_n B_2n / (2n (2n-1) x^(2n-1))
    double ser = inv * (1.0 / 12 - inv2 * (1.0 / 360 - inv2 * (1.0 / 1260 - inv2 * (1.0 / 1680 - inv2 * (1.0 / 1188 - inv2 * (691.0 / 360360 - inv2 * (1.0 / 156)))))));
    return (x - 0.5) * log(x) - x + 0.91893853320467274178 + ser - shift;
}

// psi'(x) (trigamma; scipy.special.polygamma(1, .) call sites gamma.py:210,
// dirichlet.py:230) for x > 0: upward recurrence to x >= 10, then the asymptotic series
// 1/x + 1/(2x^2) + sum_n B_2n / x^(2n+1).
__host__ __device__ inline double vmp_trigamma(double x)
{
    if (!(x > 0.0)) return (x == 0.0) ? INFINITY : NAN;
    double s = 0.0;
    while (x < 10.0) { s += 1.0 / (x * x); x += 1.0; }
    double inv = 1.0 / x, inv2 = inv * inv;
    double ser = inv * inv2 * (1.0 / 6 - inv2 * (1.0 / 30 - inv2 * (1.0 / 42 - inv2 * (1.0 / 30 - inv2 * (5.0 / 66 - inv2 * (691.0 / 2730 - inv2 * (7.0 / 6)))))));
    return s + inv + 0.5 * inv2 + ser;
}

#ifdef __HIPCC__
// Running log-determinant without a log per pivot: the product of the pivots is kept in
// `prod` and folded into `ld` only when it leaves a safe range (fp64 log is ~1000 cycles on
// the serial critical path of a Gauss-Jordan sweep).
__device__ inline void logdet_accumulate(double piv, double &prod, double &ld)
{
    prod *= piv;
    if (!(prod < 1e120 && prod > 1e-120)) {
        ld += log(prod);
        prod = 1.0;
    }
}

__device__ inline double logdet_finish(double prod, double ld) { return ld + log(prod); }

// 1/x to fp64 round-off: hardware estimate + two Newton steps (avoids the ~40-instruction
// IEEE division sequence on the serial critical path).
__device__ inline double fast_recip(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}

__device__ inline void lds_fence()
{
    // LDS operations of one wavefront complete in issue order: waiting for the
    // outstanding ones makes this wave's writes visible to all of its lanes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}


// in: v = element (i,j) of an SPD matrix (lanes >= D*D idle); out: element of the inverse
__device__ inline double wave_spd_inverse(double v, int D, int i, int j, bool act, double *M,
                                          double *logdet, int *bad)
{
    double ld = 0.0, prod = 1.0;
    const int l = threadIdx.x & 63;
    for (int p = 0; p < D; ++p) {
        M[l] = v;
        lds_fence();
        const double piv = M[p * D + p];
        const double ci = act ? M[i * D + p] : 0.0, rj = act ? M[p * D + j] : 0.0;
        if (!(piv > 0.0)) *bad = 1;
        logdet_accumulate(piv, prod, ld);
        const double d = fast_recip(piv);
        if (i == p) v = (j == p) ? d : rj * d;
        else if (j == p) v = -ci * d;
        else v = v - ci * rj * d;
        lds_fence();
    }
    *logdet = logdet_finish(prod, ld);
    return v;
}

// exp(x) for x <= 0 (softmax arguments after subtracting the maximum; -inf allowed): the
// library routine spends a third of its instructions on overflow / special-case handling that
// cannot occur here, and on this chip fp64 VALU work is not hidden behind fp64 MFMA work (both
// run on the same fp64 units).  Cody-Waite reduction by ln 2, degree-12 Taylor polynomial on
// |r| <= ln2/2 (truncation 2e-16 relative), scaling by ldexp (underflows to 0 by itself).
__device__ __forceinline__ double exp_nonpos(double x)
{
    x = fmax(x, -800.0);
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// ---------------------------------------------------------------------------
// Wavefront (64 lanes) and workgroup reductions, fixed order => deterministic.
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // valid in lane 0
}

// Sum over a workgroup of NT threads (NT multiple of 64, <= 1024); the result
// is returned to ALL threads.  `red` is LDS scratch of >= NT/64 doubles.
template <int NT>
__device__ inline double block_sum(double v, double *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}
#endif
