/*
 * vmp_hip.h -- C ABI of libvmp_hip.so: the MI355X (gfx950) VMP hot path.
 *
 * The reference (bayespy/bayespy) has no FFI: its "operator interface" for this
 * path is the Python Distribution / Deterministic node protocol
 * (bayespy/inference/vmp/nodes/stochastic.py:16-80, expfamily.py:17-70,
 * deterministic.py:16-96) whose bodies are NumPy/SciPy call sites.  Each entry
 * point below replaces one group of those call sites; the citation next to it
 * names the reference code it stands in for.  INTEGRATION.md shows the ctypes
 * binding a bayespy maintainer would add.
 *
 * Conventions
 *  - all device pointers are plain `double*` / `int64_t*` into HBM, IEEE fp64;
 *  - every function returns an int32 status (VMP_OK or a negative VMP_ERR_*),
 *    is asynchronous on the context's HIP stream, never owns caller memory and
 *    is thread-compatible (one context per host thread);
 *  - "plates" are the reference's leading broadcast axes; the big observation
 *    plate N is always the contiguous (coalescing) axis of device arrays.
 */
#ifndef VMP_HIP_H
#define VMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMP_OK               0
#define VMP_ERR_INVALID     -1  /* shape / plate mismatch      -> ValueError (node.py:343-356)            */
#define VMP_ERR_NOT_POSDEF  -2  /* "Matrix not positive definite" (utils/linalg.py:58-59)                 */
#define VMP_ERR_HIP         -3  /* HIP runtime failure         -> RuntimeError                            */
#define VMP_ERR_UNSUPPORTED -4  /* size outside the built kernels -> NotImplementedError (wishart.py:138) */
#define VMP_ERR_FLOATING    -5  /* invalid / divide            -> FloatingPointError (gamma.py:142-144)   */
#define VMP_ERR_NOT_POSITIVE -6 /* "Natural parameters should be positive" (dirichlet.py:147-148)         */

typedef struct vmp_ctx vmp_ctx;

/* ---- context, memory ----------------------------------------------------- */

/* Create a context on `device`.  `stream` is a hipStream_t (NULL = the null
 * stream); the caller keeps ownership of the stream. */
int32_t vmp_ctx_create(int32_t device, void *stream, vmp_ctx **out);
int32_t vmp_ctx_destroy(vmp_ctx *ctx);
int32_t vmp_ctx_set_stream(vmp_ctx *ctx, void *stream);
int32_t vmp_ctx_sync(vmp_ctx *ctx);
/* Number of compute units of the context's device (256 on MI355X). */
int32_t vmp_ctx_num_cu(vmp_ctx *ctx);
/* Human-readable text of the last error recorded on this context. */
const char *vmp_last_error(vmp_ctx *ctx);
const char *vmp_version(void);

int32_t vmp_malloc(vmp_ctx *ctx, size_t bytes, void **ptr);
int32_t vmp_free(vmp_ctx *ctx, void *ptr);
int32_t vmp_memcpy_h2d(vmp_ctx *ctx, void *dst, const void *src, size_t bytes);
int32_t vmp_memcpy_d2h(vmp_ctx *ctx, void *dst, const void *src, size_t bytes);
int32_t vmp_memset_zero(vmp_ctx *ctx, void *dst, size_t bytes);

/* ---- fused probabilistic-PCA / factor-analysis block --------------------- *
 *
 * Model block  Y = GaussianARD(SumMultiply('i,i', W, X), tau),  W ~ GaussianARD(0, alpha),
 * X ~ GaussianARD(0, x_prec), tau, alpha ~ Gamma   (bayespy/demos/pca.py:22-61),
 * fully observed (scalar mask).  One VB iteration reads Y exactly once.
 *
 * All replicated (small) quantities live in ONE device block of doubles, the
 * "state", whose layout is given by vmp_pca_get_layout(); the caller allocates
 * `total` doubles.  The statistics region S is what ranks all-reduce (sum)
 * when the observation plate N is sharded.
 */
typedef struct vmp_pca_layout {
    int64_t DP, KP;     /* padded dims the kernels use (DP mult. of 32, KP of 16)            */
    int64_t off_S;      /* (DP+KP) x KP : rows [0,D) = sum_n y_n <x_n>^T (dot.py:581 msg to W) */
                        /*               rows [DP,DP+K) = sum_n <x_n><x_n>^T                   */
    int64_t len_S;      /* (DP+KP)*KP                                                          */
    int64_t off_Syy;    /* 1 : sum_dn y_dn^2  (constant)                                       */
    int64_t off_tau;    /* 4 : a, b, <tau>, <log tau>                (gamma.py:142-148)        */
    int64_t off_alpha;  /* 4*KP : a[KP], b[KP], <alpha>[KP], <log alpha>[KP]                   */
    int64_t off_W;      /* D x KP   <W>, row-major ld=KP             (gaussian.py:692-699)     */
    int64_t off_CW;     /* KP x KP  Cov_W (shared by all d)                                    */
    int64_t off_Sww;    /* KP x KP  sum_d <w_d w_d^T>                                          */
    int64_t off_CX;     /* KP x KP  Cov_X (shared by all n; zero for delta-initialised X)      */
    int64_t off_A;      /* KP x DP  A = <tau> Cov_X <W>^T, zero padded                         */
    int64_t off_G;      /* DP x DP  Gram matrix G = Y Y^T (constant; summed over ranks once)   */
    int64_t off_scal;   /* 8 : [0] log|Lambda_W|  [1] log|Lambda_X|  [2] residual  [3] status  */
    int64_t off_L;      /* 8 : L_Y, L_X, L_W, L_tau, L_alpha, L_total                          */
    int64_t total;      /* doubles in the state block                                          */
} vmp_pca_layout;

int32_t vmp_pca_get_layout(int32_t D, int32_t K, vmp_pca_layout *out);
/* Bytes of scratch (per-workgroup partial statistics) the pass needs. */
int32_t vmp_pca_workspace_bytes(vmp_ctx *ctx, int32_t D, int32_t K, size_t *bytes);

/* Prior moments for tau / alpha (ExponentialFamily.initialize_from_prior,
 * expfamily.py:168-184), zero Cov_X, zero statistics. */
int32_t vmp_pca_init_state(vmp_ctx *ctx, int32_t D, int32_t K,
                           double a0_tau, double b0_tau,
                           double a0_alpha, double b0_alpha, double *state);

/* sum_dn y^2 over the local shard -> state[off_Syy]  (part of E9/E16,
 * gaussian.py:628-635).  Y is (D, N) row-major with leading dimension ldy. */
int32_t vmp_pca_syy(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N,
                    int32_t D, int32_t K, double *state, void *workspace);

/* Statistics of a GIVEN X (delta moments after initialize_from_value,
 * expfamily.py:193-204): S <- [Y X^T ; X X^T] over the local shard.
 * X is (K, N) row-major, leading dimension ldx. */
int32_t vmp_pca_stats_from_x(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N,
                             int32_t D, int32_t K, const double *X, int64_t ldx,
                             double *state, void *workspace);

/* W.update(): GaussianARDDistribution.compute_phi_from_parents + messages E3/E4
 * + compute_moments_and_cgf (gaussian.py:649-706, dot.py:581).  Uses S (already
 * summed over ranks), <tau>, <alpha>; writes W, CW, Sww, log|Lambda_W|. */
int32_t vmp_pca_update_w(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total,
                         double *state);

/* X.update(), replicated half: Lambda_X = x_prec I + <tau> Sww, Cov_X, A
 * (gaussian.py:649-706 with messages E5/E6 of dot.py:581). */
int32_t vmp_pca_prepare_x(vmp_ctx *ctx, int32_t D, int32_t K, double x_prec,
                          double *state);

/* G <- Y Y^T over the local shard (set-up; the caller all-reduces G once when the
 * plate is sharded).  With a scalar mask every <x_n> is the same linear map of y_n,
 * so the messages to W (dot.py:581) collapse onto G:  sum y<x>^T = G A^T and
 * sum <x><x>^T = A G A^T. */
int32_t vmp_pca_gram(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N,
                     int32_t D, int32_t K, double *state, void *workspace);

/* X.update(), plate half, Gram form (default): for every local n  <x_n> = A y_n
 * is written to X ((K,N) row-major) -- read Y once, write <x> once, HBM-bound --
 * then S <- [G A^T ; A G A^T] from the (already global) Gram matrix: no
 * per-iteration collective.  fp64 MFMA (v_mfma_f64_16x16x4_f64). */
int32_t vmp_pca_xpass(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N,
                      int32_t D, int32_t K, double *X, int64_t ldx,
                      double *state, void *workspace);

/* X.update(), plate half, streaming-statistics form: for every local n
 *   <x_n> = A y_n  (written to X, (K,N) row-major),
 *   S <- [sum y_n <x_n>^T ; sum <x_n><x_n>^T]  (local partial; caller all-reduces:
 *   the child->parent message sum over the sharded plate, node.py:650, dot.py:581).
 * fp64 MFMA (v_mfma_f64_16x16x4_f64), MFMA-bound. */
int32_t vmp_pca_pass(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N,
                     int32_t D, int32_t K, double *X, int64_t ldx,
                     double *state, void *workspace);

/* tau.update(): message gaussian.py:2363-2369 collapsed to traces (dot.py:355,403)
 * + gamma.py:116-148. */
int32_t vmp_pca_update_tau(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total,
                           double a0, double b0, double *state);
/* alpha.update(): gaussian.py:2361-2369 + gamma.py:116-148. */
int32_t vmp_pca_update_alpha(vmp_ctx *ctx, int32_t D, int32_t K,
                             double a0, double b0, double *state);
/* VB.loglikelihood_lowerbound (vmp.py:192-199) = sum of
 * ExponentialFamily.lower_bound_contribution (expfamily.py:400-480) over
 * Y, X, W, tau, alpha -> state[off_L .. off_L+5]. */
int32_t vmp_pca_lower_bound(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total,
                            double x_prec,
                            double a0_tau, double b0_tau,
                            double a0_alpha, double b0_alpha, double *state);

/* Elapsed milliseconds of the most recent vmp_pca_xpass / vmp_pca_pass on this context,
 * measured with HIP events on the context's stream (blocks until done);
 * enabled by vmp_ctx_set_timing(ctx, 1). */
int32_t vmp_ctx_set_timing(vmp_ctx *ctx, int32_t enabled);
int32_t vmp_pca_last_pass_ms(vmp_ctx *ctx, double *ms_pass, double *ms_reduce);

#ifdef __cplusplus
}
#endif
#endif /* VMP_HIP_H */
