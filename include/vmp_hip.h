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

/* ---- collective: plate sums over ranks ------------------------------------- *
 *
 * One process per GPU, the outermost observation plate sharded over the ranks.  Every sum
 * the reference takes over that plate -- child -> parent messages (node.py:650 ->
 * utils/misc.py:805, dot.py:581) and per-node lower-bound terms (expfamily.py:470-480) --
 * is a local partial sum followed by vmp_allreduce_sum_f64: an fp64 sum all-reduce (RCCL
 * over xGMI) enqueued on the context's stream, in place.  The communicator lives in the
 * context: rank 0 calls vmp_comm_unique_id, the caller ships the 128 bytes to the other
 * ranks by any means (MPI, a file, torch.distributed), every rank calls vmp_comm_init_rank.
 * A context without a communicator is a world of one rank (the all-reduce is the identity).
 * RCCL is loaded on first use (dlopen "librccl.so.1"); single-GPU use never touches it. */
typedef struct vmp_comm_id { char bytes[128]; } vmp_comm_id;
int32_t vmp_comm_unique_id(vmp_ctx *ctx, vmp_comm_id *id);
int32_t vmp_comm_init_rank(vmp_ctx *ctx, const vmp_comm_id *id, int32_t rank, int32_t world);
int32_t vmp_comm_destroy(vmp_ctx *ctx);
int32_t vmp_comm_info(vmp_ctx *ctx, int32_t *rank, int32_t *world);
int32_t vmp_allreduce_sum_f64(vmp_ctx *ctx, double *buf, int64_t count);

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
    int64_t off_mu;     /* D x KP   constant prior mean of W (zero unless the caller writes it;  */
                        /*          read by vmp_pca_small_ops_mean with has_mean = 1)           */
    int64_t off_mstat;  /* 2*KP : sum_d mu_dk <w_dk> | sum_d mu_dk^2 (written by the W update)   */
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

/* The replicated-node updates below (W, the replicated half of X, tau, alpha, the lower
 * bound) are latency-bound K x K work.  vmp_pca_small_ops runs a list of them, in the
 * order given (the order VB.update visits the nodes, vmp.py:154-160), inside ONE
 * single-workgroup launch; the five named entry points that follow are the one-operation
 * forms of the same kernel. */
enum vmp_pca_op {
    VMP_PCA_OP_W = 1,      /* vmp_pca_update_w     */
    VMP_PCA_OP_XPREP = 2,  /* vmp_pca_prepare_x    */
    VMP_PCA_OP_TAU = 3,    /* vmp_pca_update_tau   */
    VMP_PCA_OP_ALPHA = 4,  /* vmp_pca_update_alpha */
    VMP_PCA_OP_ELBO = 5    /* vmp_pca_lower_bound  */
};
#define VMP_PCA_MAX_OPS 8
int32_t vmp_pca_small_ops(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total, double x_prec,
                          double a0_tau, double b0_tau, double a0_alpha, double b0_alpha,
                          int32_t nops, const int32_t *ops, double *state);

/* The same with a constant NON-ZERO prior mean of W (GaussianARD(mu, alpha, ...): gaussian.py:649-670
 * phi0_d = <alpha> * mu_d + message; the Gamma message to alpha and the bound term of W see
 * <(w - mu)^2>, gaussian.py:2344-2369).  mu lies in state[off_mu] (D x KP, written by the caller
 * after vmp_pca_init_state); has_mean = 0 is vmp_pca_small_ops.  The operations then run as the
 * general single-workgroup kernel (the LDS-resident fused forms are built for mu = 0). */
int32_t vmp_pca_small_ops_mean(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_total, double x_prec,
                               double a0_tau, double b0_tau, double a0_alpha, double b0_alpha,
                               int32_t nops, const int32_t *ops, int32_t has_mean, double *state);

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
 * per-iteration collective.  fp64 MFMA (v_mfma_f64_16x16x4_f64).
 *
 * Padding: with D, K at their padded sizes and ldy, ldx >= 32 * ceil(N / 32) the ragged last
 * tile runs through the predication-free kernel as well, which also writes the pad columns
 * X[:, N .. 32*ceil(N/32)) (= A times the pad columns of Y); with smaller leading dimensions
 * only columns < N are touched.
 *
 * Ordering: in this form no other update of the iteration reads X, so the plate pass is
 * issued on the context's internal plate stream (from a private copy of A) and the call
 * returns with S queued on the main stream; the replicated-node updates of the next
 * iteration overlap the pass.  Passes are ordered among themselves.  Anything that reads
 * X outside this library on the context's stream must call vmp_pca_xjoin first
 * (vmp_ctx_sync, vmp_pca_pass and vmp_pca_stats_from_x join implicitly). */
int32_t vmp_pca_xpass(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N,
                      int32_t D, int32_t K, double *X, int64_t ldx,
                      double *state, void *workspace);

/* Make the context's stream wait for the outstanding vmp_pca_xpass (no host block). */
int32_t vmp_pca_xjoin(vmp_ctx *ctx);
/* Gram form: the messages to W of the latest latent pass, S = [G A^T ; A G A^T] in the state block
 * (dot.py:581 collapsed onto the Gram matrix), are formed lazily -- by the fused tau / alpha / bound
 * kernel when it comes next, else by this call, which every other entry point that reads S makes
 * itself (vmp_pca_small_ops*, vmp_pca_xjoin, vmp_ctx_sync).  A binding that reads the state block
 * directly (vmp_memcpy_d2h) calls it first.  No-op when nothing is pending. */
int32_t vmp_pca_ensure_gram(vmp_ctx *ctx);

/* Tile-major layout of the plate arrays.  Y is constant after Y.observe()
 * (stochastic.py:223-250), so it is re-laid-out ONCE into blocks of 32 plate elements,
 *     Yt[tile][d][j] = Y[d][32 tile + j],   tile < ceil(N/32), d < DP, j < 32,
 * zero padded in d and in the last tile.  The plate pass then streams one contiguous
 * 8*32*DP-byte span per tile (every load instruction of a wavefront covers 1 KB of consecutive
 * addresses) instead of DP row streams that lie 8*ldy bytes apart.  X may use the same layout
 * with KP rows per tile ([tile][k][j]).  vmp_pca_tiled_doubles gives the array sizes. */
int32_t vmp_pca_tiled_doubles(int32_t D, int32_t K, int64_t N, int64_t *y_doubles,
                              int64_t *x_doubles);
int32_t vmp_pca_tile_y(vmp_ctx *ctx, const double *Y, int64_t ldy, int64_t N,
                       int32_t D, int32_t K, double *Yt);
/* to_tiled != 0: Xt <- X (K rows of the row-major (K, ldx) array); else X <- Xt. */
int32_t vmp_pca_tile_x(vmp_ctx *ctx, int32_t to_tiled, double *X, int64_t ldx, int64_t N,
                       int32_t D, int32_t K, double *Xt);
/* vmp_pca_xpass on a tile-major Y.  x_tiled != 0: X is tile-major too (ldx ignored);
 * x_tiled == 0: X is row-major with KP rows and ldx >= 32*ceil(N/32) (pad rows / columns are
 * written).  Results are bit-identical to vmp_pca_xpass. */
int32_t vmp_pca_xpass_tiled(vmp_ctx *ctx, const double *Yt, int64_t N, int32_t D, int32_t K,
                            double *X, int64_t ldx, int32_t x_tiled,
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


/* ---- fused PCA block WITH MISSING VALUES ---------------------------------------------- *
 *
 * The model block of the fused PCA entry points above with Y.observe(y, mask=array)
 * (bayespy/demos/pca.py:80-82; masks: node.py:457-526, stochastic.py:223-250).  Every plate
 * n (and every row d) has its own K x K posterior; the reference materialises (1,N,K,K)
 * second moments, contracts them with einsum (dot.py:355,403,581) and loops in Python over
 * the plates for the Cholesky factorisations (utils/linalg.py:31-63).  Here X.update() runs
 * chunk by chunk over the plates, three kernels per chunk on the fp64 matrix cores
 * (bayespy_amd/csrc/vmp_mpca.hip), and only the statistics
 *     M_d = sum_n m_dn <x x^T>_n  (packed lower triangle),   r_d = sum_n m_dn y_dn <x_n>
 * survive a chunk (state[off_M]: what ranks all-reduce).  D <= 128, K <= 32.
 * Array layouts (Ymt, the two bit layouts of the mask, Xm, the scratch of a chunk) are
 * documented at the top of vmp_mpca.hip; vmp_mpca_sizes gives their sizes. */
typedef struct vmp_mpca_layout {
    int64_t DP, KP;      /* padded dims: DP in {32,64,128}, KP in {16,32}                        */
    int64_t P, PT, LR;   /* packed triangle size KP(KP+1)/2, its 16-column tiles, row length
                            LR = 16 (PT + KP/16) of [packed | K-vector] rows                      */
    int64_t off_tau;     /* 8 : a, b, <tau>, <log tau>                                            */
    int64_t off_alpha;   /* 4*KP : a, b, <alpha>, <log alpha>                                     */
    int64_t off_scal;    /* 16: [0] sum m y^2 [1] sum m [2] sum_n tr<xx>_n [3] sum_n log|Cov_n|
                            [4] plates N [5] status [6] <tau> of the last X pass [7] residual     */
    int64_t off_L;       /* 8 : L_Y, L_X, L_W, L_tau, L_alpha, L_total                            */
    int64_t off_W;       /* DP x KP      <w_d>                                                    */
    int64_t off_WW;      /* DP x KP x KP <w_d w_d^T>                                              */
    int64_t off_ldW;     /* DP           log|Cov_d|                                               */
    int64_t off_M;       /* DP x LR      packed M_d | r_d                                         */
    int64_t off_panel;   /* B operands of the precision GEMM (fragment order), current W          */
    int64_t off_panel_x; /* ... as seen by the last X.update()                                    */
    int64_t off_Sxx;     /* KP x KP      sum_n <x x^T>_n (all plates; rotations)                  */
    int64_t off_rowobs;  /* DP           observations per dimension d (summed over ranks by the
                            caller); rows with none are ignored plates of W (node.py:457-526)      */
    int64_t total;
} vmp_mpca_layout;

typedef struct vmp_mpca_sizes_t {
    int64_t ymt_doubles;        /* tile-major m*y                                                 */
    int64_t mask_words;         /* uint32 words of EACH of the two bit layouts of the mask        */
    int64_t xm_doubles;         /* <x_n>, plate-major (32 ceil(N/32)) x KP                        */
    int64_t lam_doubles;        /* scratch of one chunk: chunk x LR                               */
    int64_t xxf_doubles;        /* scratch of one chunk: packed <xx>_n                            */
    int64_t workspace_doubles;  /* per-workgroup partial statistics                               */
} vmp_mpca_sizes_t;

enum vmp_mpca_op { VMP_MPCA_OP_TAU = 1, VMP_MPCA_OP_ALPHA = 2, VMP_MPCA_OP_ELBO = 3 };

int32_t vmp_mpca_get_layout(int32_t D, int32_t K, vmp_mpca_layout *out);
int32_t vmp_mpca_sizes(vmp_ctx *ctx, int32_t D, int32_t K, int64_t N, int64_t chunk,
                       vmp_mpca_sizes_t *out);
int32_t vmp_mpca_init_state(vmp_ctx *ctx, int32_t D, int32_t K, double a0_tau, double b0_tau,
                            double a0_alpha, double b0_alpha, double *state);
/* Set-up after Y.observe(y, mask): Ymt <- m*y tile-major (values at masked entries are never
 * read: NaN placeholders are fine), the two bit layouts of the mask, sum m y^2, sum m and the
 * observation count of every dimension into the state.  mask: uint8 (D, ldm) row-major, NULL = all observed. */
int32_t vmp_mpca_prepare(vmp_ctx *ctx, const double *Y, int64_t ldy, const uint8_t *mask,
                         int64_t ldm, int64_t N, int32_t D, int32_t K, double *Ymt,
                         uint32_t *Mb1, uint32_t *Mb2, double *state, void *workspace);
/* X.update() = vmp_mpca_x_begin (snapshot of <tau>, <w>, <ww> and the plate count, summed over
 * ranks by the caller) + vmp_mpca_x_chunk over consecutive chunks of plates (n0 a multiple of
 * 32; VMP_MPCA_FIRST on the first).  VMP_MPCA_FROM_VALUE: statistics of the <x_n> already in Xm
 * (initialize_from_value, expfamily.py:193-204) instead of an update. */
int32_t vmp_mpca_x_begin(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n_plates, double *state);
#define VMP_MPCA_FIRST      1   /* first chunk of a pass: statistics are overwritten, not added to */
#define VMP_MPCA_FROM_VALUE 2   /* delta moments of the <x_n> in Xm instead of an update           */
#define VMP_MPCA_PRIOR      4   /* prior moments: <x_n> = 0, <xx>_n = I / x_prec                   */
#define VMP_MPCA_INSPECT    8   /* recompute <x_n>, <xx>_n of the chunk only (no statistics)       */
int32_t vmp_mpca_x_chunk(vmp_ctx *ctx, int32_t D, int32_t K, int64_t n0, int64_t nplates,
                         int32_t flags, double x_prec, const double *Ymt,
                         const uint32_t *Mb1, const uint32_t *Mb2, double *Xm, double *Lam,
                         double *XXf, double *state, void *workspace);
/* All chunks of one pass over the plates [0, N) (flags as above, VMP_MPCA_FIRST implied).
 * nsets = 1: Lam / XXf hold one chunk, the chunks run in order on the context's stream.
 * nsets = 2: they hold two chunks each and consecutive chunks are pipelined over three library
 * streams: the matrix-core GEMMs of the neighbouring chunks run beside the vector-ALU-bound sweep
 * of a chunk.  The statistics accumulate in chunk order either way (identical results). */
int32_t vmp_mpca_x_pass(vmp_ctx *ctx, int32_t D, int32_t K, int64_t N, int64_t chunk,
                        int32_t nsets, int32_t flags, double x_prec, const double *Ymt,
                        const uint32_t *Mb1, const uint32_t *Mb2, double *Xm, double *Lam,
                        double *XXf, double *state, void *workspace);
/* W.update() (mode 0); mode 1: delta moments of the <w_d> in state[off_W]; mode 2: prior. */
int32_t vmp_mpca_update_w(vmp_ctx *ctx, int32_t D, int32_t K, int32_t mode, double *state);
/* tau.update(), alpha.update(), the lower bound terms: a list of vmp_mpca_op, one launch. */
int32_t vmp_mpca_small_ops(vmp_ctx *ctx, int32_t D, int32_t K, double x_prec, double a0_tau,
                           double b0_tau, double a0_alpha, double b0_alpha, int32_t nops,
                           const int32_t *ops, double *state);
/* <x x^T>_n of the first nplates plates of the chunk scratch, unpacked to (nplates, K, K). */
int32_t vmp_mpca_unpack_xx(vmp_ctx *ctx, int32_t D, int32_t K, int64_t nplates,
                           const double *XXf, double *out);

/* ---- fused linear state-space model block (BASELINE.json config 5) ----------------------- *
 *
 * bayespy/demos/lssm.py:34-103 with a plate of B sequences: X = GaussianMarkovChain(mu0, Lam0,
 * A, nu, n=T, plates=(B,)), Y = GaussianARD(SumMultiply('i,i', C, X), tau) fully observed.
 * Dynamics, noise and mask are shared by the sequences, so the covariance recursion of
 * linalg.block_banded_solve (utils/linalg.py:468-575, gaussian_markov_chain.py:89-123) runs ONCE
 * (vmp_lssm_cov) and only the means are per-sequence (vmp_lssm_smooth: one thread per
 * sequence, time-major arrays Yt[t][m][b], Z[t][i][b], b contiguous); the other nodes and the
 * bound read plate sums only.  D <= 16 states (round 6: for 8 < D <= 16 the sweeps carry the state
 * only -- on the projected data tau C^T Y -- and the plate sums are formed behind them, the shared
 * covariance recursion runs on one workgroup with its blocks in LDS), M <= 64 observations per step (beyond M = 8, or 16
 * at D <= 4, the sweeps run on the projected data tau C^T y with a separate y <x>^T pass).  Details: bayespy_amd/csrc/vmp_lssm.hip, formulas: oracle/lssm.py. */
typedef struct vmp_lssm_layout {
    int64_t off_tau;      /* 4: a, b, <tau>, <log tau>                                             */
    int64_t off_gamma, off_alpha, off_nu;   /* 4*D each: a[D], b[D], mean[D], log-mean[D]          */
    int64_t off_mu0, off_Lam0, off_ldLam0;  /* D, D*D, 1: constants of the initial state           */
    int64_t off_Cm, off_CovC, off_SCC;      /* M*D <c_m>, D*D shared Cov_C, D*D sum_m <c c^T>      */
    int64_t off_Am, off_AA, off_ldA;        /* D*D <a_i>, D*D*D <a_i a_i^T>, D log|Cov_A_i|        */
    int64_t off_Dg;       /* 4*D*D: diagonal blocks of the chain precision (t=0, inner, last), E   */
    int64_t off_h0;       /* D: Lam0 mu0                                                           */
    int64_t off_covsums;  /* 5*D*D+8: output of vmp_lssm_cov (+ its state between segments)       */
    int64_t off_raw, len_raw;  /* mean-part plate sums of vmp_lssm_smooth (what ranks all-reduce)  */
    int64_t off_S;        /* Sxx | Spp | Snn | Snp | S00 (D*D each) | s0 (D) | Syx (M*D)           */
    int64_t off_scal;     /* 8: [0] sum y^2 [1] log|Phi| [2] status [3] <tau> of the last X pass
                                [4] log|Cov_C|                                                     */
    int64_t off_L;        /* 16: L_Y, L_C, L_A, L_X, L_gamma, L_alpha, L_tau, L_nu, total          */
    int64_t total;
} vmp_lssm_layout;

enum vmp_lssm_op {
    VMP_LSSM_OP_STATS = 1, VMP_LSSM_OP_C, VMP_LSSM_OP_GAMMA, VMP_LSSM_OP_XPREP, VMP_LSSM_OP_A,
    VMP_LSSM_OP_ALPHA, VMP_LSSM_OP_TAU, VMP_LSSM_OP_NU, VMP_LSSM_OP_ELBO
};

int32_t vmp_lssm_limits(int32_t *max_D, int32_t *max_M);
int32_t vmp_lssm_get_layout(int32_t D, int32_t M, vmp_lssm_layout *out);
int32_t vmp_lssm_workspace_doubles(int32_t D, int32_t M, int64_t B, int32_t T, int64_t *n);
/* Y (M, B, T) -> Yt (T, M, BL) time-major (once: Y is constant after observe); sum y^2 -> *syy. */
int32_t vmp_lssm_relayout_y(vmp_ctx *ctx, const double *Y, int32_t M, int64_t B, int32_t T,
                            int64_t BL, double *Yt, double *syy, void *workspace);
/* X (B, T, D) <-> Z (T, D, BL) */
int32_t vmp_lssm_x_layout(vmp_ctx *ctx, double *X, int32_t D, int64_t B, int32_t T, int64_t BL,
                          double *Z, int32_t to_time_major);
/* The shared D x D recursion over T.  Dg0 / Dgm / DgT: diagonal blocks of the precision at t = 0,
 * 0 < t < T-1, t = T-1; E = Phi[t, t+1].  Out: Sinv (T,D,D), J (T-1,D,D), sums (5 D^2 + 4). */
int32_t vmp_lssm_cov(vmp_ctx *ctx, int32_t T, int32_t D, const double *Dg0, const double *Dgm,
                     const double *DgT, const double *E, double *Sinv, double *J, double *sums);
/* Forward + backward vector recursions of all sequences and their plate sums (given != 0: the
 * sums of the <x> already in Z, no recursion: initialize_from_value). */
int32_t vmp_lssm_smooth(vmp_ctx *ctx, int32_t given, const double *Yt, int32_t M, int64_t B,
                        int32_t T, int64_t BL, int32_t D, const double *Cm, const double *tau,
                        const double *h0, const double *Sinv, const double *J, double *Z,
                        double *stats, void *workspace);
/* X.update() in one call = vmp_lssm_cov + vmp_lssm_smooth(given = 0), with the backward half of the
 * covariance recursion (not needed by the per-sequence passes) on a side stream beside them;
 * stream-ordered: everything is complete for later work on the context's stream. */
int32_t vmp_lssm_x_update(vmp_ctx *ctx, int32_t T, int32_t D, const double *Dg0, const double *Dgm,
                          const double *DgT, const double *E, double *Sinv, double *J,
                          double *covsums, const double *Yt, int32_t M, int64_t B, int64_t BL,
                          const double *Cm, const double *tau, const double *h0, double *Z,
                          double *stats, void *workspace);
/* <x_bt> <- R <x_bt> on the time-major means Z (T, D, BL): the state-space rotation of
 * transformations.py:1167-1176 applied to the plate-sized array; R: D x D row-major, device. */
int32_t vmp_lssm_rotate_x(vmp_ctx *ctx, int32_t D, int32_t T, int64_t B, int64_t BL, const double *R,
                          double *Z);
/* Replicated-node updates / the bound, a list of vmp_lssm_op in one launch.  priors: host array
 * of 8 doubles, the Gamma (a0, b0) of tau, gamma, alpha, nu. */
int32_t vmp_lssm_small_ops(vmp_ctx *ctx, int32_t D, int32_t M, int32_t T, double B_total,
                           const double *priors, int32_t nu_latent, int32_t nops,
                           const int32_t *ops, double *state);

/* ---- fused linear state-space model block, ARRAY masks ----------------- *
 *
 * The same model observed through ``Y.observe(y, mask=array)`` (bayespy/demos/lssm.py:132,
 * :239-246: ``mask = random.mask(M, N, p=0.3); mask[:, 30:80] = False``), the mask
 * broadcastable to (M, B, T).  A message is multiplied by the child's mask before the plate
 * sum (node.py:570-655), so
 *   - every sequence has its own block-tridiagonal precision: diagonal blocks
 *     prior + <tau> sum_m mask_mbt <c_m c_m^T> (gaussian_markov_chain.py:542-627, dot.py:425-633),
 *     hence its own D x D covariance recursion (linalg.block_banded_solve, utils/linalg.py:468-575),
 *     run by a group of FOUR LANES PER SEQUENCE (the rows of the blocks dealt over a DPP quad) in
 *     registers beside the mean recursion; vmp_tune_set("lssmm_lanes", 1): one thread per sequence
 *     (D <= 4);
 *   - every row of C has its own posterior (gaussian.py:649-706 with the per-row message
 *     sum_bt mask_mbt <x_bt x_bt^T>);
 *   - rows / sequences without any observation are ignored plates (node.py:486-526,
 *     expfamily.py:470-480).
 * Time-major arrays, b contiguous: Yt[t][m][b] (zero where masked), Mw[t][b] (uint64, bit m =
 * mask_mbt), F[t][NS + D][b] (forward sweep: S_t^-1 packed lower triangle | z_t), Z[t][D][b] (<x>),
 * P[t][NS][b] (<x x^T> packed), NS = D (D + 1) / 2.  D <= 8 states, M <= 64 observed dimensions,
 * M D^2 <= 2048 (the tables of the sweeps in LDS).
 * Details: bayespy_amd/csrc/vmp_lssmm.hip, vmp_lssmm_dev.h; formulas: oracle/lssm.py
 * (MaskedLSSMOracle).  Ops: enum vmp_lssm_op (STATS is a no-op here). */
typedef struct vmp_lssmm_layout {
    int64_t NS;           /* D (D + 1) / 2                                                          */
    int64_t off_tau;      /* 4: a, b, <tau>, <log tau>                                              */
    int64_t off_gamma, off_alpha, off_nu;   /* 4*D each: a[D], b[D], mean[D], log-mean[D]           */
    int64_t off_mu0, off_Lam0, off_ldLam0;  /* D, D*D, 1                                            */
    int64_t off_Cm;       /* M*D  <c_m>                                                             */
    int64_t off_CovC;     /* M*D*D  Cov(c_m) per row                                                */
    int64_t off_ldC;      /* M  log|Cov(c_m)|                                                       */
    int64_t off_SCC;      /* D*D  sum over the observed rows of <c_m c_m^T>                         */
    int64_t off_Am, off_AA, off_ldA;        /* D*D, D*D*D, D                                        */
    int64_t off_tab, len_tab;   /* tables of the sweeps, written by XPREP (full matrices, rows
                                   contiguous): base (3*D*D: t = 0, inner, last) | E (D*D) | h0 (D) |
                                   tau c_m (M*D) | tau <c_m c_m^T> (M*D*D)                          */
    int64_t off_setup, len_setup;  /* vmp_lssmm_prepare (summed over ranks): sum mask y^2 |
                                   sequences with data | observations per row n_m (M)              */
    int64_t off_raw, len_raw;   /* plate sums of an X pass (summed over ranks):
                                   sum_t P (NS) | sum <x_t+1 x_t^T> (D*D) | P_0 (NS) | P_T-1 (NS) |
                                   x_0 (D) | sum_b log|Phi_b| (1)   -- sequences with data only --
                                   | XX_m = sum_bt mask P (M*NS) | Syx_m = sum_bt y x (M*D)         */
    int64_t off_scal;     /* 8: [0] status [1] <tau> of the last XPREP                              */
    int64_t off_L;        /* 16: L_Y, L_C, L_A, L_X, L_gamma, L_alpha, L_tau, L_nu, total           */
    int64_t total;
} vmp_lssmm_layout;

int32_t vmp_lssmm_limits(int32_t *max_D, int32_t *max_M);
int32_t vmp_lssmm_get_layout(int32_t D, int32_t M, vmp_lssmm_layout *out);
int32_t vmp_lssmm_workspace_doubles(int32_t D, int32_t M, int64_t B, int32_t T, int64_t *n);
/* Set-up (the data are constant after observe): Y (M, B, T) row-major and the mask as bytes with
 * ELEMENT strides (sm, sb, st; 0 = broadcast axis) -> Yt (zero where masked; values there are
 * never read, NaN placeholders are fine), Mw, seqobs[b] = 1.0 if sequence b has any observation,
 * state[off_setup ...]. */
int32_t vmp_lssmm_prepare(vmp_ctx *ctx, const double *Y, const uint8_t *mask, int64_t sm,
                          int64_t sb, int64_t st, int32_t M, int64_t B, int32_t T, int64_t BL,
                          int32_t D, double *Yt, uint64_t *Mw, double *seqobs, double *state,
                          void *workspace);
/* X.update(): forward sweep (per-sequence covariance + mean recursions -> F), backward sweep
 * (-> Z, P, chain sums; for D <= 4, M <= 8 also XX_m, Syx_m), otherwise a statistics pass over the
 * stored Z, P (XX_m, Syx_m); the sums land in state[off_raw ...].
 * given = 1: the sums of the <x> already in Z as point masses (initialize_from_value);
 * given = 2: q(X) unchanged, only XX_m / Syx_m again from the stored Z, P (Y re-observed). */
int32_t vmp_lssmm_x_update(vmp_ctx *ctx, int32_t given, const double *Yt, const uint64_t *Mw,
                           const double *seqobs, int32_t M, int64_t B, int32_t T, int64_t BL,
                           int32_t D, double *state, double *F, double *Z, double *P,
                           void *workspace);
/* <x x^T> <- R <x x^T> R^T on the packed plate array P (T, NS, BL): the state rotation of
 * inference/transformations.py (reference: gaussian_markov_chain.py rotate, gaussian.py:1693-1741)
 * applied on the device; the means go through vmp_lssm_rotate_x. */
int32_t vmp_lssmm_rotate_p(vmp_ctx *ctx, int32_t D, int32_t T, int64_t B, int64_t BL, const double *R,
                           double *P);
/* Replicated-node updates / the bound: a list of vmp_lssm_op in one launch. */
int32_t vmp_lssmm_small_ops(vmp_ctx *ctx, int32_t D, int32_t M, int32_t T, const double *priors,
                            int32_t nu_latent, int32_t nops, const int32_t *ops, double *state);

/* ---- fused full-covariance Gaussian-mixture block -------------------------- *
 *
 * Model block  Y = Mixture(z, Gaussian, mu, Lambda), z = Categorical(alpha),
 * alpha = Dirichlet(a0), mu = GaussianARD(0, beta0, shape=(D,), plates=(K,)),
 * Lambda = Wishart(n0, V0, plates=(K,))   (bayespy/demos/mog.py:17-64), Y fully
 * observed, D <= 32, K <= 64.  One VB iteration reads Y exactly once and writes the
 * responsibilities once.  Y is (N, D) row-major (the reference's layout), the
 * responsibilities R are (N, K) row-major.
 */
typedef struct vmp_gmm_layout {
    int64_t DP, KP;        /* padded D (4 or 8) and K (16, 32, 64)                          */
    int64_t FS;            /* 1 + D + D*D : row length of the statistics                      */
    int64_t FP, F2P;       /* padded feature count [y_a y_b (a<=b), y_d, 1] of both MFMA phases */
    int64_t off_T, len_T;  /* KP x FS : per cluster [R_k, sum_n r y (D), sum_n r y y^T (D*D)] */
                           /*   -- the plate sums of mixture.py:126-158 + node.py:650          */
    int64_t off_zs;        /* sum_n logsumexp(phi_n), sum_nk r phi   (for L_z)                */
    int64_t off_alpha;     /* alpha[KP], <log pi>[KP]                  (dirichlet.py:150-158) */
    int64_t off_mu;        /* KP x D   <mu_k>                                                 */
    int64_t off_Cmu;       /* KP x D x D  Cov(mu_k)                                           */
    int64_t off_logdetLmu; /* KP  log|Lambda_mu_k|                                            */
    int64_t off_nk;        /* KP  Wishart degrees of freedom                                  */
    int64_t off_Vk;        /* KP x D x D  Wishart inverse scale                               */
    int64_t off_Lam;       /* KP x D x D  <Lambda_k> = n_k V_k^-1        (wishart.py:184)     */
    int64_t off_logdetLam; /* KP  <log|Lambda_k|>                          (wishart.py:185)   */
    int64_t off_logdetV;   /* KP  log|V_k|                                                    */
    int64_t off_C;         /* KP x F2P coefficients of ell_nk in the compact feature order     */
    int64_t off_prior;     /* alpha0[KP], beta0, n0, log|V0|, 5 pad, V0[D*D]                  */
    int64_t off_scal;      /* 8 : [3] status                                                  */
    int64_t off_L;         /* 8 : L_Y, L_z, L_alpha, L_mu, L_Lambda, L_total                  */
    int64_t total;
} vmp_gmm_layout;

int32_t vmp_gmm_get_layout(int32_t D, int32_t K, vmp_gmm_layout *out);
int32_t vmp_gmm_workspace_bytes(vmp_ctx *ctx, int32_t D, int32_t K, size_t *bytes);
/* Store the priors (host arrays alpha0[K], V0[D*D]) and initialise every node from its
 * prior (expfamily.py:168-184). */
int32_t vmp_gmm_init_state(vmp_ctx *ctx, int32_t D, int32_t K, const double *alpha0_host,
                           double beta0, double n0, const double *V0_host, double *state);
/* z.initialize_from_value(labels): one-hot responsibilities (categorical.py:30-46,
 * bit-exact) written to R and their statistics to T. */
int32_t vmp_gmm_stats_from_labels(vmp_ctx *ctx, const double *Y, int64_t N, int32_t D, int32_t K,
                                  const int64_t *labels, double *R, double *state,
                                  void *workspace);
/* mu.update(): gaussian.py:649-706 with the mixture-weighted messages of
 * gaussian.py:2451-2454 / mixture.py:126-158. */
int32_t vmp_gmm_update_mu(vmp_ctx *ctx, int32_t D, int32_t K, double *state);
/* Lambda.update(): wishart.py:153-188 with the messages gaussian.py:2516-2520. */
int32_t vmp_gmm_update_lambda(vmp_ctx *ctx, int32_t D, int32_t K, double *state);
/* z.update(), replicated half: coefficients of E[log p(y_n | k)] + <log pi_k>
 * (mixture.py:67-104, expfamily.py:45-61).  prior_only != 0 keeps only <log pi_k>
 * (q(z) initialised from its prior). */
int32_t vmp_gmm_prepare_z(vmp_ctx *ctx, int32_t D, int32_t K, int32_t prior_only, double *state);
/* z.update(), plate half -- THE pass: phi = C feat(y), r = normalized_exp(phi)
 * (multinomial.py:114-120, utils/misc.py:1388-1401) written to R, and the statistics
 * T = r^T [1, y, y y^T] (local partial; the caller all-reduces T and zs). fp64 MFMA. */
int32_t vmp_gmm_pass(vmp_ctx *ctx, const double *Y, int64_t N, int32_t D, int32_t K, double *R,
                     double *state, void *workspace);
/* alpha.update(): dirichlet.py:113-160 with the message categorical -> [r]. */
int32_t vmp_gmm_update_alpha(vmp_ctx *ctx, int32_t D, int32_t K, double *state);
/* Lower bound of Y, z, alpha, mu, Lambda (expfamily.py:400-480). */
int32_t vmp_gmm_lower_bound(vmp_ctx *ctx, int32_t D, int32_t K, double *state);

/* ---- generic plate-broadcast kernels -------------------------------------- *
 *
 * Arrays are fp64 with explicit ELEMENT strides per axis; stride 0 marks a
 * broadcast axis (the reference's unit / missing plate axes).
 */
#define VMP_MAX_DIMS          8
#define VMP_MAX_OPERANDS      6
#define VMP_EWISE_MAX_OPS     48
#define VMP_EWISE_MAX_CONSTS  8

/* out[kept axes] = scale * sum_{axes in reduce_mask} prod_i in_i[...]
 * -- misc.sum_multiply (utils/misc.py:851-933, np.einsum call site :906) and the
 * plate sum of Node._message_to_parent via misc.sum_multiply_to_plates
 * (node.py:650, utils/misc.py:805-844); `scale` carries broadcasting_multiplier
 * (utils/misc.py:761-802).  in_strides is [nin][ndim] row-major; out_strides[ndim]
 * (ignored on reduced axes).  Deterministic (fixed-order) reduction. */
int32_t vmp_sum_multiply(vmp_ctx *ctx, int32_t ndim, const int64_t *shape, int32_t nin,
                         const double *const *in, const int64_t *in_strides,
                         const int64_t *out_strides, uint32_t reduce_mask, double scale,
                         double *out, void *workspace, size_t workspace_bytes);
size_t vmp_sum_multiply_workspace_bytes(void);

/* Fused broadcast elementwise formula: a postfix program (opcode | operand<<8)
 * over <= VMP_MAX_OPERANDS inputs evaluated per output element with a 4-deep
 * stack; `out` is contiguous with the given shape.  Replaces the NumPy ufunc
 * chains inside the Distribution formulas (e.g. gaussian.py:675-678, :2344-2369,
 * gamma.py:142-148, dirichlet.py:150-158, expfamily.py:449-468). */
enum {
    VMP_OP_IN = 0, VMP_OP_CONST, VMP_OP_ADD, VMP_OP_SUB, VMP_OP_MUL, VMP_OP_DIV, VMP_OP_NEG,
    VMP_OP_LOG, VMP_OP_EXP, VMP_OP_SQR, VMP_OP_SQRT, VMP_OP_RECIP, VMP_OP_DIGAMMA,
    VMP_OP_LGAMMA, VMP_OP_MAX, VMP_OP_MIN, VMP_OP_WHERE_NZ, VMP_OP_DUP, VMP_OP_SWAP,
    VMP_OP_TRIGAMMA,
    VMP_OP__COUNT
};
int32_t vmp_ewise(vmp_ctx *ctx, int32_t ndim, const int64_t *shape, int32_t nin,
                  const double *const *in, const int64_t *in_strides, int32_t nops,
                  const int32_t *ops, int32_t nconsts, const double *consts, double *out);

/* A sequence of calls recorded into a HIP graph and replayed: the launches of a VB sweep are the
 * same every iteration -- same kernels, same shapes, only the contents of the arrays move -- so a
 * binding records one sweep (every call of this library between begin and end is recorded on the
 * context's stream instead of run; the arrays must be allocated before, nothing may be read on
 * the host inside) and launches the graph once per iteration afterwards.  The context needs a
 * stream of its own (vmp_ctx_create / vmp_ctx_set_stream: not the legacy default stream).  A binding
 * that manages its own arrays uses these four.  The Python front end (plans/graph_iter.py) cannot:
 * its sweeps ALLOCATE -- every array operation of the engine returns a fresh device array -- and an
 * allocation inside a stream capture must come from a pool that lives as long as the graph, which
 * is what torch.cuda.graph provides (torch's caching allocator is the allocator of this package:
 * DESIGN.md section 1).  It therefore opens the capture through torch and issues the SAME library
 * calls on the capturing stream; vmp_copy_many / vmp_pack_outputs / vmp_queue_commit are the pieces
 * of such a recording that live behind this boundary. */
int32_t vmp_graph_begin(vmp_ctx *ctx);
int32_t vmp_graph_end(vmp_ctx *ctx, void **graph);
int32_t vmp_graph_launch(vmp_ctx *ctx, void *graph);
int32_t vmp_graph_destroy(vmp_ctx *ctx, void *graph);

/* n device-to-device copies of `count[i]` doubles each (contiguous, non-overlapping) as ONE launch
 * per 24 of them on the context's stream -- the copy-back of the state arrays of a recorded sweep
 * (graph_iter.py; the reference has no analogue: it rebinds NumPy arrays, stochastic.py:223-273). */
int32_t vmp_copy_many(vmp_ctx *ctx, int32_t n, const double *const *src, double *const *dst,
                      const int64_t *count);

/* out[i] for i < n: kind[i] = 0: the double at src[i]; 1 / 2: 1.0 if any of the count[i] int32 /
 * double values at src[i] is non-zero (NaN included), else 0.0 -- the lower-bound terms and the
 * validity flags of a recorded sweep (expfamily.py:400-480; "Matrix not positive definite",
 * utils/linalg.py:58-59) gathered by ONE launch for the single device-to-host read of a replay. */
int32_t vmp_pack_outputs(vmp_ctx *ctx, int32_t n, const void *const *src, const int64_t *count,
                         const int32_t *kind, double *out);

/* Queue of SMALL operations.  Between vmp_queue_begin and vmp_queue_end, vmp_ewise and
 * vmp_sum_multiply calls on small arrays (<= 2048 outputs, <= 32768 products) and vmp_spd_batched
 * calls on a few small matrices (8 < n <= 32, batch <= 4) are recorded on the host and run, in
 * order, by ONE launch of an interpreter kernel -- when any other entry point of the generic part
 * of this library needs the stream (it flushes first), when 128 of them are collected, at
 * vmp_queue_flush or at the outermost vmp_queue_end.  A VB sweep of the generic engine
 * is ~90 such operations on scalars and K x K arrays (the Gamma / ARD formulas of gamma.py:142-148,
 * gaussian.py:2344-2369, the bound terms of expfamily.py:449-468) between a dozen plate-sized
 * kernels.  The interpreter keeps the small arrays of a launch in LDS: results are written to
 * memory AND to a 112 KB arena, a record reads what an earlier record of the launch produced --
 * or a small array from outside, copied in when the launch starts -- from there, and only a record
 * that reads MEMORY written earlier in the launch waits for those stores; the records themselves are
 * staged through LDS.  ("small_queue_lds" = 0: every operand from memory, a fence per record.)
 * Results do not depend on the grouping.  The caller must flush before it reads an
 * output on the host or hands it to work outside this library.  begin / end nest; a context whose
 * stream records a HIP graph keeps the records of its flushes for the life of the context; their
 * device copies are made by vmp_queue_commit, which belongs between the end of the recording and
 * the first launch of the graph (vmp_graph_end calls it; a binding that records through another
 * API calls it itself).  vmp_tune_set("small_queue", 0) makes begin / end no-ops;
 * "small_queue_ew" / "small_queue_sm" = 0 keep formulas / sums and inverses out of the queue. */
int32_t vmp_queue_begin(vmp_ctx *ctx);
int32_t vmp_queue_flush(vmp_ctx *ctx);
int32_t vmp_queue_end(vmp_ctx *ctx);
int32_t vmp_queue_commit(vmp_ctx *ctx);
int32_t vmp_queue_stats(vmp_ctx *ctx, int64_t *launches, int64_t *ops);

/* Batched SPD inverse and log-determinant of `batch` contiguous n x n matrices
 * (n <= 64): linalg.chol + chol_inv + chol_logdet (utils/linalg.py:31-223), one
 * workgroup / wavefront per matrix instead of a Python loop over plates.
 * Ainv / logdet may be NULL; info[b] = 1 where a matrix is not positive definite
 * ("Matrix not positive definite", utils/linalg.py:58-59). */
int32_t vmp_spd_batched(vmp_ctx *ctx, int32_t n, int64_t batch, const double *A, double *Ainv,
                        double *logdet, int32_t *info);

/* Moments and log-normaliser of a Gaussian with a full covariance per plate from its natural
 * parameters, one pass (GaussianARDDistribution.compute_moments_and_cgf, gaussian.py:680-706,
 * i.e. linalg.chol + chol_inv + chol_solve + outer + chol_logdet fused):
 *   Cov = (-2 phi1)^-1,  u0 = Cov phi0,  u1 = u0 u0^T + Cov,
 *   g = -1/2 u0 . phi0 + 1/2 log|-2 phi1|.
 * phi0, u0: batch x n; phi1, u1: batch x n x n; g: batch; info[b] = 1 where -2 phi1 is not
 * positive definite.  Built for 8 < n <= 32 (row-per-lane kernel); other sizes go through
 * vmp_spd_batched + vmp_sum_multiply. */
int32_t vmp_gaussian_moments(vmp_ctx *ctx, int32_t n, int64_t batch, const double *phi0,
                             const double *phi1, double *u0, double *u1, double *g,
                             int32_t *info);

/* ONE pass for the update of a Gaussian node whose posterior covariance is SHARED by its N plates
 * (a plate-free precision: GaussianARD / Gaussian under a scalar mask) -- the natural parameters
 * from the prior and the message (GaussianARDDistribution.compute_phi_from_parents,
 * gaussian.py:649-670), the posterior means (compute_moments_and_cgf, gaussian.py:672-706:
 * <x> = Cov phi0), the Dot / SumMultiply message the node receives when it is given as its
 * operands (dot.py:581: m_n = B^T y_n), and the plate sums its neighbours ask for next (dot.py:581
 * for the other parent, expfamily.py:449-468):
 *     m_n = B^T y_n                 Y form: Y (D x N, element strides y_sd, y_sn, one of them 1),
 *                                   B (D x K, strides b_sd, b_sk); m0 = NULL
 *         | m0[n, :]                message form: rows of a given array (strides m0_sn, m0_sk); Y = NULL
 *     x_n = Cov (p0 + m_n)          p0: K doubles or NULL (= 0); Cov: K x K contiguous
 *     stats = [ sum_n x_n (K) ; sum_n x_n x_n^T (K x K) ; sum_n y_n x_n^T (D x K; Y form only) ]
 * x: N x K with element strides (x_sn, x_sk).  Y form: fp64 MFMA tile kernel (8 N (D + K) bytes of
 * algorithmic traffic, 4 N D K + 2 N K^2 flops), D <= 256; message form: 16 N K bytes.  K <= 64.
 * Partial sums are combined in fixed order.  `workspace`: at least
 * vmp_gaussian_shared_update_workspace_bytes(D, K) bytes (D = 0 for the message form). */
size_t vmp_gaussian_shared_update_workspace_bytes(int32_t D, int32_t K);
int32_t vmp_gaussian_shared_update(vmp_ctx *ctx, int64_t N, int32_t K, int32_t D,
                                   const double *Y, int64_t y_sd, int64_t y_sn, const double *B,
                                   int64_t b_sd, int64_t b_sk, const double *m0, int64_t m0_sn,
                                   int64_t m0_sk, const double *p0, const double *cov, double *x,
                                   int64_t x_sn, int64_t x_sk, double *stats, void *workspace,
                                   size_t workspace_bytes);

/* Row softmax moments of Multinomial/Categorical: p = normalized_exp(phi),
 * lse = logsumexp(phi) (multinomial.py:114-120, utils/misc.py:1366-1401). */
int32_t vmp_softmax_moments(vmp_ctx *ctx, int64_t rows, int32_t K, const double *phi, double *p,
                            double *lse);

/* One-hot fixed moments of Categorical (categorical.py:30-46, :93-114); integer
 * indexing, bit-exact.  *info != 0 when a label is outside [0, K). */
int32_t vmp_onehot_i64(vmp_ctx *ctx, int64_t n, int32_t K, const int64_t *labels, double *out,
                       int32_t *info);

/* Plate re-indexing of the deterministic nodes that move plates.  Arrays are contiguous and
 * seen as (outer, axis, inner).
 *   vmp_take_axis:  dst[o, dst_off + j, i] = src[o, idx ? idx[j] : j, i]  for j < n; dst has
 *     dst_len rows on the axis.  np.take on a plate axis (Take._compute_moments, take.py:72-81);
 *     with idx = NULL the block copy of Concatenate._compute_moments (concatenate.py:130-167).
 *     idx is a device array of n valid row numbers in [0, src_len).
 *   vmp_segment_sum_axis:  dst[o, l, i] = sum over t in [ptr[l], ptr[l+1]) of
 *     src[o, perm[t], i]  for l < out_len -- misc.put_simple (utils/misc.py:549-585), the
 *     accumulating inverse of take used by Take._compute_message_to_parent (take.py:83-94).
 *     ptr (out_len + 1) and perm (src_len) are device arrays: the CSR map from target row to
 *     its source rows, built once on the host; the order of the additions is fixed, so the
 *     result is bit-reproducible. */
int32_t vmp_take_axis(vmp_ctx *ctx, int64_t outer, int64_t src_len, int64_t inner,
                      const double *src, int64_t n, const int64_t *idx, double *dst,
                      int64_t dst_len, int64_t dst_off);
int32_t vmp_segment_sum_axis(vmp_ctx *ctx, int64_t outer, int64_t src_len, int64_t inner,
                             const double *src, int64_t out_len, const int64_t *ptr,
                             const int64_t *perm, double *dst);

/* Strided batched fp64 contraction on the matrix cores (v_mfma_f64_16x16x4_f64):
 *     C[b, m, n] = scale * sum_k A[b, m, k] * B[b, k, n]
 * with arbitrary ELEMENT strides on every axis (0 = broadcast batch axis), up to three
 * batch axes, split-K with fixed-order combination.  The dense N x K x D contractions
 * that SumMultiply / sum_multiply collapse to (dot.py:355, :403, :581 with array masks;
 * mixture.py:126-158; gaussian_markov_chain.py:462-475) instead of
 * np.einsum(optimize=False) (utils/misc.py:906).  `workspace` holds split-K partials. */
int32_t vmp_gemm_strided(vmp_ctx *ctx, int32_t nbatch_dims, const int64_t *bshape, int64_t M,
                         int64_t N, int64_t K, const double *A, const int64_t *a_bstride,
                         int64_t a_ms, int64_t a_ks, const double *B, const int64_t *b_bstride,
                         int64_t b_ks, int64_t b_ns, double *C, const int64_t *c_bstride,
                         int64_t c_ms, int64_t c_ns, double scale, void *workspace,
                         size_t workspace_bytes);

/* Forward-backward recursion of `nchains` categorical Markov chains with K states and N
 * transitions (N + 1 time instances): random.alpha_beta_recursion (utils/random.py:357-422),
 * the moments of CategoricalMarkovChain (categorical_markov_chain.py:107-117).
 *   logp0: K unnormalised log-probabilities of the first state per chain (chain stride
 *          p0_bstride elements, 0 = shared);
 *   logP:  N x K x K slices per chain, logP[n, i, j] = log p(z_{n+1} = j | z_n = i) + evidence
 *          of instance n + 1, each slice contiguous (chain stride P_bstride, time stride
 *          P_tstride elements; 0 = shared / time-invariant).
 * Out: z0 (nchains x K) = q(z_0), zz (nchains x N x K x K) = q(z_n, z_{n+1}), g (nchains) = minus
 * the log-normaliser.  workspace: nchains * N * K doubles.  K <= 64. */
int32_t vmp_alpha_beta_recursion(vmp_ctx *ctx, int32_t N, int32_t K, int64_t nchains,
                                 const double *logp0, int64_t p0_bstride, const double *logP,
                                 int64_t P_bstride, int64_t P_tstride, double *z0, double *zz,
                                 double *g, void *workspace, size_t workspace_bytes);

/* Block-tridiagonal SPD solve = Kalman filter + RTS smoother of the Gaussian Markov chain
 * (linalg.block_banded_solve, utils/linalg.py:468-575, called by
 * gaussian_markov_chain.py:89-123).  A: nm x T x K x K diagonal blocks, B: nm x (T-1) x K x K
 * super-diagonal blocks, y: ny x T x K right-hand sides; nm == 1 (shared dynamics) or
 * nm == ny.  Out: V (diagonal blocks of the inverse), C (super-diagonal blocks), x
 * (solutions), ldet[nm] (log-determinant); info[b] = 1 where a block is not positive
 * definite.  K <= 16 (a wavefront per matrix sequence up to K = 8, a workgroup above). */
int32_t vmp_block_banded_solve(vmp_ctx *ctx, int32_t T, int32_t K, int64_t nm, int64_t ny,
                               const double *A, const double *B, const double *y, double *V,
                               double *C, double *x, double *ldet, int32_t *info);

/* Measurement knob: overrides a launch parameter the library otherwise takes from its
 * environment variable / default ("xpass_nt", "xpass_wgs_per_cu", "xpass_occ",
 * "plate_stream", ...); process-wide, for A/B harnesses (tools/xpass_lab.hip). */
int32_t vmp_tune_set(const char *key, int32_t value);

/* Elapsed milliseconds of the most recent vmp_pca_xpass / vmp_pca_pass on this context,
 * measured with HIP events on the stream the pass kernel was launched on (blocks until done);
 * enabled by vmp_ctx_set_timing(ctx, 1). */
int32_t vmp_ctx_set_timing(vmp_ctx *ctx, int32_t enabled);
int32_t vmp_pca_last_pass_ms(vmp_ctx *ctx, double *ms_pass, double *ms_reduce);

/* Durations of the (up to 64, up to `cap`) most recent timed plate passes (PCA or GMM) in issue
 * order, then forget them: one HIP event triple per pass, so a measurement loop reads them
 * after its timed region instead of blocking inside every iteration. */
int32_t vmp_pass_times_ms(vmp_ctx *ctx, double *ms_pass, double *ms_reduce, int32_t cap,
                          int32_t *count);

#ifdef __cplusplus
}
#endif
#endif /* VMP_HIP_H */
