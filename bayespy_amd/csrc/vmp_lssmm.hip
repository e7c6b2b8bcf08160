// vmp_lssmm.hip -- fused linear state-space model block with ARRAY masks (gfx950).
//
// Model: bayespy/demos/lssm.py:34-103 (optionally with a plate of B sequences) observed through
// ``Y.observe(y, mask=mask)`` with an array mask broadcastable to (M, B, T) -- the reference's own
// canonical use (demos/lssm.py:239-246).  With a mask the block-tridiagonal precision of q(X_b)
// differs per sequence (diagonal blocks prior + <tau> sum_m mask_mbt <c_m c_m^T>), so unlike
// vmp_lssm.hip there is no shared covariance recursion: ONE THREAD PER SEQUENCE runs the D x D
// block LDL^T recursion of linalg.block_banded_solve (utils/linalg.py:468-575) in registers beside
// the mean recursion.  Sweeps over time-major arrays (b contiguous, every access of a time step
// coalesced over the sequences):
//   lssmm_forward_kernel    reads Yt, Mw; writes F = [S_t^-1 (packed) | z_t]
//   lssmm_backward_kernel   reads F; writes Z = <x_t>, P = <x_t x_t^T> (packed); chain sums
//   lssmm_stats_kernel      reads P, Z, Yt, Mw; XX_m = sum_bt mask P, Syx_m = sum_bt y x
// Algorithmic bytes per (sequence, step): 8 (M + D + NS) read+write of the data, means and second
// moments; the three sweeps move 8 (2M + 3 NS + 3 D) + 16 (mask word twice): F is written and read
// back (the backward recursion needs S_t^-1 of every step).
// The per-thread arithmetic lives in vmp_lssmm_dev.h (host+device; the CPU suite runs the same
// text against oracle/lssm.py).  All plate sums are combined in a fixed order.
#include "vmp_common.h"
#include "vmp_lssmm_dev.h"

namespace {

constexpr int WNT = 64;       // one wavefront per workgroup: few sequences must spread over many CUs
constexpr int RNT = 256;
constexpr int MG1 = 8;        // rows of C per thread of the statistics pass, one thread per sequence (B = 1e5,
                              // M = 8: ONE pass over P, Z, Y at 468 registers, 3.6 ms, instead of two at 232, 5.15 ms)
constexpr int MG4 = 8;        // rows of C per lane group of the statistics pass, D <= 4
constexpr int MG8 = 4;        // ... D > 4 (two rows of the matrices per lane)

struct dg_fn {
    __host__ __device__ double operator()(double x) const { return vmp_digamma(x); }
};
struct lg_fn {
    __host__ __device__ double operator()(double x) const { return vmp_lgamma(x); }
};

// ---------------------------------------------------------------------------------------------
// set-up: Y (M, B, T) + mask (strided bytes) -> Yt (T, M, BL), Mw (T, BL), counts
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RNT)
lssmm_prepare_kernel(const double *__restrict__ Y, const uint8_t *__restrict__ mask, int64_t sm,
                     int64_t sb, int64_t st, int M, int64_t B, int T, int64_t BL,
                     double *__restrict__ Yt, unsigned long long *__restrict__ Mw,
                     double *__restrict__ partial, unsigned long long *__restrict__ cnt)
{
    // 32 x 32 tiles of the (b, t) plane through LDS: both sides coalesced
    __shared__ double tile[32][33];
    __shared__ unsigned char mt[32][33];
    __shared__ double red[RNT / 64];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const int64_t nbt = (B + 31) / 32, ntt = (T + 31) / 32;
    double syy = 0.0;
    for (int64_t blk = blockIdx.x; blk < nbt * ntt * M; blk += gridDim.x) {
        const int m = (int)(blk / (nbt * ntt));
        const int64_t r = blk - (int64_t)m * nbt * ntt;
        const int64_t bt = r / ntt, tt = r - bt * ntt;
        __syncthreads();
        unsigned long long c = 0;
        for (int j = ty; j < 32; j += 8) {
            const int64_t b = bt * 32 + j;
            const int t = (int)(tt * 32) + tx;
            double v = 0.0;
            unsigned char mk = 0;
            if (b < B && t < T) {
                mk = mask[m * sm + b * sb + (int64_t)t * st] ? 1 : 0;
                if (mk) v = Y[((int64_t)m * B + b) * T + t];     // values at masked entries: never read
            }
            tile[j][tx] = v;
            mt[j][tx] = mk;
            syy += v * v;
            c += mk;
        }
        // integer count of this row's observations (order-independent: exact)
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt[m], c);
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int t = (int)(tt * 32) + j;
            const int64_t b = bt * 32 + tx;
            if (t < T && b < BL) {
                Yt[((int64_t)t * M + m) * BL + b] = tile[tx][j];
                if (mt[tx][j]) atomicOr(&Mw[(int64_t)t * BL + b], 1ull << m);
            }
        }
    }
    syy = block_sum<RNT>(syy, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = syy;
}

// seqobs[b] = 1.0 when sequence b has any observation (pad columns 0); their count
__global__ void __launch_bounds__(RNT)
lssmm_seqobs_kernel(const unsigned long long *__restrict__ Mw, int64_t B, int T, int64_t BL,
                    double *__restrict__ seqobs, unsigned long long *__restrict__ cnt_b)
{
    const int64_t b = (int64_t)blockIdx.x * RNT + threadIdx.x;
    unsigned long long any = 0;
    if (b < B)
        for (int t = 0; t < T; ++t) any |= Mw[(int64_t)t * BL + b];
    if (b < BL) seqobs[b] = any ? 1.0 : 0.0;
    unsigned long long c = any ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(cnt_b, c);
}

__global__ void __launch_bounds__(64)
lssmm_setup_finish_kernel(const double *__restrict__ partial, int n,
                          const unsigned long long *__restrict__ cnt, int M, double *__restrict__ out)
{
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += partial[i];
        out[0] = s;
        out[1] = (double)cnt[M];
    }
    for (int m = threadIdx.x; m < M; m += 64) out[2 + m] = (double)cnt[m];
}

// ---------------------------------------------------------------------------------------------
// sweeps: G lanes per sequence (vmp_lssmm_dev.h), one wavefront per workgroup
// ---------------------------------------------------------------------------------------------
struct sweep_args {
    lssmm_seq_args S;
    const double *seqobs;
    int64_t B;
    int tab_len;
    int nib_len;            // doubles of the nibble tables of the forward sweep (0: none)
    int pstride;            // doubles per workgroup in ``partial``
    double *partial;        // per workgroup
    double *status;         // state[off_scal]
    int given;
};

// sum over the sequences of the wavefront, separately for every lane position of the group
template <int G>
__device__ __forceinline__ double seqs_sum(double v)
{
#pragma unroll
    for (int off = G; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int D, int G, int MB>
__global__ void __launch_bounds__(WNT)
lssmm_forward_kernel(sweep_args A)
{
    extern __shared__ double tab[];
    for (int e = threadIdx.x; e < A.tab_len; e += WNT) tab[e] = A.S.tab[e];
    __syncthreads();
    lssmm_seq_args S = A.S;
    S.tab = tab;
    // sums of the observation blocks by mask nibble, behind the tables (when they fit)
    S.nib = nullptr;
    if (A.nib_len > 0) {
        lssmm_build_nibbles(D, S.M, tab, tab + A.tab_len, (int)threadIdx.x, WNT);
        __syncthreads();
        S.nib = tab + A.tab_len;
    }
    const int64_t tid = (int64_t)blockIdx.x * WNT + threadIdx.x;
    const int64_t bq = tid / G;
    const int lane = (int)(tid % G);
    // the tail of the last wavefront works on the zero-filled padding columns b in [B, BL)
    const bool live = bq < A.B;
    const int64_t b = bq;
    int bad = 0;
    double ld = lssmm_forward_seq<D, G, MB>(S, b, lane, bad);
    // log|Phi_b| of the sequences with data (an ignored plate adds nothing to the bound)
    ld = (live && lane == 0) ? ld * A.seqobs[b] : 0.0;
    ld = wave_sum(ld);
    if (threadIdx.x == 0) A.partial[blockIdx.x] = ld;
    if (bad && live) A.status[0] = (double)VMP_ERR_NOT_POSDEF;
}

struct put_store {
    double *out;
    __device__ void operator()(int slot, double v) const { out[slot] = v; }
};

// partial[blk]: chain sums (CL) and, MF > 0, the statistics XX (M, NS) | Syx (M, D) behind them
template <int D, int G, int MF, bool GIVEN>
__global__ void __launch_bounds__(WNT)
lssmm_backward_kernel(sweep_args A)
{
    using AC = lssmm_acc<D, G, MF>;
    constexpr int NS = D * (D + 1) / 2;
    constexpr int CL = 3 * NS + D * D + D;        // chain sums without log|Phi|
    extern __shared__ double tab[];
    for (int e = threadIdx.x; e < A.tab_len; e += WNT) tab[e] = A.S.tab[e];
    __syncthreads();
    lssmm_seq_args S = A.S;
    S.tab = tab;
    const int64_t tid = (int64_t)blockIdx.x * WNT + threadIdx.x;
    const int64_t bq = tid / G;
    const int lane = (int)(tid % G);
    const bool live = bq < A.B;
    const int64_t b = bq;
    double acc[AC::len];
    lssmm_backward_seq<D, G, MF, GIVEN>(S, b, lane, acc);
    // chain sums: sequences with data only; the statistics carry the mask themselves
    const double wc = live ? A.seqobs[b] : 0.0;
#pragma unroll
    for (int e = 0; e < AC::len; ++e) {
        const double v = e < AC::chain ? acc[e] * wc : (live ? acc[e] : 0.0);
        acc[e] = seqs_sum<G>(v);
    }
    if (threadIdx.x < G) {
        double *out = A.partial + (int64_t)blockIdx.x * A.pstride;
        lssmm_put_chain<D, G>(lane, acc, put_store{out});
        if (MF > 0)
            lssmm_put_stats<D, G, (MF > 0 ? MF : 1)>(lane, 0, S.M, acc + AC::XX, acc + AC::Syx,
                                                    put_store{out + CL});
    }
}

// partial[blk]: XX (M, NS) | Syx (M, D); blockIdx.y = the group of MG rows of C
template <int D, int G, int MG>
__global__ void __launch_bounds__(WNT)
lssmm_stats_kernel(sweep_args A)
{
    constexpr int R = (D + G - 1) / G;
    constexpr int AL = MG * R * (D + 1);
    const int64_t tid = (int64_t)blockIdx.x * WNT + threadIdx.x;
    const int64_t bq = tid / G;
    const int lane = (int)(tid % G);
    const bool live = bq < A.B;
    const int64_t b = bq;
    const int m0 = blockIdx.y * MG;
    double acc[AL];
    lssmm_stats_seq<D, G, MG>(A.S, b, lane, m0, acc);
#pragma unroll
    for (int e = 0; e < AL; ++e) acc[e] = seqs_sum<G>(live ? acc[e] : 0.0);
    if (threadIdx.x < G)
        lssmm_put_stats<D, G, MG>(lane, m0, A.S.M, acc, acc + MG * R * D,
                                  put_store{A.partial + (int64_t)blockIdx.x * A.pstride});
}

// out[j] = sum_blk partial[blk * stride + j], fixed order; 16 row-lanes x 16 outputs per workgroup
__global__ void __launch_bounds__(RNT)
lssmm_sum_kernel(const double *__restrict__ partial, int n, int stride, int len,
                 double *__restrict__ out)
{
    __shared__ double tile[16][17];
    const int kx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + kx;
    double acc = 0.0;
    if (j < len)
        for (int b = ry; b < n; b += 16) acc += partial[(int64_t)b * stride + j];
    tile[ry][kx] = acc;
    __syncthreads();
    if (ry == 0 && j < len) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += tile[r][kx];
        out[j] = s;
    }
}

// P[t][.][b] <- R P R^T: one thread per (step, sequence), R through scalar loads
template <int D>
__global__ void __launch_bounds__(RNT)
lssmm_rotate_p_kernel(const double *__restrict__ R, int T, int64_t B, int64_t BL, double *__restrict__ P)
{
    constexpr int NS = D * (D + 1) / 2;
    __shared__ double Rs[D * D];
    if (threadIdx.x < D * D) Rs[threadIdx.x] = R[threadIdx.x];
    __syncthreads();
    const int64_t n = (int64_t)T * B;
    for (int64_t e = (int64_t)blockIdx.x * RNT + threadIdx.x; e < n; e += (int64_t)gridDim.x * RNT) {
        const int64_t t = e / B, b = e - t * B;
        lssmm_rotate_packed<D>(Rs, P + t * NS * BL + b, BL);
    }
}

struct wave_sync {
    __device__ void operator()() const { __syncthreads(); }
};

__global__ void __launch_bounds__(64)
lssmm_small_kernel(lssmm_small_args A, double *__restrict__ gst)
{
    extern __shared__ double st_lds[];
    __shared__ double scratch[lssmm_small_scratch(64)];
    const int total = (int)A.L.total;
    for (int e = threadIdx.x; e < total; e += 64) st_lds[e] = gst[e];
    __syncthreads();
    lssmm_small_body(A, st_lds, scratch, (int)threadIdx.x, 64, wave_sync(), dg_fn(), lg_fn());
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += 64) gst[e] = st_lds[e];
}

constexpr int GMAX = 4;        // lanes per sequence of the default form

inline int64_t nwg(int64_t B, int G) { return (B * G + WNT - 1) / WNT; }

// lanes per sequence: 4 (rows dealt over a DPP quad) while the sequences alone do not fill the
// chip -- below 32768 of them four-lane groups are fewer than eight wavefronts per SIMD and the
// shorter serial chain of a step wins (B = 1e4: 2.5 against 4.0 ms per iteration) --, one thread per
// sequence beyond (a third of the instructions per sequence; B = 1e5: 12.2 against 17 ms).  D > 4
// exists with four lanes only.  vmp_tune_set("lssmm_lanes", 1 | 4) fixes the form.
inline int lanes_for(int D, int64_t B)
{
    const int g = vmp_tune_get("lssmm_lanes", 0);
    if (D > 4) return GMAX;
    if (g == 1) return 1;
    if (g == GMAX) return GMAX;
    return B >= 32768 ? 1 : GMAX;
}

// the backward sweep carries the statistics of all rows of C (G = 4 only: 40 accumulators a lane; one
// thread per sequence would need 112 of them on top of its matrices: 512 registers and 2.3 KB of
// scratch, measured 13.0 against 12.2 ms at B = 1e5)
inline bool fused_stats(int D, int M, int G)
{
    return G == GMAX && D <= 4 && M <= LSSMM_MFUSE && vmp_tune_get("lssmm_fuse", 1) != 0;
}

// workspace (doubles): [sweep partials | set-up partials] then M + 1 integer counters
inline int64_t ws_partials(int D, int M, int64_t B)
{
    const int NS = D * (D + 1) / 2;
    const int64_t g = nwg(B, GMAX) > 0 ? nwg(B, GMAX) : 1;
    const int64_t a = g * (3 * NS + D * D + D + (int64_t)M * (NS + D)) + g;   // backward (+ stats) + log|Phi|
    const int64_t r = 256 * 8;                                 // prepare partials
    return (a > r ? a : r) + 64;
}

}  // namespace

extern "C" {

int32_t vmp_lssmm_limits(int32_t *max_D, int32_t *max_M)
{
    if (max_D) *max_D = LSSMM_DMAX;
    if (max_M) *max_M = LSSMM_MMAX;
    return VMP_OK;
}

int32_t vmp_lssmm_get_layout(int32_t D, int32_t M, vmp_lssmm_layout *out)
{
    if (!out || !lssmm_dims_ok(D, M)) return VMP_ERR_INVALID;
    lssmm_fill_layout(D, M, out);
    return VMP_OK;
}

int32_t vmp_lssmm_workspace_doubles(int32_t D, int32_t M, int64_t B, int32_t T, int64_t *n)
{
    (void)T;
    if (!n || !lssmm_dims_ok(D, M) || B < 0) return VMP_ERR_INVALID;
    *n = ws_partials(D, M, B) + LSSMM_MMAX + 8;
    return VMP_OK;
}

int32_t vmp_lssmm_prepare(vmp_ctx *ctx, const double *Y, const uint8_t *mask, int64_t sm,
                          int64_t sb, int64_t st, int32_t M, int64_t B, int32_t T, int64_t BL,
                          int32_t D, double *Yt, uint64_t *Mw, double *seqobs, double *state,
                          void *workspace)
{
    VMP_REQUIRE(ctx, ctx && Y && mask && Yt && Mw && seqobs && state && workspace, VMP_ERR_INVALID,
                "null argument");
    VMP_REQUIRE(ctx, lssmm_dims_ok(D, M) && B >= 0 && T >= 1 && BL >= B && BL >= 1, VMP_ERR_INVALID,
                "bad dims (D <= 8, M <= 64, M D^2 <= 2048)");
    vmp_lssmm_layout L;
    lssmm_fill_layout(D, M, &L);
    double *partial = reinterpret_cast<double *>(workspace);
    unsigned long long *cnt =
        reinterpret_cast<unsigned long long *>(partial + ws_partials(D, M, B));
    const int64_t nblk = ((B + 31) / 32) * ((T + 31) / 32) * M;
    int64_t g = nblk < (int64_t)ctx->num_cu * 8 ? nblk : (int64_t)ctx->num_cu * 8;
    VMP_HIP_CHECK(ctx, hipMemsetAsync(cnt, 0, (M + 1) * sizeof(unsigned long long), ctx->stream));
    VMP_HIP_CHECK(ctx, hipMemsetAsync(Yt, 0, (size_t)T * M * BL * sizeof(double), ctx->stream));
    VMP_HIP_CHECK(ctx, hipMemsetAsync(Mw, 0, (size_t)T * BL * sizeof(uint64_t), ctx->stream));
    if (g > 0)
        hipLaunchKernelGGL(lssmm_prepare_kernel, dim3((unsigned)g), dim3(RNT), 0, ctx->stream, Y, mask,
                           sm, sb, st, M, B, T, BL, Yt, reinterpret_cast<unsigned long long *>(Mw),
                           partial, cnt);
    const int64_t gb = (BL + RNT - 1) / RNT;
    hipLaunchKernelGGL(lssmm_seqobs_kernel, dim3((unsigned)gb), dim3(RNT), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long *>(Mw), B, T, BL, seqobs, cnt + M);
    hipLaunchKernelGGL(lssmm_setup_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, partial, (int)g,
                       cnt, M, state + L.off_setup);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_lssmm_x_update(vmp_ctx *ctx, int32_t given, const double *Yt, const uint64_t *Mw,
                           const double *seqobs, int32_t M, int64_t B, int32_t T, int64_t BL,
                           int32_t D, double *state, double *F, double *Z, double *P,
                           void *workspace)
{
    VMP_REQUIRE(ctx, ctx && Yt && Mw && seqobs && state && F && Z && P && workspace, VMP_ERR_INVALID,
                "null argument");
    VMP_REQUIRE(ctx, lssmm_dims_ok(D, M) && B >= 0 && T >= 1 && BL >= B, VMP_ERR_INVALID,
                "bad dims (D <= 8, M <= 64, M D^2 <= 2048)");
    VMP_REQUIRE(ctx, BL % 64 == 0, VMP_ERR_INVALID,
                "BL must be a multiple of 64 (the last wavefront works on the padding columns)");
    vmp_lssmm_layout L;
    lssmm_fill_layout(D, M, &L);
    const lssmm_raw ro = lssmm_raw_offsets(D, M);
    const lssmm_tab to = lssmm_tab_offsets(D, M);
    const int NS = (int)L.NS;
    double *partial = reinterpret_cast<double *>(workspace);
    double *raw = state + L.off_raw;
    const int G = lanes_for(D, B);
    const bool fuse = fused_stats(D, M, G) && given != 2;
    const int CL = ro.chain_len - 1;
    const int SL = M * (NS + D);
    sweep_args A;
    A.S.Yt = Yt;
    A.S.Mw = Mw;
    A.S.F = F;
    A.S.Z = Z;
    A.S.P = P;
    A.S.tab = state + L.off_tab;
    A.S.M = M;
    A.S.T = T;
    A.S.BL = BL;
    A.seqobs = seqobs;
    A.B = B;
    A.tab_len = to.len;
    A.nib_len = vmp_tune_get("lssmm_nibbles", 1) ? lssmm_nibble_len(D, M) : 0;
    A.S.nib = nullptr;
    A.pstride = fuse ? CL + SL : CL;
    A.partial = partial;
    A.status = state + L.off_scal;
    A.given = given == 1 ? 1 : 0;
    const int64_t g = nwg(B, G);
    const size_t lds = (size_t)to.len * sizeof(double);
    hipStream_t s = ctx->stream;
    hipEvent_t *ev = ctx->timing ? vmp_next_events(ctx) : nullptr;
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[0], s));
    double *pld = partial + g * (CL + SL);        // log|Phi| partials behind the backward partials
    // D = 1..8 with four lanes per sequence; D <= 4 also as one thread per sequence
#define LSSMM_FOR_DG(MACRO)                                                                          \
    if (G == 1) {                                                                                    \
        switch (D) { case 1: MACRO(1, 1) break; case 2: MACRO(2, 1) break; case 3: MACRO(3, 1) break; \
                     default: MACRO(4, 1) break; }                                                   \
    } else {                                                                                         \
        switch (D) { case 1: MACRO(1, 4) break; case 2: MACRO(2, 4) break; case 3: MACRO(3, 4) break; \
                     case 4: MACRO(4, 4) break; case 5: MACRO(5, 4) break; case 6: MACRO(6, 4) break; \
                     case 7: MACRO(7, 4) break; default: MACRO(8, 4) break; }                        \
    }
    // given == 2: q(X) is unchanged (Y re-observed: new data / mask) -- only the sums that involve the
    // data and the mask are taken again from the stored <x>, <x x^T>; chain sums and log|Phi| stay
    if (given != 2) {
        if (g > 0 && !given) {
            sweep_args Af = A;
            Af.partial = pld;
            const size_t ldsf = lds + (size_t)A.nib_len * sizeof(double);
            // M <= 8 with the nibble tables: the straight-line step (observation terms a step ahead)
            if (M <= 8 && D <= 4 && A.nib_len > 0) {
                // (D <= 4 only: no instance of this form is compiled for the wider blocks)
#define LSSMM_FWD(d)                                                                                 \
    if (G == 1) hipLaunchKernelGGL((lssmm_forward_kernel<d, 1, 8>), dim3((unsigned)g), dim3(WNT), ldsf, s, Af);  \
    else hipLaunchKernelGGL((lssmm_forward_kernel<d, GMAX, 8>), dim3((unsigned)g), dim3(WNT), ldsf, s, Af);
                switch (D) { case 1: LSSMM_FWD(1) break; case 2: LSSMM_FWD(2) break;
                             case 3: LSSMM_FWD(3) break; default: LSSMM_FWD(4) break; }
#undef LSSMM_FWD
            } else {
#define LSSMM_FWD(d, gg) hipLaunchKernelGGL((lssmm_forward_kernel<d, gg, 64>), dim3((unsigned)g), dim3(WNT), ldsf, s, Af);
                LSSMM_FOR_DG(LSSMM_FWD)
#undef LSSMM_FWD
            }
        }
        if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[1], s));
        if (g > 0) {
            if (fuse) {
#define LSSMM_BWDF(d)                                                                                \
    if (A.given) hipLaunchKernelGGL((lssmm_backward_kernel<d, GMAX, LSSMM_MFUSE, true>), dim3((unsigned)g), dim3(WNT), lds, s, A); \
    else hipLaunchKernelGGL((lssmm_backward_kernel<d, GMAX, LSSMM_MFUSE, false>), dim3((unsigned)g), dim3(WNT), lds, s, A);
                switch (D) { case 1: LSSMM_BWDF(1) break; case 2: LSSMM_BWDF(2) break;
                             case 3: LSSMM_BWDF(3) break; default: LSSMM_BWDF(4) break; }
#undef LSSMM_BWDF
            } else {
#define LSSMM_BWD(d, gg)                                                                             \
    if (A.given) hipLaunchKernelGGL((lssmm_backward_kernel<d, gg, 0, true>), dim3((unsigned)g), dim3(WNT), lds, s, A); \
    else hipLaunchKernelGGL((lssmm_backward_kernel<d, gg, 0, false>), dim3((unsigned)g), dim3(WNT), lds, s, A);
                LSSMM_FOR_DG(LSSMM_BWD)
#undef LSSMM_BWD
            }
        }
        // chain sums, log|Phi|: fixed-order sums over the workgroups (an empty local plate: zeros)
        hipLaunchKernelGGL(lssmm_sum_kernel, dim3((unsigned)((CL + 15) / 16)), dim3(RNT), 0, s, partial,
                           (int)g, A.pstride, CL, raw);
        if (fuse)
            hipLaunchKernelGGL(lssmm_sum_kernel, dim3((unsigned)((SL + 15) / 16)), dim3(RNT), 0, s,
                               partial + CL, (int)g, A.pstride, SL, raw + ro.XX);
        if (given || g == 0)
            VMP_HIP_CHECK(ctx, hipMemsetAsync(raw + ro.ld, 0, sizeof(double), s));
        else
            hipLaunchKernelGGL(lssmm_sum_kernel, dim3(1), dim3(RNT), 0, s, pld, (int)g, 1, 1, raw + ro.ld);
    } else if (ev) {
        VMP_HIP_CHECK(ctx, hipEventRecord(ev[1], s));
    }
    if (!fuse) {
        sweep_args As = A;
        As.pstride = SL;
        const int MGd = G == 1 ? MG1 : (D <= 4 ? MG4 : MG8);
        const int ng = (M + MGd - 1) / MGd;
        if (g > 0) {
#define LSSMM_STATS(d, gg) hipLaunchKernelGGL((lssmm_stats_kernel<d, gg, (gg == 1 ? MG1 : (d <= 4 ? MG4 : MG8))>), dim3((unsigned)g, (unsigned)ng), dim3(WNT), 0, s, As);
            LSSMM_FOR_DG(LSSMM_STATS)
#undef LSSMM_STATS
        }
        hipLaunchKernelGGL(lssmm_sum_kernel, dim3((unsigned)((SL + 15) / 16)), dim3(RNT), 0, s, partial,
                           (int)g, SL, SL, raw + ro.XX);
    }
#undef LSSMM_FOR_DG
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[2], s));
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_lssmm_rotate_p(vmp_ctx *ctx, int32_t D, int32_t T, int64_t B, int64_t BL, const double *R,
                           double *P)
{
    VMP_REQUIRE(ctx, ctx && R && P, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, D >= 1 && D <= LSSMM_DMAX && T >= 1 && B >= 0 && BL >= B, VMP_ERR_INVALID,
                "bad dims");
    if (B == 0) return VMP_OK;
    int64_t g = ((int64_t)T * B + RNT - 1) / RNT;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (g > cap) g = cap;
    switch (D) {
#define LSSMM_ROT(d) case d: hipLaunchKernelGGL(lssmm_rotate_p_kernel<d>, dim3((unsigned)g), dim3(RNT), 0, ctx->stream, R, T, B, BL, P); break;
        LSSMM_ROT(1) LSSMM_ROT(2) LSSMM_ROT(3) LSSMM_ROT(4) LSSMM_ROT(5) LSSMM_ROT(6) LSSMM_ROT(7) LSSMM_ROT(8)
#undef LSSMM_ROT
    }
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_lssmm_small_ops(vmp_ctx *ctx, int32_t D, int32_t M, int32_t T, const double *priors,
                            int32_t nu_latent, int32_t nops, const int32_t *ops, double *state)
{
    VMP_REQUIRE(ctx, ctx && priors && ops && state, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, lssmm_dims_ok(D, M) && T >= 1 && nops >= 1 && nops <= 12, VMP_ERR_INVALID,
                "bad dims");
    lssmm_small_args A;
    lssmm_fill_layout(D, M, &A.L);
    A.D = D;
    A.M = M;
    A.T = T;
    A.nops = nops;
    for (int i = 0; i < nops; ++i) A.ops[i] = ops[i];
    for (int i = 0; i < 8; ++i) A.pri[i] = priors[i];
    A.nu_latent = nu_latent;
    const size_t lds = (size_t)A.L.total * sizeof(double);
    if (lds + sizeof(double) * lssmm_small_scratch(64) > 64 * 1024) {
        // beyond the default 64 KB of a workgroup (D = 8 with many rows of C): gfx950 has 160 KB
        static bool raised[64] = {false};          // per device of the process
        const int dev = ctx->device & 63;
        if (!raised[dev]) {
            VMP_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(lssmm_small_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   120 * 1024));
            raised[dev] = true;
        }
    }
    hipLaunchKernelGGL(lssmm_small_kernel, dim3(1), dim3(64), lds, ctx->stream, A, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // extern "C"
