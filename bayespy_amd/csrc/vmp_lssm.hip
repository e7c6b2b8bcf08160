// vmp_lssm.hip -- fused linear state-space model block (BASELINE.json config 5; gfx950).
//
// Model (bayespy/demos/lssm.py:34-103 with a plate of B sequences):
//   x_b0 ~ N(mu0, Lam0^-1),  x_bt ~ N(A x_b,t-1, diag(nu)^-1)        GaussianMarkovChain, n = T
//   y_mbt ~ N(c_m . x_bt, 1/tau)                                      observed, scalar mask
// With dynamics, noise and mask shared by all sequences the block-tridiagonal precision of
// q(X_b) is the SAME matrix for every b (gaussian_markov_chain.py:89-123).  The
// reference's linalg.block_banded_solve (utils/linalg.py:468-575: a Python loop over T calling
// SciPy per block, on (B,T,D,D) arrays) therefore splits into
//   lssm_cov_kernel       ONE D x D recursion over T: block LDL^T forward (S_t^-1, J_t = S_t^-1 E_t,
//                         log|Phi|), inverse blocks backward (V_t, Cov(x_t, x_t+1)) and their sums;
//   lssm_forward_kernel   per sequence: h_t = <tau> sum_m y_mbt <c_m> (+ Lam0 mu0),
//                         z_t = h_t - J_t-1^T z_t-1                      (state in registers)
//   lssm_backward_kernel  per sequence: <x_t> = S_t^-1 z_t - J_t <x_t+1>, written over z, and the
//                         plate sums every other node and the bound read:
//                           sum <x_t><x_t>^T, sum <x_t+1><x_t>^T, sum y_mbt <x_bt>, first / last terms.
// One thread owns one sequence; arrays are TIME-MAJOR so that the B threads of a time step read
// and write consecutive addresses:  Yt[t][m][b] (re-laid-out once: Y is constant after observe()),
// Z[t][i][b].  No (B,T,D,D) array exists; per iteration Y is read twice and Z written twice.
#include "vmp_common.h"

namespace {

constexpr int NT = 256;
// threads per workgroup of the per-sequence passes (one thread per sequence).  1e5 sequences are
// only 1563 wavefronts for 1024 SIMDs; single-wavefront workgroups spread them more evenly over the
// CUs but measured no better (forward 1.83 vs 1.72 ms, backward equal): kept at 256.
constexpr int SNT = 256;
constexpr int DMAX = 16;         // states of the fused block (D <= 8: everything in registers; 9..16: the big-state path)
constexpr int DREG = 8;          // largest D of the register-resident kernels
// smallest D of the SPLIT form: sweeps that carry the state only + the plate sums as a separate pass
// on the matrix cores (instances exist from 7; tune key lssm_split_from).  Below it the backward
// sweep carries the sums in its registers (the D = 7 / 8 instances of that kernel spill).  Within the
// split form the sweeps themselves are the register kernels up to DREG unless lssm_split_sweeps = 1
// asks for the matrix-core sweeps there too (measured slower at D <= 8: half of every tile is padding).
inline int lssm_split_from() { return vmp_tune_get("lssm_split_from", 7); }
inline bool lssm_big(int D) { return D > DREG || (D >= 7 && D >= lssm_split_from()); }

// ---------------------------------------------------------------------------------------------
// set-up: Y (M, B, T) sequence-major -> Yt (T, M, BL) time-major, BL >= B (pad columns zero)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT)
lssm_relayout_kernel(const double *__restrict__ Y, int M, int64_t B, int T, int64_t BL,
                     double *__restrict__ Yt, double *__restrict__ partial)
{
    // 32 x 32 tiles of the (b, t) plane through LDS: both sides coalesced
    __shared__ double tile[32][33];
    __shared__ double red[NT / 64];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const int64_t nbt = (B + 31) / 32, ntt = (T + 31) / 32;
    double syy = 0.0;
    for (int64_t blk = blockIdx.x; blk < nbt * ntt * M; blk += gridDim.x) {
        const int m = (int)(blk / (nbt * ntt));
        const int64_t r = blk - (int64_t)m * nbt * ntt;
        const int64_t bt = r / ntt, tt = r - bt * ntt;
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int64_t b = bt * 32 + j;
            const int t = (int)(tt * 32) + tx;
            double v = 0.0;
            if (b < B && t < T) v = Y[((int64_t)m * B + b) * T + t];
            tile[j][tx] = v;
            syy += v * v;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int t = (int)(tt * 32) + j;
            const int64_t b = bt * 32 + tx;
            if (t < T && b < BL) Yt[((int64_t)t * M + m) * BL + b] = tile[tx][j];
        }
    }
    syy = block_sum<NT>(syy, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = syy;
}

// Z (T, D, BL) time-major <-> X (B, T, D) sequence-major (set-up / read-out)
__global__ void __launch_bounds__(NT)
lssm_x_layout_kernel(double *__restrict__ X, int D, int64_t B, int T, int64_t BL,
                     double *__restrict__ Z, int to_time_major)
{
    const int64_t total = (int64_t)B * T * D;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * NT) {
        // e enumerates (t, i, b) so that the time-major side is coalesced
        const int64_t t = e / ((int64_t)D * B);
        const int64_t r = e - t * D * B;
        const int i = (int)(r / B);
        const int64_t b = r - (int64_t)i * B;
        double *zp = Z + (t * D + i) * BL + b;
        double *xp = X + (b * T + t) * D + i;
        if (to_time_major) *zp = *xp;
        else *xp = *zp;
    }
}

// <x_bt> <- R <x_bt> for every sequence and time step (the rotation x -> R x of the state space,
// transformations.py:1167-1176 / gaussian_markov_chain.py:51-65); Z is time-major (T, D, BL).
template <int D>
__global__ void __launch_bounds__(NT)
lssm_rotate_kernel(const double *__restrict__ R, int T, int64_t B, int64_t BL, double *__restrict__ Z)
{
    double r[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) r[i][j] = R[i * D + j];
    const int64_t total = (int64_t)T * B;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int64_t t = e / B, b = e - t * B;
        double *zp = Z + t * D * BL + b;
        double x[D];
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = zp[(int64_t)i * BL];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) s += r[i][j] * x[j];
            zp[(int64_t)i * BL] = s;
        }
    }
}

__global__ void __launch_bounds__(NT)
lssm_sum_kernel(const double *__restrict__ partial, int n, int stride, int len, double *__restrict__ out)
{
    // out[j] = sum_b partial[b * stride + j], fixed order.  A workgroup owns 16 neighbouring outputs;
    // its 16 row-lanes split the partial blocks (eight loads in flight each) and meet in LDS -- with
    // a thread per output the ~400 blocks were summed one dependent load after the other (80 us).
    __shared__ double tile[16][17];
    const int kx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + kx;
    double acc = 0.0;
    if (j < len) {
        int b = ry;
        for (; b + 7 * 16 < n; b += 8 * 16) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + 16 * u) * stride + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; b < n; b += 16) acc += partial[(int64_t)b * stride + j];
    }
    tile[ry][kx] = acc;
    __syncthreads();
    if (ry == 0 && j < len) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += tile[r][kx];
        out[j] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// shared covariance recursion: ONE wavefront, lane (i, j) = l / D, l % D owns one matrix element
// ---------------------------------------------------------------------------------------------
struct cov_args {
    int T, D;
    int shortcut;                    // big kernel: ulp of the stationarity rule (tune key lssm_cov_shortcut; 0 = off)
    // inputs (device, D x D row-major unless noted)
    const double *Dg0, *Dgm, *DgT;   // diagonal blocks of Phi: t = 0, 0 < t < T-1, t = T-1
    const double *E;                 // super-diagonal block Phi[t, t+1] (same for all t)
    // outputs
    double *Sinv;                    // T x D x D
    double *J;                       // (T-1) x D x D      J_t = S_t^-1 E
    double *sums;                    // 5 D^2 + 2: sum_t V_t | V_0 | V_{T-1} | sum_t Cov(x_t,x_t+1) | (unused)
                                     //            then log|Phi| and a not-positive-definite flag
};

// ---- D x D algebra of ONE wavefront in registers: lane l = (i, j) = (l / D, l % D) owns element
// (i, j); operands travel by lane permutes (ds_bpermute, no LDS storage, no barriers) ----------
__device__ inline double lane_get(double v, int src)
{
    return __shfl(v, src, 64);
}

// ---- D = 4: the matrix occupies ONE 16-lane DPP row (lane = 4 i + j), so the operands travel by
// DPP moves (vector ALU, a few cycles) instead of ds_bpermute (the LDS crossbar, ~130 cycles each
// on the serial path of the recursion):
//   element (i, K) to the lanes of row i     : quad_perm [K, K, K, K]
//   element (K, j) to the lanes of column j  : four row_newbcast of lanes 4 K + 0..3, selected by j
template <int K>
__device__ __forceinline__ double quad_bcast(double v)
{
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int LANE>
__device__ __forceinline__ double row16_bcast(double v)
{
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + LANE, 0xf, 0xf, true);
}

__device__ __forceinline__ double sel4(double b0, double b1, double b2, double b3, int idx)
{
    const double lo = (idx & 1) ? b1 : b0, hi = (idx & 1) ? b3 : b2;
    return (idx & 2) ? hi : lo;
}

// element (K, c) of a 4 x 4 matrix for this lane's c (c = j: column operand; c = i: transposed use)
template <int K>
__device__ __forceinline__ double col_get4(double v, int c)
{
    return sel4(row16_bcast<4 * K + 0>(v), row16_bcast<4 * K + 1>(v), row16_bcast<4 * K + 2>(v),
                row16_bcast<4 * K + 3>(v), c);
}

template <int P>
__device__ __forceinline__ void spd_inverse4_steps(double &v, int i, int j, double &prod, double &ex,
                                                   int &bad)
{
    if constexpr (P < 4) {
        const double piv = row16_bcast<5 * P>(v);
        const double ci = quad_bcast<P>(v), rj = col_get4<P>(v, j);
        if (!(piv > 0.0)) bad = 1;
        const double q = prod * piv;
        ex += (double)__builtin_amdgcn_frexp_exp(q);
        prod = __builtin_amdgcn_frexp_mant(q);
        const double d = fast_recip(piv);
        if (i == P) v = (j == P) ? d : rj * d;
        else if (j == P) v = -ci * d;
        else v = v - ci * rj * d;
        spd_inverse4_steps<P + 1>(v, i, j, prod, ex, bad);
    }
}

// sum_k a[i][k] * bk[k]   (bk[k] = the lane's element (k, j) of the right operand, given)
__device__ __forceinline__ double rowdot4(double a, const double (&bk)[4])
{
    return quad_bcast<0>(a) * bk[0] + quad_bcast<1>(a) * bk[1] + quad_bcast<2>(a) * bk[2]
           + quad_bcast<3>(a) * bk[3];
}

// sum_k ak[k] * b[k][j]   (ak[k] given per lane)
__device__ __forceinline__ double coldot4(const double (&ak)[4], double b, int j)
{
    return ak[0] * col_get4<0>(b, j) + ak[1] * col_get4<1>(b, j) + ak[2] * col_get4<2>(b, j)
           + ak[3] * col_get4<3>(b, j);
}

// (A B)[i][j] for this lane; ta / tb: take A / B transposed
template <int D>
__device__ __forceinline__ double reg_matmul(double a, double b, int i, int j, bool ta, bool tb)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double x = lane_get(a, ta ? k * D + i : i * D + k);
        const double y = lane_get(b, tb ? j * D + k : k * D + j);
        s += x * y;
    }
    return s;
}

// in: element (i, j) of an SPD matrix; out: element of the inverse.  Pivot-free Gauss-Jordan,
// the pivots multiplied into (prod, ex) = mantissa x 2^ex (no logarithm on the serial path).
template <int D>
__device__ __forceinline__ double reg_spd_inverse(double v, int i, int j, bool act, double &prod,
                                                  double &ex, int &bad)
{
#pragma unroll
    for (int p = 0; p < D; ++p) {
        const double piv = lane_get(v, p * D + p);
        const double ci = lane_get(v, i * D + p), rj = lane_get(v, p * D + j);
        if (!(piv > 0.0)) bad = 1;
        const double q = prod * piv;
        ex += (double)__builtin_amdgcn_frexp_exp(q);
        prod = __builtin_amdgcn_frexp_mant(q);
        const double d = fast_recip(piv);
        if (i == p) v = (j == p) ? d : rj * d;
        else if (j == p) v = -ci * d;
        else v = v - ci * rj * d;
    }
    return act ? v : 0.0;
}

// all elements of the two D x D iterates agree within 8 ulp of the largest magnitude
__device__ inline bool stationary(double a, double b, bool act)
{
    double m = act ? fabs(a) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    return __all(!act || fabs(a - b) <= 8.0 * 2.220446049250313e-16 * m);
}

// phase bit 0: forward recursion (S^-1, J, log|Phi|, status); bit 1: backward recursion (V_t,
// Cov(x_t, x_t+1) and their sums).  Only the forward half is needed by the per-sequence passes,
// so vmp_lssm_x_update runs the backward half on a side stream beside them.
// Forward steps [t0, t1) per launch (round 3): the per-sequence forward sweep consumes J_t in time
// order, so vmp_lssm_x_update cuts both recursions into segments and runs segment k + 1 of this
// kernel beside segment k of the sweep.  The state between launches (S_t, the running pivot
// product, status, the stationarity marker, "finished") lives in sums[4 D^2 ..] / sums[5 D^2 + 4 ..].
template <int D>
__global__ void __launch_bounds__(64)
lssm_cov_kernel(cov_args a, int phase, int t0, int t1)
{
    const int l = threadIdx.x, T = a.T;
    const bool act = l < D * D;
    const int i = act ? l / D : 0, j = act ? l % D : 0;
    const double e = act ? a.E[i * D + j] : 0.0;
    const double dgm = act ? a.Dgm[i * D + j] : 0.0, dgT = act ? a.DgT[i * D + j] : 0.0;
    double prod = 1.0, ex = 0.0;
    int bad = 0;
    // ---- forward: S_0 = Dg_0; S_t+1 = Dg_t+1 - E^T J_t,  J_t = S_t^-1 E -----------------------
    // In the interior (0 < t < T-1) every step applies the SAME map to S_t, a contraction towards
    // the stationary solution of the filter's Riccati equation.  Once S_t+1 agrees with S_t to
    // working precision (every element within 8 ulp of the largest element) the sequence has
    // converged and further steps would only reproduce rounding noise: S^-1, J and the
    // log-pivots of the remaining interior steps are filled in without being recomputed.
    double s = act ? a.Dg0[i * D + j] : 0.0;
    int fix_from = -1;                 // steps fix_from .. T-2 share one (S^-1, J)
    if (!(phase & 1)) fix_from = (int)a.sums[5 * D * D + 2];       // left by the forward launch
    if ((phase & 1) && t0 > 0) {
        // a later segment: an earlier one that found the stationary stretch has finished the
        // whole recursion already
        if (a.sums[5 * D * D + 6] != 0.0) return;
        s = act ? a.sums[4 * D * D + l] : 0.0;
        prod = a.sums[5 * D * D + 4];
        ex = a.sums[5 * D * D + 5];
        bad = (int)a.sums[5 * D * D + 1];
        fix_from = (int)a.sums[5 * D * D + 2];
    }
    int tend = t1 < T ? t1 : T;
    // D = 4: the constant operand E is fetched once: E[k][j] and E[k][i] for this lane
    double ekj[4] = {0.0, 0.0, 0.0, 0.0}, eki[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (D == 4) {
        ekj[0] = col_get4<0>(e, j); ekj[1] = col_get4<1>(e, j);
        ekj[2] = col_get4<2>(e, j); ekj[3] = col_get4<3>(e, j);
        eki[0] = col_get4<0>(e, i); eki[1] = col_get4<1>(e, i);
        eki[2] = col_get4<2>(e, i); eki[3] = col_get4<3>(e, i);
    }
    for (int t = t0; (phase & 1) && t < tend; ++t) {
        double p1 = 1.0, e1 = 0.0;
        double sinv;
        if constexpr (D == 4) {
            sinv = s;
            spd_inverse4_steps<0>(sinv, i, j, p1, e1, bad);
            sinv = act ? sinv : 0.0;
        } else {
            sinv = reg_spd_inverse<D>(s, i, j, act, p1, e1, bad);
        }
        {
            const double q = prod * p1;
            ex += e1 + (double)__builtin_amdgcn_frexp_exp(q);
            prod = __builtin_amdgcn_frexp_mant(q);
        }
        if (act) a.Sinv[(int64_t)t * D * D + l] = sinv;
        if (t < T - 1) {
            double jt, ej;
            if constexpr (D == 4) {
                jt = rowdot4(sinv, ekj);                                        // S^-1 E
                ej = coldot4(eki, jt, j);                                       // E^T J
            } else {
                jt = reg_matmul<D>(sinv, e, i, j, false, false);
                ej = reg_matmul<D>(e, jt, i, j, true, false);
            }
            if (act) a.J[(int64_t)t * D * D + l] = jt;
            const double snew = ((t + 1 < T - 1) ? dgm : dgT) - ej;
            // (tested every eighth step: the test itself is a 64-lane max reduction on the serial path)
            if (t >= 1 && (t & 7) == 0 && t + 1 < T - 1 && stationary(snew, s, act)) {
                const int tl = T - 2;                                   // last interior step
                for (int tt = t + 1; tt <= tl; ++tt) {
                    if (act) {
                        a.Sinv[(int64_t)tt * D * D + l] = sinv;
                        a.J[(int64_t)tt * D * D + l] = jt;
                    }
                }
                // (tl - t) more copies of this step's pivots
                ex += (double)(tl - t) * (e1 + log2(p1));
                fix_from = t;
                s = dgT - ej;                                            // S_T-1
                t = tl;
                tend = T;                       // this launch finishes the recursion
                continue;
            }
            s = snew;
        }
    }
    const double ldsum = (log(prod) + ex * 0.69314718055994530942);
    if (phase & 1) {
        if (act) a.sums[4 * D * D + l] = s;                 // state for the next segment
        if (l == 0) {
            a.sums[5 * D * D + 0] = ldsum;
            a.sums[5 * D * D + 1] = (double)bad;
            // diagnostics: the step at which the forward map became stationary (-1: never)
            a.sums[5 * D * D + 2] = (double)fix_from;
            a.sums[5 * D * D + 4] = prod;
            a.sums[5 * D * D + 5] = ex;
            a.sums[5 * D * D + 6] = (tend >= T) ? 1.0 : 0.0;
        }
        if (!(phase & 2)) return;
        __threadfence();            // the backward half below re-reads S^-1, J from memory
    }
    // ---- backward: V_T-1 = S_T-1^-1;  C_t = -J_t V_t+1;  V_t = S_t^-1 - C_t J_t^T -----------------
    double v = act ? a.Sinv[(int64_t)(T - 1) * D * D + l] : 0.0;
    double sv = v, sc = 0.0;
    const double vlast = v;
    double vprev = 0.0, cprev = 0.0;
    int bfix = -1, have_prev = 0;
    double jn = (T >= 2 && act) ? a.J[(int64_t)(T - 2) * D * D + l] : 0.0;
    double sn = (T >= 2 && act) ? a.Sinv[(int64_t)(T - 2) * D * D + l] : 0.0;
    for (int t = T - 2; t >= 0; --t) {
        const double jt = jn, si = sn;
        if (t > 0) {                                   // next step's operands: off the serial path
            jn = act ? a.J[(int64_t)(t - 1) * D * D + l] : 0.0;
            sn = act ? a.Sinv[(int64_t)(t - 1) * D * D + l] : 0.0;
        }
        // same (S^-1, J) as the step before and a V_t+1 that has stopped moving: the same V, C
        // again, down to the first step of the stationary stretch
        if (have_prev && fix_from >= 0 && t >= fix_from && t + 1 <= T - 2
            && stationary(v, vprev, act)) {
            const double cnt = (double)(t - fix_from + 1);
            sv += cnt * v;
            sc += cnt * cprev;
            bfix = t;
            t = fix_from;
            if (t > 0) {
                jn = act ? a.J[(int64_t)(t - 1) * D * D + l] : 0.0;
                sn = act ? a.Sinv[(int64_t)(t - 1) * D * D + l] : 0.0;
            }
            continue;
        }
        const double c = -reg_matmul<D>(jt, v, i, j, false, false);          // Cov(x_t, x_t+1)
        const double cj = reg_matmul<D>(c, jt, i, j, false, true);           // C J^T
        vprev = v;
        v = si - cj;
        have_prev = 1;
        cprev = c;
        sv += v;
        sc += c;
    }
    if (act) {
        a.sums[0 * D * D + l] = sv;
        a.sums[1 * D * D + l] = v;         // V_0
        a.sums[2 * D * D + l] = vlast;
        a.sums[3 * D * D + l] = sc;
    }
    if (l == 0) a.sums[5 * D * D + 3] = (double)bfix;      // diagnostics, backward map
}

// ---------------------------------------------------------------------------------------------
// per-sequence recursions
// ---------------------------------------------------------------------------------------------
// z_t = h_t - J_t-1^T z_t-1,  h_t = tau sum_m y_mbt c_m  (+ h0 at t = 0)
// CK = 0: every z_t is written to Z (T, D, BL).  CK = S > 0 (checkpoint form): only z_{kS-1},
// k = 1, 2, ..., is written, to Zc (T/S, D, BL); the backward kernel forms the S steps of a block
// again from the checkpoint in front of it (lssm_backward_ck_kernel).  t0 is a multiple of S.
// IDENT: the "observations" are the projected data H = tau C^T Y already (M = D, C = I, tau = 1):
// h_t = y_t, no D x D table of tau c_m (the big-state path: 256 uniform values at D = 16).
template <int D, int MM, int CK, bool IDENT = false>
__global__ void __launch_bounds__(SNT)
lssm_forward_kernel(const double *__restrict__ Yt, int M, int64_t B, int T, int64_t BL,
                    const double *__restrict__ Cm /* M x D */, const double *__restrict__ tau_ptr,
                    const double *__restrict__ h0 /* D */, const double *__restrict__ J,
                    double *__restrict__ Z, int t0, int t1)
{
    const int64_t b = (int64_t)blockIdx.x * SNT + threadIdx.x;
    if (b >= B) return;
    const double tau = IDENT ? 1.0 : tau_ptr[0];
    double tc[IDENT ? 1 : MM][IDENT ? 1 : D];      // tau * c_m (uniform: scalar registers)
    if constexpr (!IDENT) {
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int i = 0; i < D; ++i) tc[m][i] = (m < M) ? tau * Cm[m * D + i] : 0.0;
    }
    double z[D];
    // steps [t0, t1): a later segment picks z_{t0-1} up where the previous launch left it
#pragma unroll
    for (int i = 0; i < D; ++i) {
        if constexpr (CK == 0) z[i] = t0 > 0 ? Z[((int64_t)(t0 - 1) * D + i) * BL + b] : 0.0;
        else z[i] = t0 > 0 ? Z[((int64_t)(t0 / CK - 1) * D + i) * BL + b] : 0.0;
    }
    const double *yp = Yt + b;
    double ycur[MM], ynxt[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m)
        ycur[m] = (m < M) ? __builtin_nontemporal_load(&yp[((int64_t)t0 * M + m) * BL]) : 0.0;
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) {
#pragma unroll
            for (int m = 0; m < MM; ++m)
                ynxt[m] = (m < M) ? __builtin_nontemporal_load(&yp[((int64_t)(t + 1) * M + m) * BL]) : 0.0;
        }
        double h[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = (t == 0) ? h0[i] : 0.0;
            if constexpr (IDENT) {
                s += ycur[i];
            } else {
#pragma unroll
                for (int m = 0; m < MM; ++m) s += ycur[m] * tc[m][i];
            }
            h[i] = s;
        }
        if (t > 0) {
            const double *Jt = J + (int64_t)(t - 1) * D * D;     // uniform
            double zn[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = h[i];
#pragma unroll
                for (int k = 0; k < D; ++k) s -= Jt[k * D + i] * z[k];     // (J^T z)_i
                zn[i] = s;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) z[i] = zn[i];
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) z[i] = h[i];
        }
        if constexpr (CK == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i)
                __builtin_nontemporal_store(z[i], &Z[((int64_t)t * D + i) * BL + b]);
        } else if ((t + 1) % CK == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i)
                __builtin_nontemporal_store(z[i], &Z[((int64_t)((t + 1) / CK - 1) * D + i) * BL + b]);
        }
#pragma unroll
        for (int m = 0; m < MM; ++m) ycur[m] = ynxt[m];
    }
}

// x_t = S_t^-1 z_t - J_t x_t+1 (in place over Z) and the plate sums of this workgroup:
//   [0, D^2)            sum_bt x_t x_t^T
//   [D^2, 2 D^2)        sum_b sum_{t<T-1} x_t+1 x_t^T
//   [2 D^2, 3 D^2)      sum_b x_0 x_0^T
//   [3 D^2, 4 D^2)      sum_b x_T-1 x_T-1^T
//   [4 D^2, 4 D^2 + D)  sum_b x_0
//   then M x D          sum_bt y_mbt x_bt
template <int D, int MM>
__global__ void __launch_bounds__(SNT)
lssm_backward_kernel(const double *__restrict__ Yt, int M, int64_t B, int T, int64_t BL,
                     const double *__restrict__ Sinv, const double *__restrict__ J,
                     double *__restrict__ Z, double *__restrict__ partial, int plen, int given)
{
    __shared__ double red[SNT / 64];
    const int64_t b = (int64_t)blockIdx.x * SNT + threadIdx.x;
    const bool live = b < B;
    const int64_t bb = live ? b : 0;
    double sxx[D][D], snp[D][D], sx0[D][D], sxT[D][D], s0[D], syx[MM][D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        s0[i] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) sxx[i][j] = snp[i][j] = sx0[i][j] = sxT[i][j] = 0.0;
    }
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int i = 0; i < D; ++i) syx[m][i] = 0.0;
    double xn[D];
#pragma unroll
    for (int i = 0; i < D; ++i) xn[i] = 0.0;
    double *zp = Z + bb;
    const double *yp = Yt + bb;
    double zc[D], yc[MM];
#pragma unroll
    for (int i = 0; i < D; ++i) zc[i] = __builtin_nontemporal_load(&zp[((int64_t)(T - 1) * D + i) * BL]);
#pragma unroll
    for (int m = 0; m < MM; ++m) yc[m] = (m < M) ? __builtin_nontemporal_load(&yp[((int64_t)(T - 1) * M + m) * BL]) : 0.0;
    for (int t = T - 1; t >= 0; --t) {
        double zq[D], yq[MM];
        if (t > 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) zq[i] = __builtin_nontemporal_load(&zp[((int64_t)(t - 1) * D + i) * BL]);
#pragma unroll
            for (int m = 0; m < MM; ++m)
                yq[m] = (m < M) ? __builtin_nontemporal_load(&yp[((int64_t)(t - 1) * M + m) * BL]) : 0.0;
        }
        double x[D];
        if (given) {                        // delta moments of a given X (initialize_from_value)
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = zc[i];
        } else {
            const double *St = Sinv + (int64_t)t * D * D;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) s += St[i * D + k] * zc[k];
                x[i] = s;
            }
        }
        if (!given && t < T - 1) {
            const double *Jt = J + (int64_t)t * D * D;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = x[i];
#pragma unroll
                for (int k = 0; k < D; ++k) s -= Jt[i * D + k] * xn[k];
                x[i] = s;
            }
        }
        if (live) {
            if (!given) {
#pragma unroll
                for (int i = 0; i < D; ++i) __builtin_nontemporal_store(x[i], &zp[((int64_t)t * D + i) * BL]);
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) sxx[i][j] += x[i] * x[j];
            if (t < T - 1) {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) snp[i][j] += xn[i] * x[j];
            }
            if (t == T - 1) {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) sxT[i][j] = x[i] * x[j];
            }
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    s0[i] = x[i];
#pragma unroll
                    for (int j = 0; j <= i; ++j) sx0[i][j] = x[i] * x[j];
                }
            }
#pragma unroll
            for (int m = 0; m < MM; ++m)
#pragma unroll
                for (int i = 0; i < D; ++i) syx[m][i] += yc[m] * x[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            xn[i] = x[i];
            zc[i] = zq[i];
        }
#pragma unroll
        for (int m = 0; m < MM; ++m) yc[m] = yq[m];
    }
    // workgroup sums (fixed order), symmetric halves mirrored
    double *pb = partial + (int64_t)blockIdx.x * plen;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double a = block_sum<SNT>(j <= i ? sxx[i][j] : sxx[j][i], red);
            const double c = block_sum<SNT>(snp[i][j], red);
            const double d0 = block_sum<SNT>(j <= i ? sx0[i][j] : sx0[j][i], red);
            const double dT = block_sum<SNT>(j <= i ? sxT[i][j] : sxT[j][i], red);
            if (threadIdx.x == 0) {
                pb[i * D + j] = a;
                pb[D * D + i * D + j] = c;
                pb[2 * D * D + i * D + j] = d0;
                pb[3 * D * D + i * D + j] = dT;
            }
        }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double a = block_sum<SNT>(s0[i], red);
        if (threadIdx.x == 0) pb[4 * D * D + i] = a;
    }
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double a = block_sum<SNT>(syx[m][i], red);
            if (threadIdx.x == 0 && m < M) pb[4 * D * D + D + m * D + i] = a;
        }
}

// Checkpoint form of the backward sweep (round 3): the forward sweep has stored z only every S
// steps (Zc), so z is neither written nor read as a (T, D, BL) array -- 2 x 8 B T D bytes less per
// iteration; Y, which this kernel reads anyway for sum y x^T, is all it takes to form the S values
// of z of a block again.  Blocks of S steps from the last to the first:
//   forward in the block   z_t = h_t - J_t-1^T z_t-1 from the checkpoint z_{kS-1}: the arithmetic of
//                          lssm_forward_kernel (same order: the values are the same bit for bit),
//                          y_t and z_t of the block kept in registers
//   backward in the block  x_t = S_t^-1 z_t - J_t x_t+1, written to Z, and the plate sums -- the
//                          arithmetic and the order of lssm_backward_kernel
// The sums of the first / last step are formed after the loop from x_0 (what the loop ends on) and x_T-1
// (kept from its first step) instead of being carried through it as D x D accumulators.  Output as lssm_backward_kernel.
template <int D, int MM, int S>
__global__ void __launch_bounds__(SNT, 2)
lssm_backward_ck_kernel(const double *__restrict__ Yt, int M, int64_t B, int T, int64_t BL,
                        const double *__restrict__ Cm, const double *__restrict__ tau_ptr,
                        const double *__restrict__ h0, const double *__restrict__ Sinv,
                        const double *__restrict__ J, const double *__restrict__ Zc,
                        double *__restrict__ Z, double *__restrict__ partial, int plen)
{
    __shared__ double red[SNT / 64];
    const int64_t b = (int64_t)blockIdx.x * SNT + threadIdx.x;
    const bool live = b < B;
    const int64_t bb = live ? b : 0;
    // tau * c_m in LDS (broadcast reads): in registers these 32 uniform values would cost the
    // second wavefront per SIMD
    __shared__ double tc[MM][D];
    if (threadIdx.x < MM * D) {
        const int m = threadIdx.x / D, i = threadIdx.x % D;
        tc[m][i] = (m < M) ? tau_ptr[0] * Cm[m * D + i] : 0.0;
    }
    __syncthreads();
    // the block's z and x_T-1 wait in LDS (per-thread slots, conflict-free): 40 registers less
    __shared__ double zb[S][D][SNT];
    __shared__ double xls[D][SNT];
    const int tid = threadIdx.x;
    double sxx[D][D], snp[D][D], syx[MM][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) sxx[i][j] = snp[i][j] = 0.0;
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int i = 0; i < D; ++i) syx[m][i] = 0.0;
    double xn[D];
#pragma unroll
    for (int i = 0; i < D; ++i) xn[i] = xls[i][tid] = 0.0;
    double *zp = Z + bb;
    const double *yp = Yt + bb;
    const double *cp = Zc + bb;
    const int nblk = (T + S - 1) / S;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int tb = kb * S;
        asm volatile("" ::: "memory");      // tc is read from LDS in every block, not hoisted
        double yb[S][MM];
        // ---- the block's observations and the checkpoint in front of it -------------------
#pragma unroll
        for (int u = 0; u < S; ++u)
#pragma unroll
            for (int m = 0; m < MM; ++m)
                yb[u][m] = (m < M && tb + u < T)
                    ? __builtin_nontemporal_load(&yp[((int64_t)(tb + u) * M + m) * BL]) : 0.0;
        double z[D];
#pragma unroll
        for (int i = 0; i < D; ++i)
            z[i] = kb > 0 ? __builtin_nontemporal_load(&cp[((int64_t)(kb - 1) * D + i) * BL]) : 0.0;
        // ---- forward within the block ------------------------------------------------------
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int t = tb + u;
            if (t < T) {
                double h[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double s = (t == 0) ? h0[i] : 0.0;
#pragma unroll
                    for (int m = 0; m < MM; ++m) s += yb[u][m] * tc[m][i];
                    h[i] = s;
                }
                if (t > 0) {
                    const double *Jt = J + (int64_t)(t - 1) * D * D;     // uniform
                    double zn[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double s = h[i];
#pragma unroll
                        for (int k = 0; k < D; ++k) s -= Jt[k * D + i] * z[k];     // (J^T z)_i
                        zn[i] = s;
                    }
#pragma unroll
                    for (int i = 0; i < D; ++i) z[i] = zn[i];
                } else {
#pragma unroll
                    for (int i = 0; i < D; ++i) z[i] = h[i];
                }
            }
#pragma unroll
            for (int i = 0; i < D; ++i) zb[u][i][tid] = z[i];
        }
        // ---- backward within the block -----------------------------------------------------
#pragma unroll
        for (int u = S - 1; u >= 0; --u) {
            const int t = tb + u;
            if (t < T) {
                const double *St = Sinv + (int64_t)t * D * D;
                double x[D];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) s += St[i * D + k] * zb[u][k][tid];
                    x[i] = s;
                }
                if (t < T - 1) {
                    const double *Jt = J + (int64_t)t * D * D;
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double s = x[i];
#pragma unroll
                        for (int k = 0; k < D; ++k) s -= Jt[i * D + k] * xn[k];
                        x[i] = s;
                    }
                }
                if (live) {
#pragma unroll
                    for (int i = 0; i < D; ++i)
                        __builtin_nontemporal_store(x[i], &zp[((int64_t)t * D + i) * BL]);
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) sxx[i][j] += x[i] * x[j];
                    if (t < T - 1) {
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) snp[i][j] += xn[i] * x[j];
                    }
#pragma unroll
                    for (int m = 0; m < MM; ++m)
#pragma unroll
                        for (int i = 0; i < D; ++i) syx[m][i] += yb[u][m] * x[i];
                }
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    xn[i] = x[i];
                    if (t == T - 1) xls[i][tid] = x[i];
                }
            }
        }
    }
    // x_0 is what the loop ended on; x_T-1 was kept from the first step
    double xl[D];
#pragma unroll
    for (int i = 0; i < D; ++i) xl[i] = xls[i][tid];
    // workgroup sums (fixed order), symmetric halves mirrored
    double *pb = partial + (int64_t)blockIdx.x * plen;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double a = block_sum<SNT>(j <= i ? sxx[i][j] : sxx[j][i], red);
            const double c = block_sum<SNT>(snp[i][j], red);
            const double d0 = block_sum<SNT>(live ? (j <= i ? xn[i] * xn[j] : xn[j] * xn[i]) : 0.0, red);
            const double dT = block_sum<SNT>(live ? (j <= i ? xl[i] * xl[j] : xl[j] * xl[i]) : 0.0, red);
            if (threadIdx.x == 0) {
                pb[i * D + j] = a;
                pb[D * D + i * D + j] = c;
                pb[2 * D * D + i * D + j] = d0;
                pb[3 * D * D + i * D + j] = dT;
            }
        }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double a = block_sum<SNT>(live ? xn[i] : 0.0, red);
        if (threadIdx.x == 0) pb[4 * D * D + i] = a;
    }
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double a = block_sum<SNT>(syx[m][i], red);
            if (threadIdx.x == 0 && m < M) pb[4 * D * D + D + m * D + i] = a;
        }
}

int plen_of(int D, int M) { return 4 * D * D + D + M * D; }

// workspace: [partial sums of the sweeps | relayout partials] then the checkpoints of the forward
// sweep, (T / CKS, D, BL) with BL <= the sequences rounded up to 256
// ---------------------------------------------------------------------------------------------
// big-state path (8 < D <= 16; round 6).  The register-resident sweeps hold the state AND the
// D x D plate sums of a sequence in one thread (D = 8 spills); here the sweeps carry the state
// only, and the plate sums are formed behind them:
//   lssm_forward_kernel<D, D, 0, true>   on the projected data H = tau C^T Y
//   lssm_backward_plain_kernel<D>        x_t = S_t^-1 z_t - J_t x_t+1 in place over Z
//   lssm_pairsum_kernel<D>               sum_{b, t in [T0, T1)} A[t + dt][r][b] Z[t][i][b] for eight rows
//                                        r of an array A (T, R, BL) per workgroup row: sum x x^T
//                                        (A = Z), sum x_t+1 x_t^T (A = Z, dt = 1), the first / last
//                                        step's terms (one-step ranges), sum y x^T (A = Yt)
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(SNT)
lssm_backward_plain_kernel(int64_t B, int T, int64_t BL, const double *__restrict__ Sinv,
                           const double *__restrict__ J, double *__restrict__ Z)
{
    const int64_t b = (int64_t)blockIdx.x * SNT + threadIdx.x;
    if (b >= B) return;
    double *zp = Z + b;
    double xn[D], zc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        xn[i] = 0.0;
        zc[i] = __builtin_nontemporal_load(&zp[((int64_t)(T - 1) * D + i) * BL]);
    }
    for (int t = T - 1; t >= 0; --t) {
        double zq[D];
        if (t > 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) zq[i] = __builtin_nontemporal_load(&zp[((int64_t)(t - 1) * D + i) * BL]);
        }
        const double *St = Sinv + (int64_t)t * D * D;
        double x[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += St[i * D + k] * zc[k];
            x[i] = s;
        }
        if (t < T - 1) {
            const double *Jt = J + (int64_t)t * D * D;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = x[i];
#pragma unroll
                for (int k = 0; k < D; ++k) s -= Jt[i * D + k] * xn[k];
                x[i] = s;
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            __builtin_nontemporal_store(x[i], &zp[((int64_t)t * D + i) * BL]);
            xn[i] = x[i];
            zc[i] = zq[i];
        }
    }
}

// The same two sweeps ON THE MATRIX CORES (default for 8 < D <= 16; tune key lssm_big_mfma).  A
// wavefront owns 32 consecutive sequences as two 16-column tiles (even / odd columns, so that a
// lane's 16-byte access serves both); the state of a tile is ONE accumulator of
// v_mfma_f64_16x16x4_f64 -- lane l holds rows (l >> 4) + 4 r of column l & 15 -- and that is also the
// instruction's B-operand layout for k-step r: z_t = h_t - J_t-1^T z_t-1 and x_t = S_t^-1 z_t - J_t x_t+1
// chain through the accumulators without moving a value between lanes.  The A operands are the
// (uniform, cache-resident) D x D blocks of the covariance recursion; operands of step t +- 1 are
// requested before the matrix instructions of step t.  4 (forward) / 8 (backward) instructions per
// tile and step instead of 2 D^2 / 4 D^2 scalar-operand multiply-adds per thread.  Rows >= D of the
// padded 16 x 16 blocks are zero; columns >= B (padding of the plate to BL) are computed like the
// others and ignored by every reader.  BL % 32 == 0, 16-byte aligned arrays.
template <int D>
__global__ void __launch_bounds__(256)
lssm_forward_mfma_kernel(const double *__restrict__ H, int64_t B, int T, int64_t BL,
                         const double *__restrict__ h0, const double *__restrict__ J,
                         double *__restrict__ Z, int t0, int t1)
{
    const int l = threadIdx.x & 63, l15 = l & 15, l4 = l >> 4;
    const int64_t b0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
    if (b0 >= B) return;
    const int64_t col = b0 + 2 * l15;
    const v2f64 zero2 = v2f64{0.0, 0.0};
    auto load_rows = [&](const double *base, int t, v2f64 (&out)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            out[r] = row < D ? *reinterpret_cast<const v2f64 *>(&base[((int64_t)t * D + row) * BL + col])
                             : zero2;
        }
    };
    // A[i = l15][k = 4 q + l4] = -J_t[k][i]
    auto load_jt = [&](int t, double (&a)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + l4;
            a[q] = (l15 < D && k < D) ? -J[(int64_t)t * D * D + k * D + l15] : 0.0;
        }
    };
    v4f64 z0 = {0.0, 0.0, 0.0, 0.0}, z1 = {0.0, 0.0, 0.0, 0.0};
    if (t0 > 0) {
        v2f64 zp[4];
        load_rows(Z, t0 - 1, zp);
#pragma unroll
        for (int r = 0; r < 4; ++r) { z0[r] = zp[r].x; z1[r] = zp[r].y; }
    }
    v2f64 h[4], hn[4];
    double a[4] = {0.0, 0.0, 0.0, 0.0}, an[4];
    load_rows(H, t0, h);
    if (t0 > 0) load_jt(t0 - 1, a);
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) {
            load_rows(H, t + 1, hn);
            load_jt(t, an);
        }
        v4f64 c0, c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            const double add = (t == 0 && row < D) ? h0[row] : 0.0;
            c0[r] = h[r].x + add;
            c1[r] = h[r].y + add;
        }
        if (t > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], z0[q], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], z1[q], c1, 0, 0, 0);
            }
        }
        z0 = c0;
        z1 = c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            if (row < D)
                *reinterpret_cast<v2f64 *>(&Z[((int64_t)t * D + row) * BL + col]) = v2f64{z0[r], z1[r]};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { h[r] = hn[r]; a[r] = an[r]; }
    }
}

// The same forward sweep WITH the projection h_t = tau C^T y_t inside (M <= 16): the observations of
// a step are loaded in the B-operand layout (row m = 4 q + (l >> 4), sequences along l & 15) and
// h_t is MQ more products into the accumulator the recursion continues in -- the (T, D, BL) array of
// projected data is neither written nor read (lssm_project_kernel: 4.1 of 24.9 ms at D = 16, B = 1e5).
template <int D, int MQ>
__global__ void __launch_bounds__(256)
lssm_forward_mfma_y_kernel(const double *__restrict__ Yt, int M, int64_t B, int T, int64_t BL,
                           const double *__restrict__ Cm, const double *__restrict__ tau_ptr,
                           const double *__restrict__ h0, const double *__restrict__ J,
                           double *__restrict__ Z, int t0, int t1)
{
    const int l = threadIdx.x & 63, l15 = l & 15, l4 = l >> 4;
    const int64_t b0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
    if (b0 >= B) return;
    const int64_t col = b0 + 2 * l15;
    const v2f64 zero2 = v2f64{0.0, 0.0};
    auto load_y = [&](int t, v2f64 (&out)[MQ]) {
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            const int m = 4 * q + l4;
            out[q] = m < M ? *reinterpret_cast<const v2f64 *>(&Yt[((int64_t)t * M + m) * BL + col]) : zero2;
        }
    };
    auto load_jt = [&](int t, double (&a)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + l4;
            a[q] = (l15 < D && k < D) ? -J[(int64_t)t * D * D + k * D + l15] : 0.0;
        }
    };
    // A[i = l15][k = m = 4 q + l4] = tau C[m][i]
    const double tau = tau_ptr[0];
    double ac[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const int m = 4 * q + l4;
        ac[q] = (l15 < D && m < M) ? tau * Cm[m * D + l15] : 0.0;
    }
    v4f64 z0 = {0.0, 0.0, 0.0, 0.0}, z1 = {0.0, 0.0, 0.0, 0.0};
    if (t0 > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            const v2f64 zp = row < D ? *reinterpret_cast<const v2f64 *>(&Z[((int64_t)(t0 - 1) * D + row) * BL + col])
                                     : zero2;
            z0[r] = zp.x;
            z1[r] = zp.y;
        }
    }
    v2f64 y[MQ], yn[MQ];
    double a[4] = {0.0, 0.0, 0.0, 0.0}, an[4];
    load_y(t0, y);
    if (t0 > 0) load_jt(t0 - 1, a);
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) {
            load_y(t + 1, yn);
            load_jt(t, an);
        }
        v4f64 c0, c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            const double add = (t == 0 && row < D) ? h0[row] : 0.0;
            c0[r] = add;
            c1[r] = add;
        }
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[q], y[q].x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[q], y[q].y, c1, 0, 0, 0);
        }
        if (t > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], z0[q], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], z1[q], c1, 0, 0, 0);
            }
        }
        z0 = c0;
        z1 = c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            if (row < D)
                *reinterpret_cast<v2f64 *>(&Z[((int64_t)t * D + row) * BL + col]) = v2f64{z0[r], z1[r]};
        }
#pragma unroll
        for (int q = 0; q < MQ; ++q) y[q] = yn[q];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = an[r];
    }
}

template <int D>
__global__ void __launch_bounds__(256)
lssm_backward_mfma_kernel(int64_t B, int T, int64_t BL, const double *__restrict__ Sinv,
                          const double *__restrict__ J, double *__restrict__ Z)
{
    const int l = threadIdx.x & 63, l15 = l & 15, l4 = l >> 4;
    const int64_t b0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
    if (b0 >= B) return;
    const int64_t col = b0 + 2 * l15;
    const v2f64 zero2 = v2f64{0.0, 0.0};
    auto load_rows = [&](int t, v2f64 (&out)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            out[r] = row < D ? *reinterpret_cast<const v2f64 *>(&Z[((int64_t)t * D + row) * BL + col])
                             : zero2;
        }
    };
    // A[i = l15][k = 4 q + l4] = sign * M_t[i][k]
    auto load_a = [&](const double *M, int t, double sign, double (&a)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + l4;
            a[q] = (l15 < D && k < D) ? sign * M[(int64_t)t * D * D + l15 * D + k] : 0.0;
        }
    };
    v4f64 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
    v2f64 z[4], zn[4];
    double as[4], aj[4] = {0.0, 0.0, 0.0, 0.0}, asn[4], ajn[4];
    load_rows(T - 1, z);
    load_a(Sinv, T - 1, 1.0, as);
    for (int t = T - 1; t >= 0; --t) {
        if (t > 0) {
            load_rows(t - 1, zn);
            load_a(Sinv, t - 1, 1.0, asn);
            load_a(J, t - 1, -1.0, ajn);
        }
        v4f64 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(as[q], z[q].x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(as[q], z[q].y, c1, 0, 0, 0);
        }
        if (t < T - 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[q], x0[q], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[q], x1[q], c1, 0, 0, 0);
            }
        }
        x0 = c0;
        x1 = c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            if (row < D)
                *reinterpret_cast<v2f64 *>(&Z[((int64_t)t * D + row) * BL + col]) = v2f64{x0[r], x1[r]};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { z[r] = zn[r]; as[r] = asn[r]; aj[r] = ajn[r]; }
    }
}

// The backward sweep WITH the plate sums (M <= 15; default, tune key lssm_fuse_stats): the smoothed
// states of a step go through a wavefront-private 16 x 16 LDS transpose per tile into the operand
// layout (state along l & 15, sequence along l >> 4), and  sum x x^T,  sum x_t+1 x_t^T,  sum [y ; 1] x^T
// accumulate in three matrix-core tiles per wavefront -- the pass of lssm_stats_wave_kernel over the
// states and the data (24 of 80 doubles per sequence and step, 4.1 of 16 ms at D = 16, B = 1e5) is
// replaced by a read of y here (8).  One partial block (48 x 16) per wavefront, the layout of the
// statistics kernels: combined by lssm_stats_reduce_kernel.  Columns >= B contribute zeros.
template <int D>
__global__ void __launch_bounds__(256)
lssm_backward_mfma_stats_kernel(int64_t B, int T, int64_t BL, const double *__restrict__ Sinv,
                                const double *__restrict__ J, double *__restrict__ Z,
                                const double *__restrict__ Yt, int M, double *__restrict__ P)
{
    constexpr int LT = 17;
    __shared__ double Ts[4][2][16 * LT];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, l15 = l & 15, l4 = l >> 4;
    const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + w;
    const int64_t b0 = wv * 32;
    if (b0 >= B) return;
    const int64_t col = b0 + 2 * l15;
    const v2f64 zero2 = v2f64{0.0, 0.0};
    auto load_rows = [&](int t, v2f64 (&out)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            out[r] = row < D ? *reinterpret_cast<const v2f64 *>(&Z[((int64_t)t * D + row) * BL + col])
                             : zero2;
        }
    };
    auto load_a = [&](const double *Mx, int t, double sign, double (&a)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + l4;
            a[q] = (l15 < D && k < D) ? sign * Mx[(int64_t)t * D * D + l15 * D + k] : 0.0;
        }
    };
    // [y ; 1] of step t in the A-operand layout: row m = l15, sequences b0 + 2 (4 q + l4) (+ 1)
    auto load_y = [&](int t, v2f64 (&out)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t c = b0 + 2 * (4 * q + l4);
            out[q] = l15 < M ? *reinterpret_cast<const v2f64 *>(&Yt[((int64_t)t * M + l15) * BL + c])
                             : (l15 == M ? v2f64{1.0, 1.0} : zero2);
        }
    };
    bool live0[4], live1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t c = b0 + 2 * (4 * q + l4);
        live0[q] = c < B;
        live1[q] = c + 1 < B;
    }
    v4f64 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
    v4f64 sxx = x0, snp = x0, syx = x0;
    double xn0[4] = {0.0, 0.0, 0.0, 0.0}, xn1[4] = {0.0, 0.0, 0.0, 0.0};     // x_t+1, operand layout
    v2f64 z[4], zn[4], ya[4], yan[4];
    double as[4], aj[4] = {0.0, 0.0, 0.0, 0.0}, asn[4], ajn[4];
    load_rows(T - 1, z);
    load_a(Sinv, T - 1, 1.0, as);
    load_y(T - 1, ya);
    double *t0s = &Ts[w][0][0], *t1s = &Ts[w][1][0];
    for (int t = T - 1; t >= 0; --t) {
        if (t > 0) {
            load_rows(t - 1, zn);
            load_a(Sinv, t - 1, 1.0, asn);
            load_a(J, t - 1, -1.0, ajn);
            load_y(t - 1, yan);
        }
        v4f64 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(as[q], z[q].x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(as[q], z[q].y, c1, 0, 0, 0);
        }
        if (t < T - 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[q], x0[q], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[q], x1[q], c1, 0, 0, 0);
            }
        }
        x0 = c0;
        x1 = c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r;
            if (row < D)
                *reinterpret_cast<v2f64 *>(&Z[((int64_t)t * D + row) * BL + col]) = v2f64{x0[r], x1[r]};
            // (rows >= D of the accumulators are zero: zero A rows)
            t0s[row * LT + l15] = x0[r];
            t1s[row * LT + l15] = x1[r];
        }
        double xt0[4], xt1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double v0 = t0s[l15 * LT + 4 * q + l4], v1 = t1s[l15 * LT + 4 * q + l4];
            xt0[q] = live0[q] ? v0 : 0.0;
            xt1[q] = live1[q] ? v1 : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sxx = __builtin_amdgcn_mfma_f64_16x16x4f64(xt0[q], xt0[q], sxx, 0, 0, 0);
            sxx = __builtin_amdgcn_mfma_f64_16x16x4f64(xt1[q], xt1[q], sxx, 0, 0, 0);
            snp = __builtin_amdgcn_mfma_f64_16x16x4f64(xn0[q], xt0[q], snp, 0, 0, 0);
            snp = __builtin_amdgcn_mfma_f64_16x16x4f64(xn1[q], xt1[q], snp, 0, 0, 0);
            syx = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[q].x, xt0[q], syx, 0, 0, 0);
            syx = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[q].y, xt1[q], syx, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xn0[q] = xt0[q];
            xn1[q] = xt1[q];
            z[q] = zn[q];
            as[q] = asn[q];
            aj[q] = ajn[q];
            ya[q] = yan[q];
        }
    }
    double *Pb = P + wv * (48 * 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Pb[(l4 + 4 * r) * 16 + l15] = sxx[r];
        Pb[(16 + l4 + 4 * r) * 16 + l15] = snp[r];
        Pb[(32 + l4 + 4 * r) * 16 + l15] = syx[r];
    }
}

// The plate sums of the big-state path ON THE MATRIX CORES (default; tune key lssm_big_mfma).  Per
// (time step, 32 sequences) a tile [x_t (16 rows) ; x_t+1 (16) ; y_t and a row of ones (MP)] x 32
// columns is staged in LDS (row stride 34: both operand patterns below are bank-conflict free) and
//     S += tile * x_t^T        (contraction over the 32 sequences, eight k-steps)
// accumulates sum x x^T (row tile 0), sum x_t+1 x_t^T (1), sum y x^T and sum x (2 ...) in the
// accumulators of the four wavefronts, which stay resident over all tiles of the workgroup.  Columns
// >= B contribute zeros.  Three launches per update: all steps; t = 0 (x_0 x_0^T, sum x_0); t = T - 1.
// Partial block of a workgroup: (32 + MP) x 16 doubles, combined in fixed order by lssm_stats_reduce_kernel.
constexpr int STN = 32, STZ = STN + 2;
template <int D>
__global__ void __launch_bounds__(256)
lssm_stats_mfma_kernel(const double *__restrict__ Z, const double *__restrict__ Yt, int M, int MP,
                       int64_t B, int T, int T0, int T1, int64_t BL, double *__restrict__ P)
{
    extern __shared__ double Zs[];                 // (32 + MP) x STZ
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, l15 = l & 15, l4 = l >> 4;
    const int RZ = 32 + MP, NT2 = RZ / 16;         // row tiles
    const int64_t nbt = (B + STN - 1) / STN;
    const int64_t ntile = (int64_t)(T1 - T0) * nbt;
    constexpr int MAXR = 2;                        // row tiles per wavefront: NT2 <= 8
    v4f64 acc[MAXR];
#pragma unroll
    for (int m = 0; m < MAXR; ++m) acc[m] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int lrow = tid >> 4, lcol = (tid & 15) * 2;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int t = T0 + (int)(tile / nbt);
        const int64_t b0 = (tile % nbt) * STN;
        const int64_t c = b0 + lcol;
        __syncthreads();                           // (the previous tile's operands have been read)
        for (int r = lrow; r < RZ; r += 16) {
            v2f64 v = v2f64{0.0, 0.0};
            const double *src = nullptr;
            if (r < 16) {
                if (r < D) src = Z + ((int64_t)t * D + r) * BL + c;
            } else if (r < 32) {
                if (r - 16 < D && t + 1 < T) src = Z + ((int64_t)(t + 1) * D + (r - 16)) * BL + c;
            } else if (r - 32 < M) {
                src = Yt + ((int64_t)t * M + (r - 32)) * BL + c;
            }
            if (src) v = *reinterpret_cast<const v2f64 *>(src);
            else if (r - 32 == M) v = v2f64{1.0, 1.0};
            if (c >= B) v.x = 0.0;
            if (c + 1 >= B) v.y = 0.0;
            *reinterpret_cast<v2f64 *>(&Zs[r * STZ + lcol]) = v;
        }
        __syncthreads();
        const double *zb = Zs + l15 * STZ + l4;            // x_t rows: the B operand
#pragma unroll
        for (int q = 0; q < STN / 4; ++q) {
            const double b = zb[4 * q];
#pragma unroll
            for (int m = 0; m < MAXR; ++m) {
                const int rt = w + 4 * m;
                if (rt < NT2) {
                    const double a = Zs[(rt * 16 + l15) * STZ + 4 * q + l4];
                    acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
                }
            }
        }
    }
    double *Pb = P + (int64_t)blockIdx.x * RZ * 16;
#pragma unroll
    for (int m = 0; m < MAXR; ++m) {
        const int rt = w + 4 * m;
        if (rt < NT2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Pb[(rt * 16 + l4 + 4 * r) * 16 + l15] = acc[m][r];
        }
    }
}

// The same sums, one WAVEFRONT per (32 sequences, chunk of TC time steps): no workgroup barrier, the
// x_t+1 rows of a step stay in LDS as the x_t rows of the next (the smoothed states are read ONCE, not
// twice), the rows of step t + 1 are in flight in registers while step t is on the matrix cores.  A
// workgroup IS one wavefront (its barriers are waits on the LDS counter); the accumulators stay
// resident over the jobs of a wavefront; same partial layout as above.
constexpr int STY = 16;           // y rows prefetched in registers (more rows: loaded at the step)
template <int D>
__global__ void __launch_bounds__(64)
lssm_stats_wave_kernel(const double *__restrict__ Z, const double *__restrict__ Yt, int M, int MP,
                       int64_t B, int T, int T0, int T1, int TC, int64_t BL, double *__restrict__ P)
{
    extern __shared__ double Zs[];                 // x ping [16] | x pong [16] | y and ones [MP], stride STZ
    const int l = threadIdx.x, l15 = l & 15, l4 = l >> 4, lcol = l15 * 2;
    const int NY = MP / 16;                        // row tiles of [y ; 1]
    const int64_t nbt = (B + STN - 1) / STN;
    const int nch = (T1 - T0 + TC - 1) / TC;
    const int64_t njob = (int64_t)nch * nbt;
    constexpr int XR = (D + 3) / 4;                // row groups (4 rows per load) of a state tile
    constexpr int MAXY = 5;                        // M + 1 <= 80
    v4f64 axx = v4f64{0.0, 0.0, 0.0, 0.0}, axn = axx, ay[MAXY];
#pragma unroll
    for (int m = 0; m < MAXY; ++m) ay[m] = axx;
    double *Yl = Zs + 32 * STZ;              // state tile k at Zs + k * 16 * STZ
    // rows that are never loaded: zero once (D .. 15 of both state tiles; M .. MP - 1 of the y tile)
    for (int r = l4; r < 32 + MP; r += 4) {
        const bool dead = r < 32 ? (r & 15) >= D : r - 32 >= M;
        if (dead) *reinterpret_cast<v2f64 *>(&Zs[r * STZ + lcol]) = v2f64{0.0, 0.0};
    }
    const v2f64 zero2 = v2f64{0.0, 0.0};
    for (int64_t job = blockIdx.x; job < njob; job += gridDim.x) {
        const int ch = (int)(job / nbt);
        const int64_t c = (job - (int64_t)ch * nbt) * STN + lcol;
        const int ta = T0 + ch * TC, tb = ta + TC < T1 ? ta + TC : T1;
        const bool ok0 = c < B, ok1 = c + 1 < B;
        auto mask = [&](v2f64 v) { if (!ok0) v.x = 0.0; if (!ok1) v.y = 0.0; return v; };
        auto load_x = [&](int t, v2f64 (&out)[XR]) {
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int row = 4 * i + l4;
                out[i] = (row < D && t < T && ok0)
                             ? *reinterpret_cast<const v2f64 *>(&Z[((int64_t)t * D + row) * BL + c]) : zero2;
            }
        };
        auto load_y = [&](int t, v2f64 (&out)[STY / 4]) {
#pragma unroll
            for (int i = 0; i < STY / 4; ++i) {
                const int row = 4 * i + l4;
                out[i] = (row < M && ok0)
                             ? *reinterpret_cast<const v2f64 *>(&Yt[((int64_t)t * M + row) * BL + c]) : zero2;
            }
        };
        auto store_x = [&](double *dst, const v2f64 (&v)[XR]) {
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int row = 4 * i + l4;
                if (row < D) *reinterpret_cast<v2f64 *>(&dst[row * STZ + lcol]) = mask(v[i]);
            }
        };
        v2f64 xr[XR], yr[STY / 4];
        __syncthreads();                           // (the previous job's operands have been read)
        load_x(ta, xr);
        store_x(Zs, xr);
        if (l4 == 0) *reinterpret_cast<v2f64 *>(&Yl[M * STZ + lcol]) = mask(v2f64{1.0, 1.0});
        load_x(ta + 1, xr);
        load_y(ta, yr);
        for (int t = ta; t < tb; ++t) {
            const int cur = (t - ta) & 1;
            store_x(Zs + (cur ^ 1) * 16 * STZ, xr);
#pragma unroll
            for (int i = 0; i < STY / 4; ++i) {
                const int row = 4 * i + l4;
                if (row < M) *reinterpret_cast<v2f64 *>(&Yl[row * STZ + lcol]) = mask(yr[i]);
            }
            for (int row = STY + l4; row < M; row += 4)
                *reinterpret_cast<v2f64 *>(&Yl[row * STZ + lcol]) =
                    mask(ok0 ? *reinterpret_cast<const v2f64 *>(&Yt[((int64_t)t * M + row) * BL + c]) : zero2);
            if (t + 1 < tb) {                      // in flight while this step multiplies
                load_x(t + 2, xr);
                load_y(t + 1, yr);
            }
            __syncthreads();
            const double *xc = Zs + (cur * 16 + l15) * STZ + l4, *xn = Zs + ((cur ^ 1) * 16 + l15) * STZ + l4;
            const double *yl = Yl + l15 * STZ + l4;
#pragma unroll
            for (int q = 0; q < STN / 4; ++q) {
                const double b = xc[4 * q];
                axx = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, axx, 0, 0, 0);
                axn = __builtin_amdgcn_mfma_f64_16x16x4f64(xn[4 * q], b, axn, 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MAXY; ++m)
                    if (m < NY)
                        ay[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(yl[m * 16 * STZ + 4 * q], b, ay[m], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    double *Pb = P + (int64_t)blockIdx.x * (32 + MP) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Pb[(l4 + 4 * r) * 16 + l15] = axx[r];
        Pb[(16 + l4 + 4 * r) * 16 + l15] = axn[r];
    }
#pragma unroll
    for (int m = 0; m < MAXY; ++m)
        if (m < NY) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Pb[(32 + m * 16 + l4 + 4 * r) * 16 + l15] = ay[m][r];
        }
}

// out[r * D + c] = sum over the workgroups' partial blocks of P[g][(row0 + r) * 16 + c], r < R, c < D
__global__ void __launch_bounds__(NT)
lssm_stats_reduce_kernel(const double *__restrict__ P, int ng, int pstride, int row0, int R, int D,
                         double *__restrict__ out)
{
    const int e = blockIdx.x * (NT / 16) + (threadIdx.x >> 4), j = threadIdx.x & 15;
    const bool ok = e < R * D;
    const int r = ok ? e / D : 0, c = ok ? e - r * D : 0;
    const int64_t src = (int64_t)(row0 + r) * 16 + c;
    double s0 = 0.0, s1 = 0.0;
    if (ok) {
        int g = j;
        for (; g + 16 < ng; g += 32) {
            s0 += P[(int64_t)g * pstride + src];
            s1 += P[(int64_t)(g + 16) * pstride + src];
        }
        if (g < ng) s0 += P[(int64_t)g * pstride + src];
    }
    double v = s0 + s1;
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if (ok && j == 0) out[e] = v;
}

constexpr int PSR = 8;            // rows of A per workgroup row of the pair-sum pass

template <int D>
__global__ void __launch_bounds__(SNT)
lssm_pairsum_kernel(const double *__restrict__ A, int R, int dt, int64_t B, int T0, int T1,
                    int64_t BL, const double *__restrict__ Z, double *__restrict__ partial, int RP)
{
    __shared__ double red[SNT / 64];
    const int64_t b = (int64_t)blockIdx.x * SNT + threadIdx.x;
    const bool live = b < B;
    const int64_t bb = live ? b : 0;
    const int r0 = blockIdx.y * PSR;
    double acc[PSR][D];
#pragma unroll
    for (int j = 0; j < PSR; ++j)
#pragma unroll
        for (int i = 0; i < D; ++i) acc[j][i] = 0.0;
    for (int t = T1 - 1; t >= T0; --t) {              // time descending, like the sweeps' sums
        double x[D], a[PSR];
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = Z[((int64_t)t * D + i) * BL + bb];
#pragma unroll
        for (int j = 0; j < PSR; ++j)
            a[j] = (r0 + j < R) ? (A ? A[((int64_t)(t + dt) * R + r0 + j) * BL + bb] : 1.0) : 0.0;
        if (live) {
#pragma unroll
            for (int j = 0; j < PSR; ++j)
#pragma unroll
                for (int i = 0; i < D; ++i) acc[j][i] += a[j] * x[i];
        }
    }
    double *pb = partial + (int64_t)blockIdx.x * RP * D;
#pragma unroll
    for (int j = 0; j < PSR; ++j)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double v = block_sum<SNT>(acc[j][i], red);
            if (threadIdx.x == 0) pb[(r0 + j) * D + i] = v;
        }
}

// <x_bt> <- R <x_bt> with R in LDS (D > 8: D^2 uniform registers do not exist)
template <int D>
__global__ void __launch_bounds__(NT)
lssm_rotate_big_kernel(const double *__restrict__ R, int T, int64_t B, int64_t BL, double *__restrict__ Z)
{
    __shared__ double rs[D * D];
    for (int e = threadIdx.x; e < D * D; e += NT) rs[e] = R[e];
    __syncthreads();
    const int64_t total = (int64_t)T * B;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int64_t t = e / B, b = e - t * B;
        double *zp = Z + t * D * BL + b;
        double x[D];
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = zp[(int64_t)i * BL];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) s += rs[i * D + j] * x[j];
            zp[(int64_t)i * BL] = s;
        }
    }
}

// the 256-thread form of stationary(): all elements of the two iterates agree within 8 ulp of the
// largest magnitude.  red: 8 doubles of LDS; every thread of the workgroup calls it.
__device__ inline bool stationary_block(double a, double b, bool act, double *red, double ulps)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    double m = act ? fabs(a) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if (l == 0) red[w] = m;
    __syncthreads();
    m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const bool ok = __all(!act || fabs(a - b) <= ulps * 2.220446049250313e-16 * m);
    if (l == 0) red[4 + w] = ok ? 1.0 : 0.0;
    __syncthreads();
    const bool all = red[4] != 0.0 && red[5] != 0.0 && red[6] != 0.0 && red[7] != 0.0;
    __syncthreads();
    return all;
}

// The shared covariance recursion for 8 < D <= 16: one workgroup of 256 threads, thread (i, j) owns
// element (i, j) of the 16 x 16 padded matrices (identity in the padding, so the pivots there are 1),
// operands through LDS -- the lane-crossing moves of lssm_cov_kernel do not reach across the four
// wavefronts.  Same recursion, same outputs (S^-1, J, the sums of V_t and Cov(x_t, x_t+1), log|Phi|),
// the same stationarity rule as lssm_cov_kernel (tested every eighth step; the interior steps behind
// a converged iterate are filled in, not recomputed); phase bit 0 = forward half, bit 1 = backward
// half (each whole in one launch).  ~4 us per computed forward step, ~1 us per backward step.
__global__ void __launch_bounds__(256)
lssm_cov_big_kernel(cov_args a, int phase, int t0, int t1)
{
    constexpr int P = 16, LP = 17;
    __shared__ double U[2][P * LP];
    __shared__ double Es[P * LP], Js[P * LP], Vs[P * LP];
    __shared__ double red[8];
    const int tid = threadIdx.x, i = tid >> 4, j = tid & 15, D = a.D, T = a.T;
    const bool act = i < D && j < D;
    const int l = i * D + j, q = i * LP + j;
    const double pad = (i == j) ? 1.0 : 0.0;
    const double e = act ? a.E[l] : 0.0;
    const double dgm = act ? a.Dgm[l] : pad, dgT = act ? a.DgT[l] : pad;
    const int DD = D * D;
    const bool shortcut = a.shortcut != 0;
    const double ulps = (double)a.shortcut;
    int fix_from = -1;                     // steps fix_from .. T-2 share one (S^-1, J)
    Es[q] = e;
    __syncthreads();
    if (phase & 1) {
        double prod = 1.0, ex = 0.0;
        int bad = 0;
        double s = act ? a.Dg0[l] : pad;
        if (t0 > 0) {
            // a later segment (see lssm_cov_kernel): state left by the one before; nothing to do
            // when an earlier segment found the stationary stretch and finished the recursion
            if (a.sums[5 * DD + 6] != 0.0) return;
            s = act ? a.sums[4 * DD + l] : pad;
            prod = a.sums[5 * DD + 4];
            ex = a.sums[5 * DD + 5];
            bad = (int)a.sums[5 * DD + 1];
            fix_from = (int)a.sums[5 * DD + 2];
        }
        int tend = t1 < T ? t1 : T;
        for (int t = t0; t < tend; ++t) {
            int cur = 0;
            const double prod0 = prod, ex0 = ex;
            U[0][q] = s;
            __syncthreads();
            double v = s;
            for (int p = 0; p < D; ++p) {
                const double *M = U[cur];
                const double piv = M[p * LP + p], ci = M[i * LP + p], rj = M[p * LP + j];
                if (!(piv > 0.0)) bad = 1;
                const double pq = prod * piv;
                ex += (double)__builtin_amdgcn_frexp_exp(pq);
                prod = __builtin_amdgcn_frexp_mant(pq);
                const double d = 1.0 / piv;
                if (i == p) v = (j == p) ? d : rj * d;
                else if (j == p) v = -ci * d;
                else v = v - ci * rj * d;
                U[cur ^ 1][q] = v;
                __syncthreads();
                cur ^= 1;
            }
            if (act) a.Sinv[(int64_t)t * DD + l] = v;
            if (t < T - 1) {
                const double *Si = U[cur];
                double jt = 0.0;
                for (int k = 0; k < D; ++k) jt += Si[i * LP + k] * Es[k * LP + j];        // S^-1 E
                if (act) a.J[(int64_t)t * DD + l] = jt;
                Js[q] = jt;
                __syncthreads();
                double ej = 0.0;
                for (int k = 0; k < D; ++k) ej += Es[k * LP + i] * Js[k * LP + j];         // E^T J
                const double snew = ((t + 1 < T - 1) ? dgm : dgT) - ej;
                if (shortcut && t >= 1 && (t & 7) == 0 && t + 1 < T - 1 && stationary_block(snew, s, act, red, ulps)) {
                    const int tl = T - 2;                               // last interior step
                    for (int tt = t + 1; tt <= tl; ++tt) {
                        if (act) {
                            a.Sinv[(int64_t)tt * DD + l] = v;
                            a.J[(int64_t)tt * DD + l] = jt;
                        }
                    }
                    // (tl - t) more copies of this step's pivots
                    ex += (double)(tl - t) * ((ex - ex0) + (log2(prod) - log2(prod0)));
                    fix_from = t;
                    s = act ? dgT - ej : pad;                           // S_T-1
                    t = tl;
                    tend = T;                       // this launch finishes the recursion
                    continue;
                }
                s = snew;
            }
        }
        if (act) a.sums[4 * DD + l] = s;                    // state for the next segment
        if (tid == 0) {
            a.sums[5 * DD + 0] = log(prod) + ex * 0.69314718055994530942;
            a.sums[5 * DD + 1] = (double)bad;
            a.sums[5 * DD + 2] = (double)fix_from;
            a.sums[5 * DD + 4] = prod;
            a.sums[5 * DD + 5] = ex;
            a.sums[5 * DD + 6] = (tend >= T) ? 1.0 : 0.0;
        }
        if (!(phase & 2)) return;
        __threadfence();
        __syncthreads();
    } else {
        fix_from = (int)a.sums[5 * DD + 2];                 // left by the forward launch
    }
    // ---- backward: V_T-1 = S_T-1^-1;  C_t = -J_t V_t+1;  V_t = S_t^-1 - C_t J_t^T -----------------
    double v = act ? a.Sinv[(int64_t)(T - 1) * DD + l] : 0.0;
    double sv = v, sc = 0.0;
    const double vlast = v;
    double vprev = 0.0, cprev = 0.0;
    int bfix = -1, have_prev = 0;
    double jn = (T >= 2 && act) ? a.J[(int64_t)(T - 2) * DD + l] : 0.0;
    double sn = (T >= 2 && act) ? a.Sinv[(int64_t)(T - 2) * DD + l] : 0.0;
    for (int t = T - 2; t >= 0; --t) {
        const double jt = jn, si = sn;
        if (t > 0) {                                   // next step's operands: off the serial path
            jn = act ? a.J[(int64_t)(t - 1) * DD + l] : 0.0;
            sn = act ? a.Sinv[(int64_t)(t - 1) * DD + l] : 0.0;
        }
        // same (S^-1, J) as the step before and a V_t+1 that has stopped moving: the same V, C again,
        // down to the first step of the stationary stretch (tested every fourth step)
        if (shortcut && have_prev && fix_from >= 0 && t >= fix_from && t + 1 <= T - 2 && (t & 3) == 0
            && stationary_block(v, vprev, act, red, ulps)) {
            const double cnt = (double)(t - fix_from + 1);
            sv += cnt * v;
            sc += cnt * cprev;
            bfix = t;
            t = fix_from;
            if (t > 0) {
                jn = act ? a.J[(int64_t)(t - 1) * DD + l] : 0.0;
                sn = act ? a.Sinv[(int64_t)(t - 1) * DD + l] : 0.0;
            }
            continue;
        }
        Js[q] = jt;
        Vs[q] = v;
        __syncthreads();
        double c = 0.0;
        for (int k = 0; k < D; ++k) c -= Js[i * LP + k] * Vs[k * LP + j];                  // -J V
        U[0][q] = c;
        __syncthreads();
        double cj = 0.0;
        for (int k = 0; k < D; ++k) cj += U[0][i * LP + k] * Js[j * LP + k];               // C J^T
        vprev = v;
        v = si - cj;
        have_prev = 1;
        cprev = c;
        sv += v;
        sc += c;
        __syncthreads();
    }
    if (act) {
        a.sums[0 * DD + l] = sv;
        a.sums[1 * DD + l] = v;         // V_0
        a.sums[2 * DD + l] = vlast;
        a.sums[3 * DD + l] = sc;
    }
    if (tid == 0) a.sums[5 * DD + 3] = (double)bfix;       // diagnostics, backward map
}

// ---------------------------------------------------------------------------------------------
// The same recursion for 8 < D <= 16 by ONE wavefront on the matrix cores (default; tune key
// lssm_cov_mfma).  A 16 x 16 (padded) matrix lives in one accumulator of v_mfma_f64_16x16x4_f64:
// register r of lane l holds element (4 r + (l >> 4), l & 15).  Register p of a SYMMETRIC matrix is
// then at once the B operand "rows 4p .. 4p+3" and the A operand "columns 4p .. 4p+3", so the
// symmetric sweep operator with 4 x 4 block pivots needs no lane moves:
//     W = M_PP^-1;  T = W M_P.;  M <- M - M_.P T;  M_.P <- M_.P W;  M_P. <- T;  M_PP <- -W
// is three matrix instructions per pivot block (after the four, M = -S^-1), and J = S^-1 E,
// E^T J, G = V J^T, J G are four each with E (accumulator layout) and J ("transposed" layout, read
// back from the output array) as ready operands.  The 4 x 4 pivot block is read into scalars
// (v_readlane) and inverted by every lane (2 x 2 block elimination, two reciprocals).
// ~1.2 us per forward step against ~5 of the 256-thread form, ~0.2 against ~1 per backward step.
__device__ __forceinline__ double rdlane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int P>
__device__ __forceinline__ void sweep_block16(v4f64 &m, int l4, int l15, double &prod, double &ex, int &bad)
{
    const double R = m[P];
    constexpr int c0 = 4 * P;
    const double k00 = rdlane(R, c0), k10 = rdlane(R, 16 + c0), k11 = rdlane(R, 16 + c0 + 1);
    const double k20 = rdlane(R, 32 + c0), k21 = rdlane(R, 32 + c0 + 1), k22 = rdlane(R, 32 + c0 + 2);
    const double k30 = rdlane(R, 48 + c0), k31 = rdlane(R, 48 + c0 + 1), k32 = rdlane(R, 48 + c0 + 2);
    const double k33 = rdlane(R, 48 + c0 + 3);
    // [[A, Bt^T], [Bt, C]] with 2 x 2 blocks
    const double detA = k00 * k11 - k10 * k10;
    const double rA = fast_recip(detA);
    const double a00 = k11 * rA, a10 = -k10 * rA, a11 = k00 * rA;          // A^-1
    const double x00 = k20 * a00 + k21 * a10, x01 = k20 * a10 + k21 * a11;  // X = Bt A^-1
    const double x10 = k30 * a00 + k31 * a10, x11 = k30 * a10 + k31 * a11;
    const double s00 = k22 - (x00 * k20 + x01 * k21);                       // S = C - X Bt^T
    const double s10 = k32 - (x10 * k20 + x11 * k21);
    const double s11 = k33 - (x10 * k30 + x11 * k31);
    const double detS = s00 * s11 - s10 * s10;
    const double rS = fast_recip(detS);
    const double w22 = s11 * rS, w32 = -s10 * rS, w33 = s00 * rS;          // S^-1
    const double w20 = -(w22 * x00 + w32 * x10), w21 = -(w22 * x01 + w32 * x11);   // -S^-1 X
    const double w30 = -(w32 * x00 + w33 * x10), w31 = -(w32 * x01 + w33 * x11);
    const double w00 = a00 - (x00 * w20 + x10 * w30);                       // A^-1 - X^T W21
    const double w10 = a10 - (x01 * w20 + x11 * w30);
    const double w11 = a11 - (x01 * w21 + x11 * w31);
    if (!(k00 > 0.0) || !(detA > 0.0) || !(s00 > 0.0) || !(detS > 0.0)) bad = 1;
    {
        const double pq = prod * (detA * detS);
        ex += (double)__builtin_amdgcn_frexp_exp(pq);
        prod = __builtin_amdgcn_frexp_mant(pq);
    }
    // W[a = l4][b = l15 & 3]
    const int b = l15 & 3;
    const double r0 = l4 == 0 ? w00 : (l4 == 1 ? w10 : (l4 == 2 ? w20 : w30));
    const double r1 = l4 == 0 ? w10 : (l4 == 1 ? w11 : (l4 == 2 ? w21 : w31));
    const double r2 = l4 == 0 ? w20 : (l4 == 1 ? w21 : (l4 == 2 ? w22 : w32));
    const double r3 = l4 == 0 ? w30 : (l4 == 1 ? w31 : (l4 == 2 ? w32 : w33));
    const double wsel = b == 0 ? r0 : (b == 1 ? r1 : (b == 2 ? r2 : r3));
    const bool inP = (l15 >> 2) == P;
    const double Wpad = inP ? wsel : 0.0, Wa = l15 < 4 ? wsel : 0.0;
    const v4f64 zero = {0.0, 0.0, 0.0, 0.0};
    const v4f64 Tt = __builtin_amdgcn_mfma_f64_16x16x4f64(Wa, R, zero, 0, 0, 0);
    const double Tp = Tt[0];                                                // (W M_P.)[l4][l15]
    m = __builtin_amdgcn_mfma_f64_16x16x4f64(-R, Tp, m, 0, 0, 0);           // trailing update
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = inP ? 0.0 : m[r];                    // (cancelled to rounding)
    m = __builtin_amdgcn_mfma_f64_16x16x4f64(R, Wpad, m, 0, 0, 0);          // M_.P W
    m[P] = inP ? -wsel : Tp;
}

__device__ __forceinline__ bool stationary16(const v4f64 &a, const v4f64 &b, double ulps)
{
    double mx = 0.0, df = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mx = fmax(mx, fabs(a[r]));
        df = fmax(df, fabs(a[r] - b[r]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    return __all(df <= ulps * 2.220446049250313e-16 * mx);
}

__global__ void __launch_bounds__(64)
lssm_cov_mfma_kernel(cov_args a, int phase, int t0, int t1)
{
    const int l = threadIdx.x, l4 = l >> 4, l15 = l & 15, D = a.D, T = a.T, DD = D * D;
    const bool colok = l15 < D;
    auto ld_acc = [&](const double *src, double pad) {
        v4f64 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + l4;
            v[r] = (row < D && colok) ? src[row * D + l15] : (row == l15 ? pad : 0.0);
        }
        return v;
    };
    auto st_acc = [&](double *dst, const v4f64 &v, double sign) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + l4;
            if (row < D && colok) dst[row * D + l15] = sign * v[r];
        }
    };
    // J_t in the "transposed" layout: register q = J[l15][4 q + l4]
    auto ld_jt = [&](const double *src) {
        v4f64 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + l4;
            v[q] = (colok && k < D) ? src[l15 * D + k] : 0.0;
        }
        return v;
    };
    const bool shortcut = a.shortcut != 0;
    const double ulps = (double)a.shortcut;
    const v4f64 zero = {0.0, 0.0, 0.0, 0.0};
    int fix_from = -1;
    if (phase & 1) {
        const v4f64 e = ld_acc(a.E, 0.0), dgm = ld_acc(a.Dgm, 1.0), dgT = ld_acc(a.DgT, 1.0);
        double prod = 1.0, ex = 0.0;
        int bad = 0;
        v4f64 s = ld_acc(a.Dg0, 1.0);
        if (t0 > 0) {
            if (a.sums[5 * DD + 6] != 0.0) return;
            s = ld_acc(a.sums + 4 * DD, 1.0);
            prod = a.sums[5 * DD + 4];
            ex = a.sums[5 * DD + 5];
            bad = (int)a.sums[5 * DD + 1];
            fix_from = (int)a.sums[5 * DD + 2];
        }
        int tend = t1 < T ? t1 : T;
        for (int t = t0; t < tend; ++t) {
            const double prod0 = prod, ex0 = ex;
            v4f64 m = s;
            sweep_block16<0>(m, l4, l15, prod, ex, bad);
            if (D > 4) sweep_block16<1>(m, l4, l15, prod, ex, bad);
            if (D > 8) sweep_block16<2>(m, l4, l15, prod, ex, bad);
            if (D > 12) sweep_block16<3>(m, l4, l15, prod, ex, bad);
            v4f64 sinv;
#pragma unroll
            for (int r = 0; r < 4; ++r) sinv[r] = -m[r];
            st_acc(a.Sinv + (int64_t)t * DD, sinv, 1.0);
            if (t < T - 1) {
                v4f64 j = zero, ej = zero;
#pragma unroll
                for (int q = 0; q < 4; ++q) j = __builtin_amdgcn_mfma_f64_16x16x4f64(sinv[q], e[q], j, 0, 0, 0);
                st_acc(a.J + (int64_t)t * DD, j, 1.0);
#pragma unroll
                for (int q = 0; q < 4; ++q) ej = __builtin_amdgcn_mfma_f64_16x16x4f64(e[q], j[q], ej, 0, 0, 0);
                v4f64 snew;
#pragma unroll
                for (int r = 0; r < 4; ++r) snew[r] = ((t + 1 < T - 1) ? dgm[r] : dgT[r]) - ej[r];
                if (shortcut && t >= 1 && (t & 7) == 0 && t + 1 < T - 1 && stationary16(snew, s, ulps)) {
                    const int tl = T - 2;
                    for (int tt = t + 1; tt <= tl; ++tt) {
                        st_acc(a.Sinv + (int64_t)tt * DD, sinv, 1.0);
                        st_acc(a.J + (int64_t)tt * DD, j, 1.0);
                    }
                    ex += (double)(tl - t) * ((ex - ex0) + (log2(prod) - log2(prod0)));
                    fix_from = t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] = dgT[r] - ej[r];
                    t = tl;
                    tend = T;
                    continue;
                }
                s = snew;
            }
        }
        st_acc(a.sums + 4 * DD, s, 1.0);
        if (l == 0) {
            a.sums[5 * DD + 0] = log(prod) + ex * 0.69314718055994530942;
            a.sums[5 * DD + 1] = (double)bad;
            a.sums[5 * DD + 2] = (double)fix_from;
            a.sums[5 * DD + 4] = prod;
            a.sums[5 * DD + 5] = ex;
            a.sums[5 * DD + 6] = (tend >= T) ? 1.0 : 0.0;
        }
        if (!(phase & 2)) return;
        __threadfence();
    } else {
        fix_from = (int)a.sums[5 * DD + 2];
    }
    // ---- backward: V_T-1 = S_T-1^-1;  G_t = V_t+1 J_t^T (= -C_t^T);  V_t = S_t^-1 + J_t G_t ---------
    v4f64 v = ld_acc(a.Sinv + (int64_t)(T - 1) * DD, 0.0);
    v4f64 sv = v, sg = zero, vprev = zero, gprev = zero;
    const v4f64 vlast = v;
    int bfix = -1, have_prev = 0;
    v4f64 jn = zero, sn = zero;
    if (T >= 2) {
        jn = ld_jt(a.J + (int64_t)(T - 2) * DD);
        sn = ld_acc(a.Sinv + (int64_t)(T - 2) * DD, 0.0);
    }
    for (int t = T - 2; t >= 0; --t) {
        const v4f64 ja = jn, si = sn;
        if (t > 0) {
            jn = ld_jt(a.J + (int64_t)(t - 1) * DD);
            sn = ld_acc(a.Sinv + (int64_t)(t - 1) * DD, 0.0);
        }
        if (shortcut && have_prev && fix_from >= 0 && t >= fix_from && t + 1 <= T - 2 && (t & 3) == 0
            && stationary16(v, vprev, ulps)) {
            const double cnt = (double)(t - fix_from + 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sv[r] += cnt * v[r];
                sg[r] += cnt * gprev[r];
            }
            bfix = t;
            t = fix_from;
            if (t > 0) {
                jn = ld_jt(a.J + (int64_t)(t - 1) * DD);
                sn = ld_acc(a.Sinv + (int64_t)(t - 1) * DD, 0.0);
            }
            continue;
        }
        v4f64 g = zero, vn = si;
#pragma unroll
        for (int q = 0; q < 4; ++q) g = __builtin_amdgcn_mfma_f64_16x16x4f64(v[q], ja[q], g, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) vn = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[q], g[q], vn, 0, 0, 0);
        vprev = v;
        v = vn;
        have_prev = 1;
        gprev = g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sv[r] += v[r];
            sg[r] += g[r];
        }
    }
    st_acc(a.sums + 0 * DD, sv, 1.0);
    st_acc(a.sums + 1 * DD, v, 1.0);          // V_0
    st_acc(a.sums + 2 * DD, vlast, 1.0);
    // sum_t Cov(x_t, x_t+1) = -(sum_t G_t)^T
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * r + l4;
        if (row < D && colok) a.sums[3 * DD + l15 * D + row] = -sg[r];
    }
    if (l == 0) a.sums[5 * DD + 3] = (double)bfix;
}

constexpr int CKS = 4;
inline int64_t ck_bl_max(int64_t B) { return (B + 255) / 256 * 256; }
inline int64_t ws_base_doubles(int D, int M, int64_t B)
{
    const int64_t g = (B + SNT - 1) / SNT;
    const int64_t a = g * plen_of(D, M);
    const int64_t r = 256 * 8 * 2;            // relayout partials (<= num_cu * 8 workgroups)
    return (a > r ? a : r) + 64;
}

// ---------------------------------------------------------------------------------------------
// replicated-node updates and the bound: D x D / M x D algebra, one thread (a few 10^4 flops).
// Formulas: oracle/lssm.py (pinned on the live reference); reference code: GaussianARD
// gaussian.py:649-706 with the Gamma wrapper :2299-2371 and the messages of dot.py:425-633 /
// gaussian_markov_chain.py:443-475, Gamma gamma.py:116-148, bound expfamily.py:400-480.
// ---------------------------------------------------------------------------------------------
struct lssm_small_args {
    vmp_lssm_layout L;
    int D, M, T, nops;
    int ops[12];
    double B;                 // sequences, summed over ranks
    double pri[8];            // Gamma priors (a0, b0) of tau, gamma, alpha, nu
    int nu_latent;
};

// in-place inverse of the SPD matrix A (D x D) by Gauss-Jordan; returns log|A|, flags bad pivots
__device__ double serial_spd_inverse(double *A, int D, int *bad)
{
    double ld = 0.0;
    for (int p = 0; p < D; ++p) {
        const double piv = A[p * D + p];
        if (!(piv > 0.0)) *bad = 1;
        ld += log(piv);
        const double d = 1.0 / piv;
        for (int j = 0; j < D; ++j) A[p * D + j] *= d;
        A[p * D + p] = d;
        for (int i = 0; i < D; ++i) {
            if (i == p) continue;
            const double c = A[i * D + p];
            for (int j = 0; j < D; ++j) A[i * D + j] -= c * A[p * D + j];
            A[i * D + p] = -c * d;
        }
    }
    return ld;
}

__device__ inline void set_gamma(double *g, int n, int k, double a, double b)
{
    g[0 * n + k] = a;
    g[1 * n + k] = b;
    g[2 * n + k] = a / b;
    g[3 * n + k] = vmp_digamma(a) - log(b);
}

__device__ inline double gamma_term(double a0, double b0, const double *g, int n, int k)
{
    const double a = g[0 * n + k], b = g[1 * n + k];
    return (a0 * log(b0) - vmp_lgamma(a0)) - (a * log(b) - vmp_lgamma(a)) + (b - b0) * g[2 * n + k]
           + (a0 - a) * g[3 * n + k];
}

// The replicated-node algebra is a few 10^4 flops of branchy scalar code: one thread.  It runs on
// a copy of the state vector in LDS (<= 18 KB, loaded and stored back by the whole wavefront):
// on the state in global memory every one of its ~50 dependent reads was an HBM round trip
// (58 us per launch, three launches per iteration).
__device__ __noinline__ void lssm_small_body(const lssm_small_args &A, double *st, double *tmp,
                                             const double *innov_pre = nullptr)
{
    const vmp_lssm_layout &L = A.L;
    const int D = A.D, M = A.M, T = A.T, DD = D * D;
    double *tau = st + L.off_tau, *gam = st + L.off_gamma, *alp = st + L.off_alpha, *nu = st + L.off_nu;
    double *Cm = st + L.off_Cm, *CovC = st + L.off_CovC, *SCC = st + L.off_SCC;
    double *Am = st + L.off_Am, *AA = st + L.off_AA, *ldA = st + L.off_ldA;
    double *S = st + L.off_S;       // full statistics: Sxx | Spp | Snn | Snp | S00 | s0[D] | Syx[M*D]
    double *Sxx = S, *Spp = S + DD, *Snn = S + 2 * DD, *Snp = S + 3 * DD, *S00 = S + 4 * DD;
    double *s0 = S + 5 * DD, *Syx = S + 5 * DD + D;
    double *sc = st + L.off_scal;   // [0] Syy [1] log|Phi| [2] status [3] tau of the last X pass [4] log|CovC|
    int bad = 0;
    for (int oi = 0; oi < A.nops; ++oi) {
        const int op = A.ops[oi];
        if (op == VMP_LSSM_OP_STATS) {
            // mean-part sums of the smoother (already summed over ranks) + B x covariance sums
            const double *raw = st + L.off_raw, *cs = st + L.off_covsums;
            for (int e = 0; e < DD; ++e) {
                const int i = e / D, j = e % D;
                const double xx = raw[e], np_ = raw[DD + e], x0 = raw[2 * DD + e], xT = raw[3 * DD + e];
                Sxx[e] = A.B * cs[e] + xx;
                Spp[e] = A.B * (cs[e] - cs[2 * DD + e]) + xx - xT;
                Snn[e] = A.B * (cs[e] - cs[DD + e]) + xx - x0;
                // <x_t+1 x_t^T> = Cov(x_t, x_t+1)^T + means
                Snp[e] = A.B * cs[3 * DD + j * D + i] + np_;
                S00[e] = A.B * cs[DD + e] + x0;
            }
            for (int i = 0; i < D; ++i) s0[i] = raw[4 * DD + i];
            for (int e = 0; e < M * D; ++e) Syx[e] = raw[4 * DD + D + e];
            sc[1] = cs[5 * DD];
            if (cs[5 * DD + 1] != 0.0) bad = 1;
        } else if (op == VMP_LSSM_OP_C) {
            // Lam_C = diag<gamma> + <tau> Sxx (shared by all m), c_m = Cov_C <tau> Syx[m]
            for (int e = 0; e < DD; ++e) tmp[e] = tau[2] * Sxx[e];
            for (int i = 0; i < D; ++i) tmp[i * D + i] += gam[2 * D + i];
            const double ld = serial_spd_inverse(tmp, D, &bad);
            for (int e = 0; e < DD; ++e) CovC[e] = tmp[e];
            sc[4] = -ld;
            for (int m = 0; m < M; ++m)
                for (int i = 0; i < D; ++i) {
                    double s = 0.0;
                    for (int k = 0; k < D; ++k) s += CovC[i * D + k] * tau[2] * Syx[m * D + k];
                    Cm[m * D + i] = s;
                }
            for (int e = 0; e < DD; ++e) {
                const int i = e / D, j = e % D;
                double s = M * CovC[e];
                for (int m = 0; m < M; ++m) s += Cm[m * D + i] * Cm[m * D + j];
                SCC[e] = s;
            }
        } else if (op == VMP_LSSM_OP_GAMMA) {
            for (int j = 0; j < D; ++j) set_gamma(gam, D, j, A.pri[2] + 0.5 * M, A.pri[3] + 0.5 * SCC[j * D + j]);
        } else if (op == VMP_LSSM_OP_XPREP) {
            // blocks of the chain precision (gaussian_markov_chain.py:270-441) and h_0
            double *Dg0 = st + L.off_Dg, *Dgm = Dg0 + DD, *DgT = Dg0 + 2 * DD, *E = Dg0 + 3 * DD;
            double *h0 = st + L.off_h0;
            const double *Lam0 = st + L.off_Lam0, *mu0 = st + L.off_mu0;
            for (int e = 0; e < DD; ++e) {
                const int j = e / D, k = e % D;
                double anua = 0.0;
                for (int i = 0; i < D; ++i) anua += nu[2 * D + i] * AA[(i * D + j) * D + k];
                const double obs = tau[2] * SCC[e];
                const double dn = (j == k) ? nu[2 * D + j] : 0.0;
                Dg0[e] = obs + Lam0[e] + (T > 1 ? anua : 0.0);
                Dgm[e] = obs + dn + anua;
                DgT[e] = obs + (T > 1 ? dn : Lam0[e]);
                E[e] = -nu[2 * D + k] * Am[k * D + j];          // Phi[t, t+1][j][k] = -nu_k A_kj
            }
            for (int i = 0; i < D; ++i) {
                double s = 0.0;
                for (int k = 0; k < D; ++k) s += Lam0[i * D + k] * mu0[k];
                h0[i] = s;
            }
            sc[3] = tau[2];
        } else if (op == VMP_LSSM_OP_A) {
            for (int i = 0; i < D; ++i) {
                for (int e = 0; e < DD; ++e) tmp[e] = nu[2 * D + i] * Spp[e];
                for (int j = 0; j < D; ++j) tmp[j * D + j] += alp[2 * D + j];
                const double ld = serial_spd_inverse(tmp, D, &bad);
                ldA[i] = -ld;
                for (int j = 0; j < D; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < D; ++k) s += tmp[j * D + k] * nu[2 * D + i] * Snp[i * D + k];
                    Am[i * D + j] = s;
                }
                for (int j = 0; j < D; ++j)
                    for (int k = 0; k < D; ++k)
                        AA[(i * D + j) * D + k] = tmp[j * D + k] + Am[i * D + j] * Am[i * D + k];
            }
        } else if (op == VMP_LSSM_OP_ALPHA) {
            for (int j = 0; j < D; ++j) {
                double s = 0.0;
                for (int i = 0; i < D; ++i) s += AA[(i * D + j) * D + j];
                set_gamma(alp, D, j, A.pri[4] + 0.5 * D, A.pri[5] + 0.5 * s);
            }
        } else if (op == VMP_LSSM_OP_TAU || op == VMP_LSSM_OP_NU || op == VMP_LSSM_OP_ELBO) {
            // residual and innovation sums from the statistics
            double syf = 0.0, sff = 0.0;
            for (int e = 0; e < M * D; ++e) syf += Cm[e] * Syx[e];
            for (int e = 0; e < DD; ++e) sff += SCC[e] * Sxx[e];
            const double resid = sc[0] - 2.0 * syf + sff;
            double innov[DMAX];
            for (int i = 0; i < D; ++i) {
                if (innov_pre) {
                    innov[i] = innov_pre[i];
                    continue;
                }
                double s = Snn[i * D + i];
                for (int j = 0; j < D; ++j) s -= 2.0 * Am[i * D + j] * Snp[i * D + j];
                for (int j = 0; j < D; ++j)
                    for (int k = 0; k < D; ++k) s += AA[(i * D + j) * D + k] * Spp[j * D + k];
                innov[i] = s;
            }
            if (op == VMP_LSSM_OP_TAU) {
                set_gamma(tau, 1, 0, A.pri[0] + 0.5 * M * A.B * T, A.pri[1] + 0.5 * resid);
                if (!(tau[1] > 0.0)) sc[2] = (double)VMP_ERR_FLOATING;
            } else if (op == VMP_LSSM_OP_NU) {
                for (int i = 0; i < D; ++i)
                    set_gamma(nu, D, i, A.pri[6] + 0.5 * A.B * (T - 1), A.pri[7] + 0.5 * innov[i]);
            } else {
                const double LOG2PI = 1.8378770664093453;
                double *Lo = st + L.off_L;
                const double *Lam0 = st + L.off_Lam0, *mu0 = st + L.off_mu0;
                Lo[0] = M * A.B * T * (-0.5 * LOG2PI + 0.5 * tau[3]) - 0.5 * tau[2] * resid;          // Y
                double lc = M * (0.5 * sc[4] + 0.5 * D);
                for (int j = 0; j < D; ++j) lc += 0.5 * M * gam[3 * D + j] - 0.5 * gam[2 * D + j] * SCC[j * D + j];
                Lo[1] = lc;                                                                             // C
                double la = 0.5 * D * D;
                for (int i = 0; i < D; ++i) la += 0.5 * ldA[i];
                for (int j = 0; j < D; ++j) {
                    double s = 0.0;
                    for (int i = 0; i < D; ++i) s += AA[(i * D + j) * D + j];
                    la += 0.5 * D * alp[3 * D + j] - 0.5 * alp[2 * D + j] * s;
                }
                Lo[2] = la;                                                                             // A
                double lx = 0.0, slognu = 0.0;
                for (int i = 0; i < D; ++i) slognu += nu[3 * D + i];
                lx = A.B * (0.5 * T * D + 0.5 * st[L.off_ldLam0] + 0.5 * (T - 1) * slognu - 0.5 * sc[1]);
                for (int e = 0; e < DD; ++e) {
                    const int i = e / D, j = e % D;
                    lx -= 0.5 * Lam0[e] * (S00[e] - s0[i] * mu0[j] - mu0[i] * s0[j] + A.B * mu0[i] * mu0[j]);
                }
                for (int i = 0; i < D; ++i) lx -= 0.5 * nu[2 * D + i] * innov[i];
                Lo[3] = lx;                                                                             // X
                double lg = 0.0, lal = 0.0, lnu = 0.0;
                for (int j = 0; j < D; ++j) {
                    lg += gamma_term(A.pri[2], A.pri[3], gam, D, j);
                    lal += gamma_term(A.pri[4], A.pri[5], alp, D, j);
                    if (A.nu_latent) lnu += gamma_term(A.pri[6], A.pri[7], nu, D, j);
                }
                Lo[4] = lg;
                Lo[5] = lal;
                Lo[6] = gamma_term(A.pri[0], A.pri[1], tau, 1, 0);
                Lo[7] = lnu;
                Lo[8] = Lo[0] + Lo[1] + Lo[2] + Lo[3] + Lo[4] + Lo[5] + Lo[6] + Lo[7];
            }
        }
    }
    if (bad) sc[2] = (double)VMP_ERR_NOT_POSDEF;
}

// A.update() with one thread per row of A (D > 8: the D inversions of D x D matrices, 16^4 dependent
// steps of one thread, were 2.4 ms of the replicated-node chain at D = 16): thread i inverts
// nu_i Spp + diag<alpha> in its own D x D scratch and forms <a_i>, <a_i a_i^T>, log|Cov_i|.  Same
// arithmetic per row as the serial form (lssm_small_body, VMP_LSSM_OP_A).
__device__ void lssm_op_a_rows(const lssm_small_args &A, double *st, double *scratch, int tid)
{
    const vmp_lssm_layout &L = A.L;
    const int D = A.D, DD = D * D;
    if (tid >= D) return;
    const int i = tid;
    double *alp = st + L.off_alpha, *nu = st + L.off_nu;
    double *Am = st + L.off_Am, *AA = st + L.off_AA, *ldA = st + L.off_ldA;
    const double *S = st + L.off_S, *Spp = S + DD, *Snp = S + 3 * DD;
    double *tmp = scratch + (int64_t)i * DD;
    int bad = 0;
    for (int e = 0; e < DD; ++e) tmp[e] = nu[2 * D + i] * Spp[e];
    for (int j = 0; j < D; ++j) tmp[j * D + j] += alp[2 * D + j];
    const double ld = serial_spd_inverse(tmp, D, &bad);
    ldA[i] = -ld;
    for (int j = 0; j < D; ++j) {
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += tmp[j * D + k] * nu[2 * D + i] * Snp[i * D + k];
        Am[i * D + j] = s;
    }
    for (int j = 0; j < D; ++j)
        for (int k = 0; k < D; ++k)
            AA[(i * D + j) * D + k] = tmp[j * D + k] + Am[i * D + j] * Am[i * D + k];
    if (bad) st[L.off_scal + 2] = (double)VMP_ERR_NOT_POSDEF;
}

// The other heavy operations of the replicated nodes dealt over the wavefront (D > 8), element by
// element the arithmetic of lssm_small_body in the same order (bit-identical to the one-thread form):
// C.update() -- Gauss-Jordan with one row per lane --, the blocks of the chain precision, and the
// innovation sums of tau / nu / the bound (D^3 products each, three times per iteration).
__device__ void lssm_op_c_par(const lssm_small_args &A, double *st, double *tmp, int tid)
{
    const vmp_lssm_layout &L = A.L;
    const int D = A.D, M = A.M, DD = D * D;
    double *tau = st + L.off_tau, *gam = st + L.off_gamma;
    double *Cm = st + L.off_Cm, *CovC = st + L.off_CovC, *SCC = st + L.off_SCC;
    const double *S = st + L.off_S, *Sxx = S, *Syx = S + 5 * DD + D;
    double *sc = st + L.off_scal;
    for (int e = tid; e < DD; e += 64) tmp[e] = tau[2] * Sxx[e];
    __syncthreads();
    if (tid < D) tmp[tid * D + tid] += gam[2 * D + tid];
    __syncthreads();
    // serial_spd_inverse, row i on lane i
    double ld = 0.0;
    int bad = 0;
    for (int p = 0; p < D; ++p) {
        const double piv = tmp[p * D + p];
        if (!(piv > 0.0)) bad = 1;
        ld += log(piv);
        const double d = 1.0 / piv;
        __syncthreads();
        if (tid == p) {
            for (int j = 0; j < D; ++j) tmp[p * D + j] *= d;
            tmp[p * D + p] = d;
        }
        __syncthreads();
        if (tid < D && tid != p) {
            const int i = tid;
            const double c = tmp[i * D + p];
            for (int j = 0; j < D; ++j) tmp[i * D + j] -= c * tmp[p * D + j];
            tmp[i * D + p] = -c * d;
        }
        __syncthreads();
    }
    for (int e = tid; e < DD; e += 64) CovC[e] = tmp[e];
    if (tid == 0) {
        sc[4] = -ld;
        if (bad) sc[2] = (double)VMP_ERR_NOT_POSDEF;
    }
    __syncthreads();
    for (int e = tid; e < M * D; e += 64) {
        const int m = e / D, i = e % D;
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += CovC[i * D + k] * tau[2] * Syx[m * D + k];
        Cm[m * D + i] = s;
    }
    __syncthreads();
    for (int e = tid; e < DD; e += 64) {
        const int i = e / D, j = e % D;
        double s = M * CovC[e];
        for (int m = 0; m < M; ++m) s += Cm[m * D + i] * Cm[m * D + j];
        SCC[e] = s;
    }
}

__device__ void lssm_op_xprep_par(const lssm_small_args &A, double *st, int tid)
{
    const vmp_lssm_layout &L = A.L;
    const int D = A.D, T = A.T, DD = D * D;
    double *tau = st + L.off_tau, *nu = st + L.off_nu;
    const double *SCC = st + L.off_SCC, *Am = st + L.off_Am, *AA = st + L.off_AA;
    double *Dg0 = st + L.off_Dg, *Dgm = Dg0 + DD, *DgT = Dg0 + 2 * DD, *E = Dg0 + 3 * DD;
    double *h0 = st + L.off_h0;
    const double *Lam0 = st + L.off_Lam0, *mu0 = st + L.off_mu0;
    for (int e = tid; e < DD; e += 64) {
        const int j = e / D, k = e % D;
        double anua = 0.0;
        for (int i = 0; i < D; ++i) anua += nu[2 * D + i] * AA[(i * D + j) * D + k];
        const double obs = tau[2] * SCC[e];
        const double dn = (j == k) ? nu[2 * D + j] : 0.0;
        Dg0[e] = obs + Lam0[e] + (T > 1 ? anua : 0.0);
        Dgm[e] = obs + dn + anua;
        DgT[e] = obs + (T > 1 ? dn : Lam0[e]);
        E[e] = -nu[2 * D + k] * Am[k * D + j];
    }
    if (tid < D) {
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += Lam0[tid * D + k] * mu0[k];
        h0[tid] = s;
    }
    if (tid == 0) st[L.off_scal + 3] = tau[2];
}

__device__ void lssm_innov_par(const lssm_small_args &A, const double *st, double *innov, int tid)
{
    const vmp_lssm_layout &L = A.L;
    const int D = A.D, DD = D * D;
    const double *Am = st + L.off_Am, *AA = st + L.off_AA;
    const double *S = st + L.off_S, *Spp = S + DD, *Snn = S + 2 * DD, *Snp = S + 3 * DD;
    if (tid < D) {
        const int i = tid;
        double s = Snn[i * D + i];
        for (int j = 0; j < D; ++j) s -= 2.0 * Am[i * D + j] * Snp[i * D + j];
        for (int j = 0; j < D; ++j)
            for (int k = 0; k < D; ++k) s += AA[(i * D + j) * D + k] * Spp[j * D + k];
        innov[i] = s;
    }
}

__global__ void __launch_bounds__(64)
lssm_small_kernel(lssm_small_args A, double *__restrict__ gst)
{
    extern __shared__ double st_lds[];
    __shared__ double tmp[DMAX * DMAX];
    __shared__ double innov_s[DMAX];
    const int total = (int)A.L.total;
    for (int e = threadIdx.x; e < total; e += 64) st_lds[e] = gst[e];
    __syncthreads();
    if (A.D <= DREG) {
        if (threadIdx.x == 0) lssm_small_body(A, st_lds, tmp);
    } else {
        // big-state path: operation by operation, A.update() dealt over the rows (its scratch,
        // D matrices of D x D, lies behind the state copy in the dynamic LDS)
        for (int oi = 0; oi < A.nops; ++oi) {
            const int op = A.ops[oi];
            if (op == VMP_LSSM_OP_A) {
                lssm_op_a_rows(A, st_lds, st_lds + (total + 7) / 8 * 8, threadIdx.x);
            } else if (op == VMP_LSSM_OP_C) {
                lssm_op_c_par(A, st_lds, tmp, threadIdx.x);
            } else if (op == VMP_LSSM_OP_XPREP) {
                lssm_op_xprep_par(A, st_lds, threadIdx.x);
            } else {
                const bool inn = op == VMP_LSSM_OP_TAU || op == VMP_LSSM_OP_NU || op == VMP_LSSM_OP_ELBO;
                if (inn) {
                    lssm_innov_par(A, st_lds, innov_s, threadIdx.x);
                    __syncthreads();
                }
                if (threadIdx.x == 0) {
                    lssm_small_args one = A;
                    one.nops = 1;
                    one.ops[0] = op;
                    lssm_small_body(one, st_lds, tmp, inn ? innov_s : nullptr);
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += 64) gst[e] = st_lds[e];
}

inline void fill_lssm_layout(int D, int M, vmp_lssm_layout *L)
{
    int64_t o = 0;
    const int DD = D * D;
    L->off_tau = o;      o += 4;
    L->off_gamma = o;    o += 4 * D;
    L->off_alpha = o;    o += 4 * D;
    L->off_nu = o;       o += 4 * D;
    L->off_mu0 = o;      o += D;
    L->off_Lam0 = o;     o += DD;
    L->off_ldLam0 = o;   o += 1;
    L->off_Cm = o;       o += (int64_t)M * D;
    L->off_CovC = o;     o += DD;
    L->off_SCC = o;      o += DD;
    L->off_Am = o;       o += DD;
    L->off_AA = o;       o += (int64_t)DD * D;
    L->off_ldA = o;      o += D;
    L->off_Dg = o;       o += 4 * DD;
    L->off_h0 = o;       o += D;
    L->off_covsums = o;  o += 5 * DD + 8;
    L->off_raw = o;      o += plen_of(D, M);
    L->len_raw = plen_of(D, M);
    L->off_S = o;        o += 5 * DD + D + (int64_t)M * D;
    L->off_scal = o;     o += 8;
    L->off_L = o;        o += 16;
    L->total = (o + 7) / 8 * 8;
}

// ---------------------------------------------------------------------------------------------
// wide observations (M beyond what the sweeps hold in registers: M > 8, or > 16 at D <= 4).
// The sweeps need y only through  h_bt = tau C^T y_bt  (D values) and through the plate sum
// sum_bt y_bt <x_bt>^T, so they run on the PROJECTED data -- H (T, D, BL) as "observations" with
// the identity as C and tau = 1, unchanged kernels -- between a projection pass in front and a
// statistics pass behind:
//   lssm_project_kernel   H[t][i][b] = sum_m y_mbt (tau c_mi)                reads Y once
//   lssm_syx_kernel       sum_bt y_mbt <x_bt,i>, eight observed dimensions per thread, sequences
//                         over the threads: reads Y once more, <x> once per eight dimensions
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(SNT)
lssm_project_kernel(const double *__restrict__ Yt, int M, int64_t B, int T, int64_t BL,
                    const double *__restrict__ Cm, const double *__restrict__ tau_ptr,
                    double *__restrict__ H)
{
    const double tau = tau_ptr[0];
    const int64_t total = (int64_t)T * B;
    for (int64_t e = (int64_t)blockIdx.x * SNT + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * SNT) {
        const int64_t t = e / B, b = e - t * B;
        const double *yp = Yt + t * M * BL + b;
        double acc[D];
#pragma unroll
        for (int i = 0; i < D; ++i) acc[i] = 0.0;
        for (int m = 0; m < M; ++m) {
            const double y = __builtin_nontemporal_load(&yp[(int64_t)m * BL]);
#pragma unroll
            for (int i = 0; i < D; ++i) acc[i] += y * (tau * Cm[m * D + i]);
        }
#pragma unroll
        for (int i = 0; i < D; ++i) H[(t * D + i) * BL + b] = acc[i];
    }
}

constexpr int SYXM = 8;           // observed dimensions per thread of the statistics pass

template <int D>
__global__ void __launch_bounds__(SNT)
lssm_syx_kernel(const double *__restrict__ Yt, int M, int64_t B, int T, int64_t BL,
                const double *__restrict__ Z, double *__restrict__ partial, int MP)
{
    __shared__ double red[SNT / 64];
    const int64_t b = (int64_t)blockIdx.x * SNT + threadIdx.x;
    const bool live = b < B;
    const int64_t bb = live ? b : 0;
    const int m0 = blockIdx.y * SYXM;
    double acc[SYXM][D];
#pragma unroll
    for (int j = 0; j < SYXM; ++j)
#pragma unroll
        for (int i = 0; i < D; ++i) acc[j][i] = 0.0;
    // time descending, like the sums of the backward sweep
    for (int t = T - 1; t >= 0; --t) {
        double x[D], y[SYXM];
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = Z[((int64_t)t * D + i) * BL + bb];
#pragma unroll
        for (int j = 0; j < SYXM; ++j)
            y[j] = (m0 + j < M) ? __builtin_nontemporal_load(&Yt[((int64_t)t * M + m0 + j) * BL + bb])
                                : 0.0;
        if (live) {
#pragma unroll
            for (int j = 0; j < SYXM; ++j)
#pragma unroll
                for (int i = 0; i < D; ++i) acc[j][i] += y[j] * x[i];
        }
    }
    double *pb = partial + (int64_t)blockIdx.x * MP * D;
#pragma unroll
    for (int j = 0; j < SYXM; ++j)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double a = block_sum<SNT>(acc[j][i], red);
            if (threadIdx.x == 0) pb[(m0 + j) * D + i] = a;
        }
}

__global__ void __launch_bounds__(64)
lssm_identity_kernel(int D, double *ident, double *one)
{
    const int l = threadIdx.x;
    if (l < D * D) ident[l] = (l / D == l % D) ? 1.0 : 0.0;
    if (l == 0) one[0] = 1.0;
}

inline bool lssm_wide(int D, int M) { return M > 16 || (M > 8 && D > 4); }
constexpr int LSSM_MAXM = 64;
// extra workspace of the wide form behind [partial sums | checkpoints]:
//   H (T, D, BL) | identity (D x D) + one | sums of the projected sweeps | partials of the Syx pass
inline int64_t wide_extra_doubles(int D, int M, int64_t B, int T)
{
    const int64_t g = (B + SNT - 1) / SNT;
    const int64_t MP = (M + SYXM - 1) / SYXM * SYXM;
    return (int64_t)T * D * ck_bl_max(B) + 64 + 64 + (plen_of(D, D) + 63) / 64 * 64 + g * MP * D + 64;
}
inline int64_t ck_doubles(int D, int64_t B, int T)
{
    return D <= 4 ? (int64_t)((T + CKS - 1) / CKS) * D * ck_bl_max(B) : 0;
}
// big-state path (D > DREG): rows of the widest pair-sum job, padded to whole workgroup rows;
// workspace = [base | pair-sum partials g x RP x D | H (T, D, BL)]
inline int big_rows(int D, int M)
{
    const int r = D > M ? D : M;
    return (r + PSR - 1) / PSR * PSR;
}
inline int64_t big_partial_doubles(int D, int M, int64_t B)
{
    // the larger of: pair-sum partials (g x RP x D), matrix-core partials (2048 x (32 + MP) x 16)
    const int64_t g = (B + SNT - 1) / SNT;
    const int64_t a = g * big_rows(D, M) * D, m = (int64_t)2048 * (32 + 96) * 16;
    return a > m ? a : m;
}
inline int64_t big_extra_doubles(int D, int M, int64_t B, int T)
{
    return big_partial_doubles(D, M, B) + 64 + 8 + (int64_t)T * D * ck_bl_max(B) + 64;
}

}  // namespace

extern "C" {

int32_t vmp_lssm_limits(int32_t *max_D, int32_t *max_M)
{
    if (max_D) *max_D = DMAX;
    if (max_M) *max_M = LSSM_MAXM;
    return VMP_OK;
}

int32_t vmp_lssm_relayout_y(vmp_ctx *ctx, const double *Y, int32_t M, int64_t B, int32_t T,
                            int64_t BL, double *Yt, double *syy, void *workspace)
{
    VMP_REQUIRE(ctx, ctx && Y && Yt && syy && workspace, VMP_ERR_INVALID, "null argument");
    // B = 0 (no sequence on this rank of a sharded run) is legal: every sum comes out as zero
    VMP_REQUIRE(ctx, M >= 1 && B >= 0 && T >= 1 && BL >= B, VMP_ERR_INVALID, "bad dims");
    double *partial = reinterpret_cast<double *>(workspace);
    const int64_t nblk = ((B + 31) / 32) * ((T + 31) / 32) * M;
    int64_t g = nblk < (int64_t)ctx->num_cu * 8 ? nblk : (int64_t)ctx->num_cu * 8;
    if (g > 0) {
        VMP_HIP_CHECK(ctx, hipMemsetAsync(Yt, 0, (size_t)T * M * BL * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(lssm_relayout_kernel, dim3((unsigned)g), dim3(NT), 0, ctx->stream, Y, M, B,
                           T, BL, Yt, partial);
    }
    hipLaunchKernelGGL(lssm_sum_kernel, dim3(1), dim3(NT), 0, ctx->stream, partial, (int)g, 1, 1, syy);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_lssm_x_layout(vmp_ctx *ctx, double *X, int32_t D, int64_t B, int32_t T, int64_t BL,
                          double *Z, int32_t to_time_major)
{
    VMP_REQUIRE(ctx, ctx && X && Z, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, D >= 1 && B >= 0 && T >= 1 && BL >= B, VMP_ERR_INVALID, "bad dims");
    int64_t g = ((int64_t)B * T * D + NT - 1) / NT;
    if (g > (int64_t)ctx->num_cu * 16) g = (int64_t)ctx->num_cu * 16;
    if (g == 0) return VMP_OK;
    hipLaunchKernelGGL(lssm_x_layout_kernel, dim3((unsigned)g), dim3(NT), 0, ctx->stream, X, D, B, T,
                       BL, Z, to_time_major);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

static int32_t launch_cov(vmp_ctx *ctx, hipStream_t s, int phase, int32_t T, int32_t D,
                          const double *Dg0, const double *Dgm, const double *DgT, const double *E,
                          double *Sinv, double *J, double *sums, int t0 = 0, int t1 = -1)
{
    if (t1 < 0) t1 = T;
    VMP_REQUIRE(ctx, ctx && Dg0 && Dgm && DgT && E && Sinv && J && sums, VMP_ERR_INVALID,
                "null argument");
    VMP_REQUIRE(ctx, T >= 1 && D >= 1, VMP_ERR_INVALID, "bad dims");
    VMP_REQUIRE(ctx, D <= DMAX, VMP_ERR_UNSUPPORTED, "the fused LSSM block supports D <= %d", DMAX);
    cov_args a;
    a.T = T;
    a.D = D;
    a.Dg0 = Dg0;
    a.Dgm = Dgm;
    a.DgT = DgT;
    a.E = E;
    a.Sinv = Sinv;
    a.J = J;
    a.sums = sums;
    a.shortcut = vmp_tune_get("lssm_cov_shortcut", 8);     // in ulp; 0: every step computed
    if (D > DREG) {
        if (vmp_tune_get("lssm_cov_mfma", 1) != 0)
            hipLaunchKernelGGL(lssm_cov_mfma_kernel, dim3(1), dim3(64), 0, s, a, phase, t0, t1);
        else
            hipLaunchKernelGGL(lssm_cov_big_kernel, dim3(1), dim3(256), 0, s, a, phase, t0, t1);
        VMP_HIP_CHECK(ctx, hipGetLastError());
        return VMP_OK;
    }
    switch (D) {
#define LSSM_COV(d) case d: hipLaunchKernelGGL(lssm_cov_kernel<d>, dim3(1), dim3(64), 0, s, a, phase, t0, t1); break;
        LSSM_COV(1) LSSM_COV(2) LSSM_COV(3) LSSM_COV(4) LSSM_COV(5) LSSM_COV(6) LSSM_COV(7) LSSM_COV(8)
#undef LSSM_COV
    }
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_lssm_cov(vmp_ctx *ctx, int32_t T, int32_t D, const double *Dg0, const double *Dgm,
                     const double *DgT, const double *E, double *Sinv, double *J, double *sums)
{
    return launch_cov(ctx, ctx ? ctx->stream : nullptr, 3, T, D, Dg0, Dgm, DgT, E, Sinv, J, sums);
}

#define LSSM_FOR_EACH(MACRO)                                                               \
    MACRO(1, 8) MACRO(2, 8) MACRO(3, 8) MACRO(4, 8) MACRO(5, 8) MACRO(6, 8) MACRO(7, 8)    \
    MACRO(8, 8) MACRO(1, 16) MACRO(2, 16) MACRO(3, 16) MACRO(4, 16)

// nseg > 1 (vmp_lssm_x_update): the forward sweep runs in nseg time segments, segment k after
// seg_ready[k] (the covariance recursion has produced its J_t)
static int32_t smooth_impl(vmp_ctx *ctx, int32_t given, const double *Yt, int32_t M, int64_t B,
                           int32_t T, int64_t BL, int32_t D, const double *Cm, const double *tau,
                           const double *h0, const double *Sinv, const double *J, double *Z,
                           double *stats, void *workspace, int nseg, const hipEvent_t *seg_ready)
{
    VMP_REQUIRE(ctx, ctx && Yt && Z && stats && workspace, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, given || (Cm && tau && h0 && Sinv && J), VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, M >= 1 && B >= 0 && T >= 1 && BL >= B && D >= 1, VMP_ERR_INVALID, "bad dims");
    VMP_REQUIRE(ctx, D <= DMAX && M <= LSSM_MAXM, VMP_ERR_UNSUPPORTED,
                "the fused LSSM block supports D <= %d states, M <= %d observed dimensions", DMAX,
                LSSM_MAXM);
    if (lssm_big(D)) {
        // big-state path: sweeps on the projected data carrying the state only, plate sums behind
        VMP_REQUIRE(ctx, BL <= ck_bl_max(B), VMP_ERR_INVALID,
                    "leading dimension beyond the workspace contract (B rounded up to 256)");
        double *wsd = reinterpret_cast<double *>(workspace);
        double *part = wsd;                                   // pair-sum partials: g x RP x D
        const int64_t g = (B + SNT - 1) / SNT;
        const int RP = big_rows(D, M);
        double *H = wsd + (ws_base_doubles(D, M, B) + big_partial_doubles(D, M, B) + 64 + 7) / 8 * 8;
        hipStream_t sw = ctx->stream;
        bool fs = false;                                      // the backward sweep made the main sums
        if (!given && g > 0) {
            int64_t gp = ((int64_t)T * B + SNT - 1) / SNT;
            if (gp > (int64_t)ctx->num_cu * 16) gp = (int64_t)ctx->num_cu * 16;
            // on the matrix cores (default) when the arrays allow 16-byte accesses per column pair
            const bool mf = vmp_tune_get("lssm_big_mfma", 1) != 0 && BL % 32 == 0 &&
                            ((reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(H)) & 15) == 0;
            // one wavefront of 32 sequences per workgroup (tune key lssm_sweep_waves: wavefronts per
            // workgroup; four until round 6): a sweep's time is that of the busiest CU, and B = 1e5 as 782
            // workgroups over 256 CUs is three on most and FOUR on some (+31 %), as 3125 it is 12 or 13
            int swv = vmp_tune_get("lssm_sweep_waves", 1);
            if (swv < 1 || swv > 4) swv = 1;
            const unsigned sth = 64u * (unsigned)swv;
            const int64_t gm = (B + 32 * swv - 1) / (32 * swv);
            // the projection inside the sweep when the observations fit four k-steps
            const bool fy = mf && M <= 16 && vmp_tune_get("lssm_fuse_project", 1) != 0 &&
                            (reinterpret_cast<uintptr_t>(Yt) & 15) == 0;
            // the plate sums inside the backward sweep: one partial block per wavefront must fit.
            // Tune key lssm_fuse_stats: 1 always, 0 never, default (2) where it was measured to pay --
            // D >= 15 and enough sequences that the sweeps, not the covariance recursion, are the
            // critical path (D = 16: -10 % at B = 1e5, +6 % at B = 2e4; D = 12: +8 % at B = 1e5: the
            // kernel holds 198 registers, two wavefronts per SIMD, and its arithmetic is that of 16
            // padded states whatever D)
            const int fsk = vmp_tune_get("lssm_fuse_stats", 2);
            fs = mf && D > DREG && M <= 15 && (fsk == 1 || (fsk == 2 && D >= 15 && B >= 40000)) &&
                 (reinterpret_cast<uintptr_t>(Yt) & 15) == 0 &&
                 ((B + 31) / 32) * (48 * 16) <= big_partial_doubles(D, M, B);
#define LSSM_BIG(d)                                                                              \
    if (D == d) {                                                                                \
        if (!fy)                                                                                 \
            hipLaunchKernelGGL(lssm_project_kernel<d>, dim3((unsigned)gp), dim3(SNT), 0, sw, Yt, \
                               M, B, T, BL, Cm, tau, H);                                         \
        for (int k = 0; k < nseg; ++k) {                                                         \
            if (seg_ready) (void)hipStreamWaitEvent(sw, seg_ready[k], 0);                        \
            const int ta = (int)((int64_t)T * k / nseg), tb = (int)((int64_t)T * (k + 1) / nseg); \
            if (tb <= ta) continue;                                                              \
            if (fy && M <= 4)                                                                    \
                hipLaunchKernelGGL((lssm_forward_mfma_y_kernel<d, 1>), dim3((unsigned)gm),       \
                                   dim3(sth), 0, sw, Yt, M, B, T, BL, Cm, tau, h0, J, Z, ta, tb); \
            else if (fy && M <= 8)                                                               \
                hipLaunchKernelGGL((lssm_forward_mfma_y_kernel<d, 2>), dim3((unsigned)gm),       \
                                   dim3(sth), 0, sw, Yt, M, B, T, BL, Cm, tau, h0, J, Z, ta, tb); \
            else if (fy)                                                                         \
                hipLaunchKernelGGL((lssm_forward_mfma_y_kernel<d, 4>), dim3((unsigned)gm),       \
                                   dim3(sth), 0, sw, Yt, M, B, T, BL, Cm, tau, h0, J, Z, ta, tb); \
            else if (mf)                                                                         \
                hipLaunchKernelGGL(lssm_forward_mfma_kernel<d>, dim3((unsigned)gm), dim3(sth), 0, \
                                   sw, H, B, T, BL, h0, J, Z, ta, tb);                           \
            else                                                                                 \
                hipLaunchKernelGGL((lssm_forward_kernel<d, d, 0, true>), dim3((unsigned)g),      \
                                   dim3(SNT), 0, sw, H, d, B, T, BL, nullptr, nullptr, h0, J, Z, \
                                   ta, tb);                                                      \
        }                                                                                        \
        if (fs)                                                                                  \
            hipLaunchKernelGGL(lssm_backward_mfma_stats_kernel<d>, dim3((unsigned)gm), dim3(sth), \
                               0, sw, B, T, BL, Sinv, J, Z, Yt, M, part);                        \
        else if (mf)                                                                             \
            hipLaunchKernelGGL(lssm_backward_mfma_kernel<d>, dim3((unsigned)gm), dim3(sth), 0,   \
                               sw, B, T, BL, Sinv, J, Z);                                        \
        else                                                                                     \
            hipLaunchKernelGGL(lssm_backward_plain_kernel<d>, dim3((unsigned)g), dim3(SNT), 0,   \
                               sw, B, T, BL, Sinv, J, Z);                                        \
    }
            const bool regsw = D <= DREG && vmp_tune_get("lssm_split_sweeps", 0) == 0;
            // register sweeps (D <= DREG): straight from the observations when M <= 8 (no projection pass)
#define LSSM_BIGR(d)                                                                             \
    if (D == d) {                                                                                \
        const bool direct = M <= 8;                                                              \
        if (!direct)                                                                             \
            hipLaunchKernelGGL(lssm_project_kernel<d>, dim3((unsigned)gp), dim3(SNT), 0, sw, Yt, \
                               M, B, T, BL, Cm, tau, H);                                         \
        for (int k = 0; k < nseg; ++k) {                                                         \
            if (seg_ready) (void)hipStreamWaitEvent(sw, seg_ready[k], 0);                        \
            const int ta = (int)((int64_t)T * k / nseg), tb = (int)((int64_t)T * (k + 1) / nseg); \
            if (tb <= ta) continue;                                                              \
            if (direct)                                                                          \
                hipLaunchKernelGGL((lssm_forward_kernel<d, 8, 0>), dim3((unsigned)g), dim3(SNT), \
                                   0, sw, Yt, M, B, T, BL, Cm, tau, h0, J, Z, ta, tb);           \
            else                                                                                 \
                hipLaunchKernelGGL((lssm_forward_kernel<d, d, 0, true>), dim3((unsigned)g),      \
                                   dim3(SNT), 0, sw, H, d, B, T, BL, nullptr, nullptr, h0, J, Z, \
                                   ta, tb);                                                      \
        }                                                                                        \
        hipLaunchKernelGGL(lssm_backward_plain_kernel<d>, dim3((unsigned)g), dim3(SNT), 0, sw,   \
                           B, T, BL, Sinv, J, Z);                                                \
    }
            if (regsw) {
                LSSM_BIGR(7) LSSM_BIGR(8)
            } else {
            LSSM_BIG(7) LSSM_BIG(8) LSSM_BIG(9) LSSM_BIG(10) LSSM_BIG(11) LSSM_BIG(12) LSSM_BIG(13) LSSM_BIG(14)
            LSSM_BIG(15) LSSM_BIG(16)
            }
#undef LSSM_BIGR
#undef LSSM_BIG
            VMP_HIP_CHECK(ctx, hipGetLastError());
        }
        // plate sums: [0, DD) x x^T | x_t+1 x_t^T | x_0 x_0^T | x_T-1 x_T-1^T | x_0 (D) | y x^T (M x D)
        const int DD = D * D;
        const int MPs = (M + 1 + 15) / 16 * 16;             // y rows + the row of ones, whole tiles
        if (vmp_tune_get("lssm_big_mfma", 1) != 0 && BL % 32 == 0 && MPs <= 96 &&
            ((reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(Yt)) & 15) == 0) {
            const int RZ = 32 + MPs;
            const size_t lds = (size_t)RZ * STZ * sizeof(double);
            const int64_t nbt = (B + STN - 1) / STN;
            double *Pm = part;                              // <= 2048 blocks x RZ x 16 (sized below)
            struct { int t0, t1; } rng[3] = {{0, T}, {0, 1}, {T - 1, T}};
            const bool wavef = vmp_tune_get("lssm_stats_form", 1) != 0;
            for (int q = 0; q < 3; ++q) {
                int64_t gs;
                if (q == 0 && fs) {
                    gs = (B + 31) / 32;                     // the partial blocks of the backward sweep
                } else if (wavef) {
                    // one wavefront per job of (32 sequences, TC steps); as many resident as LDS holds
                    const int steps = rng[q].t1 - rng[q].t0;
                    int TC = 32;
                    while (TC > 4 && (int64_t)((steps + TC - 1) / TC) * nbt < (int64_t)ctx->num_cu * 8) TC /= 2;
                    const int64_t njob = (int64_t)((steps + TC - 1) / TC) * nbt;
                    int per_cu = (int)((size_t)150 * 1024 / lds);
                    if (per_cu > 16) per_cu = 16;
                    gs = (int64_t)ctx->num_cu * per_cu;
                    const int64_t cap = big_partial_doubles(D, M, B) / ((int64_t)RZ * 16);
                    if (gs > cap) gs = cap;
                    if (gs > njob) gs = njob;
                    if (gs < 1) gs = 1;
#define LSSM_STW(d)                                                                               \
    if (D == d)                                                                                   \
        hipLaunchKernelGGL(lssm_stats_wave_kernel<d>, dim3((unsigned)gs), dim3(64), lds, sw, Z,   \
                           Yt, M, MPs, B, T, rng[q].t0, rng[q].t1, TC, BL, Pm);
                    LSSM_STW(7) LSSM_STW(8) LSSM_STW(9) LSSM_STW(10) LSSM_STW(11) LSSM_STW(12) LSSM_STW(13)
                    LSSM_STW(14) LSSM_STW(15) LSSM_STW(16)
#undef LSSM_STW
                } else {
                gs = (int64_t)(rng[q].t1 - rng[q].t0) * nbt;
                if (gs > (int64_t)ctx->num_cu * 8) gs = (int64_t)ctx->num_cu * 8;   // (latency: 8 per CU)
                if (gs > 2048) gs = 2048;
                if (gs < 1) gs = 1;
#define LSSM_ST(d)                                                                                \
    if (D == d)                                                                                   \
        hipLaunchKernelGGL(lssm_stats_mfma_kernel<d>, dim3((unsigned)gs), dim3(256), lds, sw, Z,  \
                           Yt, M, MPs, B, T, rng[q].t0, rng[q].t1, BL, Pm);
                LSSM_ST(7) LSSM_ST(8) LSSM_ST(9) LSSM_ST(10) LSSM_ST(11) LSSM_ST(12) LSSM_ST(13) LSSM_ST(14) LSSM_ST(15)
                LSSM_ST(16)
#undef LSSM_ST
                }
                auto red = [&](int row0, int R, double *out) {
                    hipLaunchKernelGGL(lssm_stats_reduce_kernel,
                                       dim3((unsigned)((R * D + NT / 16 - 1) / (NT / 16))), dim3(NT), 0,
                                       sw, Pm, (int)gs, RZ * 16, row0, R, D, out);
                };
                if (q == 0) {
                    red(0, D, stats);                        // sum x x^T
                    red(16, D, stats + DD);                  // sum x_t+1 x_t^T
                    red(32, M, stats + 4 * DD + D);          // sum y x^T
                } else if (q == 1) {
                    red(0, D, stats + 2 * DD);               // x_0 x_0^T
                    red(32 + M, 1, stats + 4 * DD);          // sum x_0
                } else {
                    red(0, D, stats + 3 * DD);               // x_T-1 x_T-1^T
                }
            }
            VMP_HIP_CHECK(ctx, hipGetLastError());
            return VMP_OK;
        }
        struct { const double *A; int R, dt, t0, t1, off, len; } jobs[6] = {
            {Z, D, 0, 0, T, 0, DD},          {Z, D, 1, 0, T - 1, DD, DD},
            {Z, D, 0, 0, 1, 2 * DD, DD},     {Z, D, 0, T - 1, T, 3 * DD, DD},
            {nullptr, 1, 0, 0, 1, 4 * DD, D}, {Yt, M, 0, 0, T, 4 * DD + D, M * D}};
        for (int q = 0; q < 6; ++q) {
            const int yb = (jobs[q].R + PSR - 1) / PSR;
            if (g > 0) {
#define LSSM_PS(d)                                                                               \
    if (D == d)                                                                                  \
        hipLaunchKernelGGL(lssm_pairsum_kernel<d>, dim3((unsigned)g, (unsigned)yb), dim3(SNT), 0, \
                           sw, jobs[q].A, jobs[q].R, jobs[q].dt, B, jobs[q].t0, jobs[q].t1, BL,  \
                           Z, part, RP);
                LSSM_PS(7) LSSM_PS(8) LSSM_PS(9) LSSM_PS(10) LSSM_PS(11) LSSM_PS(12) LSSM_PS(13) LSSM_PS(14) LSSM_PS(15)
                LSSM_PS(16)
#undef LSSM_PS
            }
            hipLaunchKernelGGL(lssm_sum_kernel, dim3((unsigned)((jobs[q].len + 15) / 16)), dim3(NT), 0,
                               sw, part, (int)g, RP * D, jobs[q].len, stats + jobs[q].off);
        }
        VMP_HIP_CHECK(ctx, hipGetLastError());
        return VMP_OK;
    }
    if (lssm_wide(D, M)) {
        // projected data in front of the unchanged sweeps, the y <x>^T sums behind them
        VMP_REQUIRE(ctx, BL <= ck_bl_max(B), VMP_ERR_INVALID,
                    "leading dimension beyond the workspace contract (B rounded up to 256)");
        double *wsd = reinterpret_cast<double *>(workspace);
        double *H = wsd + ws_base_doubles(D, M, B) + ck_doubles(D, B, T);
        double *ident = H + (int64_t)T * D * ck_bl_max(B);
        double *one = ident + 64;
        double *rawtmp = one + 64;
        double *syxp = rawtmp + (plen_of(D, D) + 63) / 64 * 64;
        const int64_t gw = (B + SNT - 1) / SNT;
        const int MP = (M + SYXM - 1) / SYXM * SYXM;
        hipStream_t sw = ctx->stream;
        hipLaunchKernelGGL(lssm_identity_kernel, dim3(1), dim3(64), 0, sw, D, ident, one);
        if (given) {
            // the inner statistics pass still reads its "observations" for the y <x>^T sums it
            // computes and discards: give it defined values, not whatever the workspace held
            VMP_HIP_CHECK(ctx, hipMemsetAsync(H, 0, (size_t)T * D * ck_bl_max(B) * sizeof(double), sw));
        }
        if (!given && gw > 0) {
            int64_t gp = ((int64_t)T * B + SNT - 1) / SNT;
            if (gp > (int64_t)ctx->num_cu * 16) gp = (int64_t)ctx->num_cu * 16;
            switch (D) {
#define LSSM_PROJ(d) case d: hipLaunchKernelGGL(lssm_project_kernel<d>, dim3((unsigned)gp), dim3(SNT), 0, sw, Yt, M, B, T, BL, Cm, tau, H); break;
                LSSM_PROJ(1) LSSM_PROJ(2) LSSM_PROJ(3) LSSM_PROJ(4) LSSM_PROJ(5) LSSM_PROJ(6)
                LSSM_PROJ(7) LSSM_PROJ(8)
#undef LSSM_PROJ
            }
        }
        VMP_HIP_CHECK(ctx, hipGetLastError());
        const int32_t rcw = smooth_impl(ctx, given, H, D, B, T, BL, D, ident, one, h0, Sinv, J, Z,
                                        rawtmp, workspace, nseg, seg_ready);
        if (rcw != VMP_OK) return rcw;
        VMP_HIP_CHECK(ctx, hipMemcpyAsync(stats, rawtmp, (size_t)(4 * D * D + D) * sizeof(double),
                                          hipMemcpyDeviceToDevice, sw));
        if (gw > 0) {
            switch (D) {
#define LSSM_SYX(d) case d: hipLaunchKernelGGL(lssm_syx_kernel<d>, dim3((unsigned)gw, (unsigned)(MP / SYXM)), dim3(SNT), 0, sw, Yt, M, B, T, BL, Z, syxp, MP); break;
                LSSM_SYX(1) LSSM_SYX(2) LSSM_SYX(3) LSSM_SYX(4) LSSM_SYX(5) LSSM_SYX(6) LSSM_SYX(7)
                LSSM_SYX(8)
#undef LSSM_SYX
            }
        }
        hipLaunchKernelGGL(lssm_sum_kernel, dim3((unsigned)((M * D + 15) / 16)), dim3(NT), 0, sw,
                           syxp, (int)gw, MP * D, M * D, stats + 4 * D * D + D);
        VMP_HIP_CHECK(ctx, hipGetLastError());
        return VMP_OK;
    }
    const int MM = M <= 8 ? 8 : 16;
    const int64_t g = (B + SNT - 1) / SNT;
    const int plen = plen_of(D, M);
    double *partial = reinterpret_cast<double *>(workspace);
    hipStream_t s = ctx->stream;
    hipEvent_t *ev = ctx->timing ? vmp_next_events(ctx) : nullptr;
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[0], s));
    // checkpoint form (D <= 4, M <= 8: the block's y and z fit the registers): z is kept every
    // CKS steps only, in the workspace behind the partial sums (vmp_lssm_workspace_doubles)
    if (!given && g > 0 && D <= 4 && MM == 8 && T >= 2 * CKS && BL <= ck_bl_max(B)
        && vmp_tune_get("lssm_checkpoint", 1) != 0) {
        double *Zc = partial + ws_base_doubles(D, M, B);
#define LSSM_CK(d)                                                                             \
    if (D == d) {                                                                              \
        for (int k = 0; k < nseg; ++k) {                                                       \
            if (seg_ready) (void)hipStreamWaitEvent(s, seg_ready[k], 0);                       \
            const int t0 = (int)((int64_t)T * k / nseg) / CKS * CKS;                           \
            const int t1 = k + 1 < nseg ? (int)((int64_t)T * (k + 1) / nseg) / CKS * CKS : T;  \
            if (t1 > t0)                                                                       \
                hipLaunchKernelGGL((lssm_forward_kernel<d, 8, CKS>), dim3((unsigned)g),        \
                                   dim3(SNT), 0, s, Yt, M, B, T, BL, Cm, tau, h0, J, Zc, t0,   \
                                   t1);                                                        \
        }                                                                                      \
        if (ev) (void)hipEventRecord(ev[1], s);                                                \
        hipLaunchKernelGGL((lssm_backward_ck_kernel<d, 8, CKS>), dim3((unsigned)g), dim3(SNT),  \
                           0, s, Yt, M, B, T, BL, Cm, tau, h0, Sinv, J, Zc, Z, partial, plen); \
    }
        LSSM_CK(1) LSSM_CK(2) LSSM_CK(3) LSSM_CK(4)
#undef LSSM_CK
        VMP_HIP_CHECK(ctx, hipGetLastError());
        if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[2], s));
        hipLaunchKernelGGL(lssm_sum_kernel, dim3((unsigned)((plen + 15) / 16)), dim3(NT), 0, s,
                           partial, (int)g, plen, plen, stats);
        VMP_HIP_CHECK(ctx, hipGetLastError());
        return VMP_OK;
    }
#define LSSM_CASE(d, mm)                                                                      \
    if (g == 0) {                                                                             \
    } else if (D == d && MM == mm) {                                                          \
        if (!given) {                                                                         \
            for (int k = 0; k < nseg; ++k) {                                                  \
                if (seg_ready) (void)hipStreamWaitEvent(s, seg_ready[k], 0);                  \
                hipLaunchKernelGGL((lssm_forward_kernel<d, mm, 0>), dim3((unsigned)g), dim3(SNT), 0, \
                                   s, Yt, M, B, T, BL, Cm, tau, h0, J, Z,                     \
                                   (int)((int64_t)T * k / nseg), (int)((int64_t)T * (k + 1) / nseg)); \
            }                                                                                 \
        }                                                                                     \
        if (ev) (void)hipEventRecord(ev[1], s);                                               \
        hipLaunchKernelGGL((lssm_backward_kernel<d, mm>), dim3((unsigned)g), dim3(SNT), 0, s,  \
                           Yt, M, B, T, BL, Sinv, J, Z, partial, plen, given);                \
    } else
    LSSM_FOR_EACH(LSSM_CASE) { return VMP_ERR_UNSUPPORTED; }
#undef LSSM_CASE
    VMP_HIP_CHECK(ctx, hipGetLastError());
    if (ev) VMP_HIP_CHECK(ctx, hipEventRecord(ev[2], s));
    hipLaunchKernelGGL(lssm_sum_kernel, dim3((unsigned)((plen + 15) / 16)), dim3(NT), 0, s, partial,
                       (int)g, plen, plen, stats);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_lssm_smooth(vmp_ctx *ctx, int32_t given, const double *Yt, int32_t M, int64_t B,
                        int32_t T, int64_t BL, int32_t D, const double *Cm, const double *tau,
                        const double *h0, const double *Sinv, const double *J, double *Z,
                        double *stats, void *workspace)
{
    return smooth_impl(ctx, given, Yt, M, B, T, BL, D, Cm, tau, h0, Sinv, J, Z, stats, workspace, 1,
                       nullptr);
}

int32_t vmp_lssm_get_layout(int32_t D, int32_t M, vmp_lssm_layout *out)
{
    if (!out || D < 1 || M < 1) return VMP_ERR_INVALID;
    if (D > DMAX || M > LSSM_MAXM) return VMP_ERR_UNSUPPORTED;
    fill_lssm_layout(D, M, out);
    return VMP_OK;
}

// X.update() in one call: covariance recursion + per-sequence passes.  The per-sequence passes
// need only the FORWARD half of the covariance recursion (S^-1, J); its backward half (one
// wavefront, ~0.5 ms at T = 1000) runs on a side stream beside them and is joined before return
// (stream-ordered: the caller's stream continues after both).
int32_t vmp_lssm_x_update(vmp_ctx *ctx, int32_t T, int32_t D, const double *Dg0, const double *Dgm,
                          const double *DgT, const double *E, double *Sinv, double *J,
                          double *covsums, const double *Yt, int32_t M, int64_t B, int64_t BL,
                          const double *Cm, const double *tau, const double *h0, double *Z,
                          double *stats, void *workspace)
{
    VMP_REQUIRE(ctx, ctx != nullptr, VMP_ERR_INVALID, "null context");
    if (!ctx->ms[0]) {
        for (int i = 0; i < 3; ++i)
            VMP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->ms[i], hipStreamNonBlocking));
        for (int i = 0; i < VMP_NME; ++i)
            VMP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->me[i], hipEventDisableTiming));
    }
    // The covariance recursion (one wavefront, ~0.45 us per step) runs on a side stream in NSEG
    // time segments; the per-sequence forward sweep follows it segment by segment on the caller's
    // stream, so only the first segment of the recursion is exposed (0.45 -> 0.11 ms at T = 1000).
    // Its backward half (V_t, Cov(x_t, x_t+1)) runs on a second side stream beside the sweeps and
    // is joined before return (stream-ordered: the caller's stream continues after all of it).
    constexpr int NSEG = 8;
    const int nseg = (T >= 64 * NSEG && vmp_tune_get("lssm_segments", 1) != 0) ? NSEG : 1;
    VMP_HIP_CHECK(ctx, hipEventRecord(ctx->me[0], ctx->stream));
    VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->ms[1], ctx->me[0], 0));
    int32_t rc = VMP_OK;
    hipEvent_t ready[NSEG];
    for (int k = 0; k < nseg && rc == VMP_OK; ++k) {
        rc = launch_cov(ctx, ctx->ms[1], 1, T, D, Dg0, Dgm, DgT, E, Sinv, J, covsums,
                        (int)((int64_t)T * k / nseg), (int)((int64_t)T * (k + 1) / nseg));
        ready[k] = ctx->me[2 + k];
        if (rc == VMP_OK) VMP_HIP_CHECK(ctx, hipEventRecord(ready[k], ctx->ms[1]));
    }
    if (rc != VMP_OK) return rc;
    VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->ms[0], ready[nseg - 1], 0));
    rc = launch_cov(ctx, ctx->ms[0], 2, T, D, Dg0, Dgm, DgT, E, Sinv, J, covsums);
    if (rc == VMP_OK) VMP_HIP_CHECK(ctx, hipEventRecord(ctx->me[1], ctx->ms[0]));
    const int32_t rc2 = smooth_impl(ctx, 0, Yt, M, B, T, BL, D, Cm, tau, h0, Sinv, J, Z, stats,
                                    workspace, nseg, ready);
    if (rc == VMP_OK) VMP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->me[1], 0));
    return rc != VMP_OK ? rc : rc2;
}

int32_t vmp_lssm_rotate_x(vmp_ctx *ctx, int32_t D, int32_t T, int64_t B, int64_t BL, const double *R,
                          double *Z)
{
    VMP_REQUIRE(ctx, ctx && R && Z, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, D >= 1 && D <= DMAX && T >= 1 && B >= 0 && BL >= B, VMP_ERR_INVALID, "bad dims");
    if (B == 0) return VMP_OK;
    int64_t g = ((int64_t)T * B + NT - 1) / NT;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (g > cap) g = cap;
    switch (D) {
#define LSSM_ROT(d) case d: hipLaunchKernelGGL(lssm_rotate_kernel<d>, dim3((unsigned)g), dim3(NT), 0, ctx->stream, R, T, B, BL, Z); break;
        LSSM_ROT(1) LSSM_ROT(2) LSSM_ROT(3) LSSM_ROT(4) LSSM_ROT(5) LSSM_ROT(6) LSSM_ROT(7) LSSM_ROT(8)
#undef LSSM_ROT
#define LSSM_ROTB(d) case d: hipLaunchKernelGGL(lssm_rotate_big_kernel<d>, dim3((unsigned)g), dim3(NT), 0, ctx->stream, R, T, B, BL, Z); break;
        LSSM_ROTB(9) LSSM_ROTB(10) LSSM_ROTB(11) LSSM_ROTB(12) LSSM_ROTB(13) LSSM_ROTB(14)
        LSSM_ROTB(15) LSSM_ROTB(16)
#undef LSSM_ROTB
    }
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_lssm_small_ops(vmp_ctx *ctx, int32_t D, int32_t M, int32_t T, double B_total,
                           const double *priors, int32_t nu_latent, int32_t nops,
                           const int32_t *ops, double *state)
{
    VMP_REQUIRE(ctx, ctx && ops && state && priors, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, nops >= 1 && nops <= 12, VMP_ERR_INVALID, "1..12 operations per call");
    lssm_small_args A;
    int32_t rc = vmp_lssm_get_layout(D, M, &A.L);
    VMP_REQUIRE(ctx, rc == VMP_OK, rc, "unsupported dims D=%d M=%d", D, M);
    A.D = D;
    A.M = M;
    A.T = T;
    A.nops = nops;
    for (int i = 0; i < nops; ++i) {
        VMP_REQUIRE(ctx, ops[i] >= VMP_LSSM_OP_STATS && ops[i] <= VMP_LSSM_OP_ELBO, VMP_ERR_INVALID,
                    "unknown operation %d", ops[i]);
        A.ops[i] = ops[i];
    }
    A.B = B_total;
    for (int i = 0; i < 8; ++i) A.pri[i] = priors[i];
    A.nu_latent = nu_latent;
    const size_t small_lds = ((size_t)(A.L.total + 7) / 8 * 8 + (D > DREG ? (size_t)D * D * D : 0))
                             * sizeof(double);
    if (small_lds > 48 * 1024) {
        // (the state vector of D = 16, M = 64 is 105 KB: gfx950 has 160 KB of LDS per workgroup)
        static bool raised[64] = {false};
        const int dev = ctx->device & 63;
        if (!raised[dev]) {
            VMP_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(lssm_small_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   150 * 1024));
            raised[dev] = true;
        }
    }
    hipLaunchKernelGGL(lssm_small_kernel, dim3(1), dim3(64), small_lds, ctx->stream, A, state);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_lssm_workspace_doubles(int32_t D, int32_t M, int64_t B, int32_t T, int64_t *n)
{
    if (!n || D < 1 || M < 1 || B < 0) return VMP_ERR_INVALID;
    *n = ws_base_doubles(D, M, B) + ck_doubles(D, B, T);
    if (D >= 7) *n += big_extra_doubles(D, M, B, T);          // (7, 8: either path, by tune key)
    if (D <= DREG && lssm_wide(D, M)) *n += wide_extra_doubles(D, M, B, T);
    return VMP_OK;
}

}  // extern "C"
