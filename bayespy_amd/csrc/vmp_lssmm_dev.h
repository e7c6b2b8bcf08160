// vmp_lssmm_dev.h -- per-sequence and replicated-node arithmetic of the state-space block with
// ARRAY masks (vmp_lssmm.hip).  Host+device text: the kernels of vmp_lssmm.hip call the sweeps with
// b = the sequence and lane = the position of the calling thread inside the sequence's lane group,
// and tests/host/lssmm_host.cpp compiles the SAME text with g++ (lane groups of one) and loops over
// b -- the CPU suite checks this arithmetic against oracle/lssm.py (MaskedLSSMOracle, pinned on the
// live reference) without a GPU.
//
// A sequence is dealt over G lanes of a wavefront (G = 4 on the device, 1 on the host and as the
// device's alternative form): lane l owns the rows l R .. l R + R - 1, R = ceil(D / G), of every
// D x D matrix of the recursion (S_t, S_t^-1, J_t, V_t, <x x^T>) and the matching entries of the
// vectors.  A product with a constant right-hand side (E) is local; a product whose right-hand
// side is a matrix of the recursion takes it from the owners (lssmm_lanes<G>::bc: DPP quad
// permutes on the device, the identity for G = 1).  The arithmetic of an entry -- the terms of its
// sums and their order -- does not depend on G.  Symmetric matrices are used through the lower
// triangle of the row owners: entry (i, k), k <= i, is the one lane(i) computed.
//
// Reference code restated: linalg.block_banded_solve (utils/linalg.py:468-575) per sequence,
// GaussianMarkovChainDistribution (gaussian_markov_chain.py:270-707), SumMultiply messages with
// the child's mask (dot.py:425-633, node.py:570-655), GaussianARD with the Gamma wrapper
// (gaussian.py:649-706, :2299-2371), Gamma (gamma.py:116-148), the bound (expfamily.py:400-480).
#pragma once

#include <stdint.h>
#include <math.h>

#include "../../include/vmp_hip.h"

#ifndef VMP_HD
#ifdef __HIPCC__
#define VMP_HD __host__ __device__ inline
#else
#define VMP_HD inline
#endif
#endif

constexpr int LSSMM_DMAX = 8;
constexpr int LSSMM_MMAX = 64;           // one bit of the mask word per observed dimension
constexpr int LSSMM_MDD_MAX = 2048;      // M D^2: the tables of the sweeps in LDS (16 KB)
constexpr int LSSMM_MFUSE = 8;           // rows of C whose statistics the backward sweep carries

VMP_HD constexpr bool lssmm_dims_ok(int D, int M)
{
    return D >= 1 && D <= LSSMM_DMAX && M >= 1 && M <= LSSMM_MMAX && M * D * D <= LSSMM_MDD_MAX;
}

// packed lower triangle: (i, j), i >= j  ->  i (i + 1) / 2 + j
VMP_HD constexpr int sym_ix(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// offsets inside the table block (state + off_tab): full (unpacked) matrices, rows contiguous
//   base (3 D^2: t = 0, inner, last) | E (D^2) | h0 (D) | tau c_m (M D) | tau <c_m c_m^T> (M D^2)
struct lssmm_tab {
    int base, E, h0, C, CC, len;
};
VMP_HD constexpr lssmm_tab lssmm_tab_offsets(int D, int M)
{
    return lssmm_tab{0, 3 * D * D, 4 * D * D, 4 * D * D + D, 4 * D * D + D + M * D,
                     4 * D * D + D + M * D + M * D * D};
}

VMP_HD void lssmm_fill_layout(int D, int M, vmp_lssmm_layout *L)
{
    int64_t o = 0;
    const int DD = D * D, NS = D * (D + 1) / 2;
    L->NS = NS;
    L->off_tau = o;      o += 4;
    L->off_gamma = o;    o += 4 * D;
    L->off_alpha = o;    o += 4 * D;
    L->off_nu = o;       o += 4 * D;
    L->off_mu0 = o;      o += D;
    L->off_Lam0 = o;     o += DD;
    L->off_ldLam0 = o;   o += 1;
    L->off_Cm = o;       o += (int64_t)M * D;
    L->off_CovC = o;     o += (int64_t)M * DD;
    L->off_ldC = o;      o += M;
    L->off_SCC = o;      o += DD;
    L->off_Am = o;       o += DD;
    L->off_AA = o;       o += (int64_t)DD * D;
    L->off_ldA = o;      o += D;
    L->off_tab = o;
    L->len_tab = lssmm_tab_offsets(D, M).len;
    o += L->len_tab;
    L->off_setup = o;
    L->len_setup = 2 + M;
    o += L->len_setup;
    L->off_raw = o;
    L->len_raw = 3 * NS + DD + D + 1 + (int64_t)M * NS + (int64_t)M * D;
    o += L->len_raw;
    L->off_scal = o;     o += 8;
    L->off_L = o;        o += 16;
    L->total = (o + 7) / 8 * 8;
}

// offsets inside the raw plate sums (state + off_raw)
struct lssmm_raw {
    int sumP, Snp, P0, PT, x0, ld, XX, Syx, chain_len, len;
};
VMP_HD lssmm_raw lssmm_raw_offsets(int D, int M)
{
    const int NS = D * (D + 1) / 2;
    lssmm_raw r;
    r.sumP = 0;
    r.Snp = NS;
    r.P0 = r.Snp + D * D;
    r.PT = r.P0 + NS;
    r.x0 = r.PT + NS;
    r.ld = r.x0 + D;
    r.chain_len = r.ld + 1;
    r.XX = r.chain_len;
    r.Syx = r.XX + M * NS;
    r.len = r.Syx + M * D;
    return r;
}

// ---------------------------------------------------------------------------------------------
// lane groups: G adjacent lanes of the wavefront hold one sequence.  bc(v, j): the value of v on
// lane j of the caller's group, on every lane of the group (j a constant after unrolling).
// ---------------------------------------------------------------------------------------------
template <int G>
struct lssmm_lanes;

template <>
struct lssmm_lanes<1> {
    static VMP_HD double bc(double v, int) { return v; }
};

#if defined(__HIPCC__)
template <>
struct lssmm_lanes<4> {
    template <int J>
    static __device__ __forceinline__ double quad(double v)
    {
        // DPP quad_perm [J, J, J, J]: a vector-ALU move, no LDS crossbar
        constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ double bc(double v, int j)
    {
        switch (j) {
        case 0: return quad<0>(v);
        case 1: return quad<1>(v);
        case 2: return quad<2>(v);
        default: return quad<3>(v);
        }
    }
};
#endif

// 1 / x for a pivot (positive, far from the ends of the exponent range): on the device the
// hardware estimate and two Newton steps (5 instructions, <= 1 ulp) instead of the IEEE division
// sequence (~15) -- four to eight of them sit on the serial path of every time step
VMP_HD double lssmm_recip(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// a value that is the same on every lane (a table entry): kept in scalar registers on the device
VMP_HD double lssmm_uniform(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
#else
    return v;
#endif
}

// Sums of the observation blocks by mask NIBBLE: nib[(n * 16 + c) D^2 + i D + k] = sum over the set
// bits j of c (ascending) of tau <c_m c_m^T>[i][k], m = 4 n + j.  With them a time step adds ONE
// table row per nibble of its mask word instead of one row per observed dimension.  Used when the
// tables fit LSSMM_NIB_MAX doubles of LDS (M = 8: D <= 8; M = 32: D <= 4), else the per-row loop.
constexpr int LSSMM_NIB_MAX = 2048;
VMP_HD int lssmm_nibble_len(int D, int M)
{
    const int len = (M + 3) / 4 * 16 * D * D;
    return len <= LSSMM_NIB_MAX ? len : 0;
}
VMP_HD void lssmm_build_nibbles(int D, int M, const double *tab, double *nib, int tid, int nthr)
{
    const int DD = D * D, len = lssmm_nibble_len(D, M);
    const int oCC = 4 * DD + D + M * D;
    for (int e = tid; e < len; e += nthr) {
        const int n = e / (16 * DD), c = (e / DD) % 16, q = e % DD;
        double s = 0.0;
        for (int j = 0; j < 4; ++j) {
            const int m = 4 * n + j;
            if (m < M && ((c >> j) & 1)) s += tab[oCC + m * DD + q];
        }
        nib[e] = s;
    }
}

// running log-determinant without a logarithm per pivot (the product is folded into ld only when
// it leaves a safe range)
VMP_HD void lssmm_ld_acc(double piv, double &prod, double &ld)
{
    prod *= piv;
    if (!(prod < 1e120 && prod > 1e-120)) {
        ld += log(prod);
        prod = 1.0;
    }
}

// arrays of one sequence: element f of time step t at base[(t * nf + f) * BL + b]
struct lssmm_seq_args {
    const double *Yt;       // (T, M, BL)
    const uint64_t *Mw;     // (T, BL)
    double *F;              // (T, NS + D, BL)
    double *Z;              // (T, D, BL)
    double *P;              // (T, NS, BL)
    const double *tab;      // tables (lssmm_tab), LDS on the device
    const double *nib;      // nibble tables (lssmm_build_nibbles) or null
    int M, T;
    int64_t BL;
};

// the rows of this lane: global index, index clamped into the matrix (a lane beyond the last row
// is an exact clone of the owner of row D - 1: same operands, same arithmetic, and what it stores
// is what the owner stores), offset of the row in the packed triangle
template <int D, int G>
struct lssmm_rows {
    static constexpr int R = (D + G - 1) / G;
    int row[R], rowc[R], tri[R];
    VMP_HD explicit lssmm_rows(int lane)
    {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            row[r] = lane * R + r;
            rowc[r] = row[r] < D ? row[r] : D - 1;
            tri[r] = rowc[r] * (rowc[r] + 1) / 2;
        }
    }
    VMP_HD bool valid(int r) const { return row[r] < D; }
    // offset of entry (row r, k) in the packed lower triangle (either side of the diagonal)
    VMP_HD int sym(int r, int k) const { return k <= rowc[r] ? tri[r] + k : k * (k + 1) / 2 + rowc[r]; }
};

// the lower triangle of this lane's rows into a packed array (entry s of the triangle at
// base[s * stride]).  Every lane issues the same D stores per row, none under a condition (a store
// inside a branch makes the compiler wait for ALL outstanding memory operations where the
// operands requested ahead are first used): an entry above the diagonal stores the diagonal
// entry once more.
template <int D, int G, int R>
VMP_HD void lssmm_store_rows(double *base, int64_t stride, const lssmm_rows<D, G> &rw,
                             const double (&m)[R][D])
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double diag = m[r][0];
#pragma unroll
        for (int k = 1; k < D; ++k) diag = (rw.rowc[r] == k) ? m[r][k] : diag;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const bool low = k <= rw.rowc[r];
            base[(int64_t)(rw.tri[r] + (low ? k : rw.rowc[r])) * stride] = low ? m[r][k] : diag;
        }
    }
}

// in-place inverse of the SPD matrix whose rows are dealt over the lanes (S[r][.] = row lane R + r)
// by the Gauss-Jordan sweep: the pivot row travels from its owner, every lane updates its rows;
// pivots go into (prod, ld) on every lane alike; a non-positive pivot sets bad.
template <int D, int G>
VMP_HD void lssmm_rows_inverse(double (*S)[D], const lssmm_rows<D, G> &rw, double &prod, double &ld,
                               int &bad)
{
    using LN = lssmm_lanes<G>;
    constexpr int R = (D + G - 1) / G;
#pragma unroll
    for (int p = 0; p < D; ++p) {
        double prs[D];
#pragma unroll
        for (int k = 0; k < D; ++k) prs[k] = LN::bc(S[p % R][k], p / R);
        const double piv = prs[p];
        if (!(piv > 0.0)) bad = 1;
        lssmm_ld_acc(piv, prod, ld);
        const double d = lssmm_recip(piv);
#pragma unroll
        for (int k = 0; k < D; ++k) prs[k] *= d;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool isp = rw.rowc[r] == p;
            const double c = S[r][p];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (k == p) continue;
                const double u = fma(-c, prs[k], S[r][k]);
                S[r][k] = isp ? prs[k] : u;
            }
            S[r][p] = isp ? d : -c * d;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the recursion part of a forward step on the rows of this lane: Sn (the diagonal block Dg_t) and
// h (h_t) in, S (S_t-1^-1) and zf (z_t-1 in full) in / out:
//   Sn -= E^T S E, h -= E^T S zf, S = Sn^-1, zf = h from its owners
// (at t = 0 the previous inverse and z are zero and the step runs the same text)
// ---------------------------------------------------------------------------------------------
template <int D, int G, int R, int ED>
VMP_HD void lssmm_forward_recur(double (&Sn)[R][D], double (&h)[R], double (&S)[R][D], double (&zf)[D],
                                const double (&ET)[R][D], const double (&E)[ED][ED], const double *tab,
                                const lssmm_rows<D, G> &rw, double &prod, double &ld, int &bad)
{
    using LN = lssmm_lanes<G>;
    constexpr bool ER = ED == D;
    constexpr int oE = 3 * D * D;
    // Tm = E^T S_t-1^-1 (rows of this lane; S_t-1^-1 from the lower triangle of its owners)
    double Tm[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int l = 0; l < D; ++l) Tm[r][l] = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int l = 0; l <= j; ++l) {
            const double s = LN::bc(S[j % R][l], j / R);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Tm[r][l] = fma(ET[r][j], s, Tm[r][l]);
                if (l != j) Tm[r][j] = fma(ET[r][l], s, Tm[r][j]);
            }
        }
    // S_t -= Tm E;  h -= Tm z_t-1
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double a = Sn[r][k];
#pragma unroll
            for (int l = 0; l < D; ++l)
                a = fma(-Tm[r][l], ER ? E[ER ? l : 0][ER ? k : 0] : tab[oE + l * D + k], a);
            Sn[r][k] = a;
        }
        double a = h[r];
#pragma unroll
        for (int l = 0; l < D; ++l) a = fma(-Tm[r][l], zf[l], a);
        h[r] = a;
    }
    lssmm_rows_inverse<D, G>(Sn, rw, prod, ld, bad);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) S[r][k] = Sn[r][k];
#pragma unroll
    for (int l = 0; l < D; ++l) zf[l] = LN::bc(h[l % R], l / R);
}

// ---------------------------------------------------------------------------------------------
// forward sweep of sequence b: S_t = Dg_t - E^T S_t-1^-1 E with Dg_t = base_t + sum_m mask tau<cc>_m,
// z_t = h_t - E^T S_t-1^-1 z_t-1 with h_t = sum_m y tau c_m (+ Lam0 mu0 at t = 0); S_t^-1 (packed
// lower triangle) and z_t go to F.  Returns log|Phi_b| = sum_t log|S_t| (on every lane of the
// group); bad on a non-positive pivot.  Every lane stores: the sequences b in [B, BL) that fill the
// last wavefront are the zero-filled padding columns of the arrays (no data, no mask bit, weight 0).
// MB = 8 (M <= 8, nibble tables present): the observation terms of step t + 1 -- table rows picked
// by the mask nibbles, eight products with y -- are formed DURING step t, beside the serial chain
// of its recursion, from operands requested two steps ahead; the step is one straight block of
// code.  MB = 64: any M, the rows of C in chunks of eight.
// ---------------------------------------------------------------------------------------------
template <int D, int G, int MB>
VMP_HD double lssmm_forward_seq(const lssmm_seq_args &A, int64_t b, int lane, int &bad)
{
    constexpr int R = (D + G - 1) / G, NS = D * (D + 1) / 2, DD = D * D;
    constexpr int oE = 3 * DD, oh0 = 4 * DD, oC = 4 * DD + D;
    constexpr bool ER = D <= 4;               // E in (scalar) registers; beyond: read from the table
    constexpr int ED = ER ? D : 1;
    const lssmm_rows<D, G> rw(lane);
    const double *tab = A.tab;
    const double *nib = A.nib;
    const int M = A.M, T = A.T;
    const int oCC = oC + M * D;
    const int64_t BL = A.BL;
    double E[ED][ED];
    if (ER) {
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int k = 0; k < D; ++k) E[ER ? j : 0][ER ? k : 0] = lssmm_uniform(tab[oE + j * D + k]);
    } else {
        E[0][0] = 0.0;
    }
    // column i of E for the rows i of this lane (E^T . from the left)
    double ET[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < D; ++j) ET[r][j] = tab[oE + j * D + rw.rowc[r]];
    double S[R][D], zf[D];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) S[r][k] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) zf[i] = 0.0;
    double prod = 1.0, ld = 0.0;
    if (MB == 8) {
        // tau c_m for the rows of this lane (zero beyond the last row of C)
        double Cr[8][R];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) Cr[m][r] = (m < M) ? tab[oC + m * D + rw.rowc[r]] : 0.0;
        // observation terms of one step from its mask word and data
        auto obs = [&](int tq, uint64_t w, const double (&y)[8], double (&So)[R][D], double (&ho)[R]) {
            const double *bs = tab + (tq == 0 ? 0 : (tq < T - 1 ? DD : 2 * DD));
            const int c0 = (int)(w & 15), c1 = (int)((w >> 4) & 15);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double *q0 = nib + (c0 * D + rw.rowc[r]) * D;
                // M <= 4: one table; the (empty) second nibble picks its all-zero first row
                const double *q1 = nib + (((M > 4 ? 16 : 0) + c1) * D + rw.rowc[r]) * D;
#pragma unroll
                for (int k = 0; k < D; ++k) So[r][k] = (bs[rw.rowc[r] * D + k] + q0[k]) + q1[k];
                double a = (tq == 0) ? tab[oh0 + rw.rowc[r]] : 0.0;
#pragma unroll
                for (int m = 0; m < 8; ++m) a = fma(y[m], Cr[m][r], a);
                ho[r] = a;
            }
        };
        auto fetch = [&](int tq, uint64_t &w, double (&y)[8]) {
            w = A.Mw[(int64_t)tq * BL + b];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                // beyond the last row: the last row once more, as zero (no load under a condition)
                const double v = A.Yt[((int64_t)tq * M + (m < M ? m : M - 1)) * BL + b];
                y[m] = (m < M) ? v : 0.0;
            }
        };
        double y1[8], So[R][D], ho[R];
        uint64_t w1;
        fetch(0, w1, y1);
        obs(0, w1, y1, So, ho);
        fetch(T > 1 ? 1 : 0, w1, y1);
        // The results of a step are stored at the START of the next one: a wait for operands
        // requested ahead also waits for every store still in flight (loads and stores complete
        // out of order with each other), so the stores go out a whole step before the next wait.
        double Sn[R][D], h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            h[r] = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) Sn[r][k] = 0.0;
        }
        for (int t = 0; t < T; ++t) {
            // step t + 1: its operands arrived during step t - 1; request those of step t + 2
            // (beyond the end: the last step once more, unused)
            const int t1 = t + 1 < T ? t + 1 : T - 1, t2 = t + 2 < T ? t + 2 : T - 1;
            double Sc[R][D], hc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                hc[r] = ho[r];
#pragma unroll
                for (int k = 0; k < D; ++k) Sc[r][k] = So[r][k];
            }
            obs(t1, w1, y1, So, ho);
            {
                // step t - 1 out (at t = 0: zeros into the slots of step 0, rewritten next time)
                const int ts = t > 0 ? t - 1 : 0;
                lssmm_store_rows<D, G>(A.F + (int64_t)ts * (NS + D) * BL + b, BL, rw, Sn);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    A.F[((int64_t)ts * (NS + D) + NS + rw.rowc[r]) * BL + b] = h[r];
            }
            fetch(t2, w1, y1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                h[r] = hc[r];
#pragma unroll
                for (int k = 0; k < D; ++k) Sn[r][k] = Sc[r][k];
            }
            lssmm_forward_recur<D, G, R, ED>(Sn, h, S, zf, ET, E, tab, rw, prod, ld, bad);
        }
        lssmm_store_rows<D, G>(A.F + (int64_t)(T - 1) * (NS + D) * BL + b, BL, rw, Sn);
#pragma unroll
        for (int r = 0; r < R; ++r) A.F[((int64_t)(T - 1) * (NS + D) + NS + rw.rowc[r]) * BL + b] = h[r];
        return ld + log(prod);
    }
    const int nn = (M + 3) / 4;
    // The operands of a step do not depend on the recursion: they are requested one chunk of MC
    // observed dimensions ahead of their use, so that the HBM latency runs beside the arithmetic of
    // the current step.
    constexpr int MC = 8;
    const int nch = (M + MC - 1) / MC;
    double yn[MC];
    uint64_t wn = A.Mw[b];
#pragma unroll
    for (int g = 0; g < MC; ++g) yn[g] = (g < M) ? A.Yt[(int64_t)g * BL + b] : 0.0;
    for (int t = 0; t < T; ++t) {
        const uint64_t w = wn;
        if (t + 1 < T) wn = A.Mw[(int64_t)(t + 1) * BL + b];
        const double *bs = tab + (t == 0 ? 0 : (t < T - 1 ? DD : 2 * DD));
        double Sn[R][D], h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int k = 0; k < D; ++k) Sn[r][k] = bs[rw.rowc[r] * D + k];
            h[r] = (t == 0) ? tab[oh0 + rw.rowc[r]] : 0.0;
        }
        if (nib) {
            for (int n = 0; n < nn; ++n) {
                const int c = (int)((w >> (4 * n)) & 15);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double *q = nib + ((n * 16 + c) * D + rw.rowc[r]) * D;
#pragma unroll
                    for (int k = 0; k < D; ++k) Sn[r][k] += q[k];
                }
            }
        }
        for (int c = 0; c < nch; ++c) {
            double y[MC];
#pragma unroll
            for (int g = 0; g < MC; ++g) y[g] = yn[g];
            int tn = t, cn = c + 1;
            if (cn == nch) {
                tn = t + 1;
                cn = 0;
            }
            if (tn < T) {
                const double *yp = A.Yt + ((int64_t)tn * M + cn * MC) * BL + b;
#pragma unroll
                for (int g = 0; g < MC; ++g) yn[g] = (cn * MC + g < M) ? yp[(int64_t)g * BL] : 0.0;
            }
#pragma unroll
            for (int g = 0; g < MC; ++g) {
                // beyond the last row: y = 0 and a zero mask bit against the (finite) entries of row M - 1
                const int m = c * MC + g < M ? c * MC + g : M - 1;
#pragma unroll
                for (int r = 0; r < R; ++r) h[r] = fma(y[g], tab[oC + m * D + rw.rowc[r]], h[r]);
                if (!nib) {
                    const double bd = (c * MC + g < M) ? (double)((w >> m) & 1) : 0.0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double *cc = tab + oCC + (m * D + rw.rowc[r]) * D;
#pragma unroll
                        for (int k = 0; k < D; ++k) Sn[r][k] = fma(bd, cc[k], Sn[r][k]);
                    }
                }
            }
        }
        lssmm_forward_recur<D, G, R, ED>(Sn, h, S, zf, ET, E, tab, rw, prod, ld, bad);
        lssmm_store_rows<D, G>(A.F + (int64_t)t * (NS + D) * BL + b, BL, rw, Sn);
#pragma unroll
        for (int r = 0; r < R; ++r) A.F[((int64_t)t * (NS + D) + NS + rw.rowc[r]) * BL + b] = h[r];
    }
    return ld + log(prod);
}

// accumulators of a lane (rows of this lane): the chain sums, then -- when the backward sweep
// carries the statistics of MF rows of C -- XX_m rows and Syx_m entries
//   sumP[R][D] | SnpT[R][D] (column i of sum <x_t+1 x_t^T>) | P0[R][D] | PT[R][D] | x0[R] |
//   XX[MF][R][D] | Syx[MF][R]
template <int D, int G, int MF>
struct lssmm_acc {
    static constexpr int R = (D + G - 1) / G;
    static constexpr int sumP = 0, Snp = R * D, P0 = 2 * R * D, PT = 3 * R * D, x0 = 4 * R * D;
    static constexpr int chain = 4 * R * D + R;
    static constexpr int XX = chain, Syx = chain + MF * R * D;
    static constexpr int len = chain + MF * R * (D + 1);
};

// ---------------------------------------------------------------------------------------------
// backward sweep of sequence b: x_t = S_t^-1 z_t - J_t x_t+1, W = J_t V_t+1 (= -Cov(x_t, x_t+1)),
// V_t = S_t^-1 + W J_t^T  (J_t = S_t^-1 E), P_t = V_t + x_t x_t^T; x -> Z, P -> P (packed).
// given: the <x> in Z are point masses (V = 0), no recursion.  MF > 0: the statistics of the rows
// m < M <= MF of C (XX_m += mask_mbt P_bt, Syx_m += y_mbt x_bt) ride along -- P and <x> never come
// back from HBM for them.  acc: lssmm_acc<D, G, MF> (this lane's rows; not weighted).
// ---------------------------------------------------------------------------------------------
template <int D, int G, int MF, bool given>
VMP_HD void lssmm_backward_seq(const lssmm_seq_args &A, int64_t b, int lane, double *acc)
{
    using LN = lssmm_lanes<G>;
    using AC = lssmm_acc<D, G, MF>;
    constexpr int R = (D + G - 1) / G, NS = D * (D + 1) / 2, DD = D * D;
    constexpr int oE = 3 * DD;
    constexpr int MFR = MF > 0 ? MF : 1;
    constexpr bool ER = D <= 4;               // E in (scalar) registers; beyond: read from the table
    const lssmm_rows<D, G> rw(lane);
    const double *tab = A.tab;
    const int M = A.M, T = A.T;
    const int64_t BL = A.BL;
    double E[ER ? D : 1][ER ? D : 1];
    if (ER) {
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int k = 0; k < D; ++k) E[ER ? j : 0][ER ? k : 0] = lssmm_uniform(tab[oE + j * D + k]);
    }
    double sumP[R][D], snpT[R][D], XX[MFR][R][D], Syx[MFR][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) sumP[r][k] = snpT[r][k] = 0.0;
#pragma unroll
    for (int m = 0; m < MFR; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            Syx[m][r] = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) XX[m][r][k] = 0.0;
        }
    // what a step hands to the one before it: <x_t+1> in full, Cov(x_t+1) as the lower triangle of
    // its owners.  Zero behind the last step: the step T - 1 runs the same text (J x = 0, W = 0).
    double Vnp[NS], xn[D];
#pragma unroll
    for (int s = 0; s < NS; ++s) Vnp[s] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) xn[i] = 0.0;
    // operands of a step: this lane's rows of S_t^-1, z_t (given: <x_t>) in full, and for the
    // statistics y_t and the mask word; with one row per lane (D <= 4) requested one step ahead
    constexpr bool PF = R == 1;
    double fn[R][D], zn[D], yn[MFR];
    uint64_t wn = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) fn[r][k] = 0.0;
#pragma unroll
    for (int m = 0; m < MFR; ++m) yn[m] = 0.0;
    auto load_step = [&](int tq, double (&f)[R][D], double (&zz)[D], double (&yy)[MFR], uint64_t &ww) {
        if (given) {
            const double *zp = A.Z + (int64_t)tq * D * BL + b;
#pragma unroll
            for (int i = 0; i < D; ++i) zz[i] = zp[(int64_t)i * BL];
        } else {
            const double *fp = A.F + (int64_t)tq * (NS + D) * BL + b;
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < D; ++k) f[r][k] = fp[(int64_t)rw.sym(r, k) * BL];
#pragma unroll
            for (int i = 0; i < D; ++i) zz[i] = fp[(int64_t)(NS + i) * BL];
        }
        if (MF > 0) {
            ww = A.Mw[(int64_t)tq * BL + b];
#pragma unroll
            for (int m = 0; m < MFR; ++m) {
                // beyond the last row: the last row once more, as zero (no load under a condition)
                const double v = A.Yt[((int64_t)tq * M + (m < M ? m : M - 1)) * BL + b];
                yy[m] = (m < M) ? v : 0.0;
            }
        }
    };
    if (PF) load_step(T - 1, fn, zn, yn, wn);
    double Pr[R][D], x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        x[r] = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) Pr[r][k] = 0.0;
    }
    // statistics of one step: XX_m += mask_m P, Syx_m += y_m x (rows m >= M: y = 0 and no mask
    // bit -- their accumulators stay zero)
    double Pq[R][D], xq[R], yq[MFR];
    uint64_t wq = 0;
#pragma unroll
    for (int m = 0; m < MFR; ++m) yq[m] = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        xq[r] = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) Pq[r][k] = 0.0;
    }
    auto stats_of = [&](uint64_t w, const double (&y)[MFR], const double (&xs)[R],
                        const double (&Ps)[R][D]) {
#pragma unroll
        for (int m = 0; m < MFR; ++m) {
            const double bd = (double)((w >> m) & 1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < D; ++k) XX[m][r][k] = fma(bd, Ps[r][k], XX[m][r][k]);
                Syx[m][r] = fma(y[m], xs[r], Syx[m][r]);
            }
        }
    };
    for (int t = T - 1; t >= 0; --t) {
        double Si[R][D], z[D], y[MFR];
        if (!PF) load_step(t, fn, zn, yn, wn);       // two rows per lane: no room for operands ahead
        const uint64_t w = wn;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < D; ++k) Si[r][k] = fn[r][k];
#pragma unroll
        for (int i = 0; i < D; ++i) z[i] = zn[i];
#pragma unroll
        for (int m = 0; m < MFR; ++m) y[m] = yn[m];
        // the step before (at t = 0: this step once more, unused)
        if (PF) load_step(t > 0 ? t - 1 : 0, fn, zn, yn, wn);
        {
            // step t + 1 out, a whole step before the next wait for operands (a wait for a load also
            // waits for every store in flight); in the first iteration: zeros into the slots of step
            // T - 1, rewritten by the next iteration
            const int ts = t < T - 1 ? t + 1 : T - 1;
            lssmm_store_rows<D, G>(A.P + (int64_t)ts * NS * BL + b, BL, rw, Pr);
            if (!given) {
#pragma unroll
                for (int r = 0; r < R; ++r) A.Z[((int64_t)ts * D + rw.rowc[r]) * BL + b] = x[r];
            }
        }
        if (MF > 0) stats_of(wq, yq, xq, Pq);      // step t + 1 (nothing in the first iteration)
        // V: rows of Cov(x_t); W: rows of J_t V_t+1 = -Cov(x_t, x_t+1); x: <x_t> of this lane's rows
        double V[R][D], W[R][D], xf[D];
        if (given) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double s = z[0];
#pragma unroll
                for (int i = 1; i < D; ++i) s = (rw.rowc[r] == i) ? z[i] : s;
                x[r] = s;
#pragma unroll
                for (int k = 0; k < D; ++k) V[r][k] = W[r][k] = 0.0;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) xf[i] = z[i];
        } else {
            double J[R][D];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) s = fma(Si[r][j], z[j], s);
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double u = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j)
                        u = fma(Si[r][j], ER ? E[ER ? j : 0][ER ? k : 0] : tab[oE + j * D + k], u);
                    J[r][k] = u;
                }
#pragma unroll
                for (int k = 0; k < D; ++k) s = fma(-J[r][k], xn[k], s);
                x[r] = s;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double u = 0.0;
#pragma unroll
                    for (int l = 0; l < D; ++l) u = fma(J[r][l], Vnp[sym_ix(l, k)], u);
                    W[r][k] = u;
                    V[r][k] = Si[r][k];
                }
            }
            // V_t = S_t^-1 + W J_t^T: the rows of J_t from their owners
#pragma unroll
            for (int j = 0; j < D; ++j)
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double jj = LN::bc(J[j % R][k], j / R);
#pragma unroll
                    for (int r = 0; r < R; ++r) V[r][j] = fma(W[r][k], jj, V[r][j]);
                }
#pragma unroll
            for (int l = 0; l < D; ++l) xf[l] = LN::bc(x[l % R], l / R);
        }
        // column i of sum <x_t+1 x_t^T> = Cov(x_t, x_t+1)^T + the means (zero at t = T - 1: xn = W = 0)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int a = 0; a < D; ++a) snpT[r][a] += fma(xn[a], x[r], -W[r][a]);
        // P_t = V_t + x_t x_t^T (rows of this lane), out; the sums
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < D; ++k) {
                Pr[r][k] = fma(x[r], xf[k], V[r][k]);
                sumP[r][k] += Pr[r][k];
            }
        if (MF > 0) {
            // hand the step to the statistics of the NEXT iteration (there they run beside the
            // serial chain of the recursion instead of behind it)
            wq = w;
#pragma unroll
            for (int m = 0; m < MFR; ++m) yq[m] = y[m];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                xq[r] = x[r];
#pragma unroll
                for (int k = 0; k < D; ++k) Pq[r][k] = Pr[r][k];
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) xn[i] = xf[i];
        if (!given) {
#pragma unroll
            for (int l = 0; l < D; ++l)
#pragma unroll
                for (int k = 0; k <= l; ++k) Vnp[sym_ix(l, k)] = LN::bc(V[l % R][k], l / R);
        }
    }
    lssmm_store_rows<D, G>(A.P + b, BL, rw, Pr);                              // step 0 out
    if (!given) {
#pragma unroll
        for (int r = 0; r < R; ++r) A.Z[(int64_t)rw.rowc[r] * BL + b] = x[r];
    }
    if (MF > 0) stats_of(wq, yq, xq, Pq);          // step 0
    // the loop ends on t = 0: Pr = P_0, x = x_0; P_T-1 (lower triangle) back from the array this
    // thread wrote (a lane that stores nothing carries weight zero)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[AC::x0 + r] = x[r];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            acc[AC::sumP + r * D + k] = sumP[r][k];
            acc[AC::Snp + r * D + k] = snpT[r][k];
            acc[AC::P0 + r * D + k] = Pr[r][k];
            acc[AC::PT + r * D + k] = A.P[((int64_t)(T - 1) * NS + rw.sym(r, k)) * BL + b];
        }
    }
    if (MF > 0) {
#pragma unroll
        for (int m = 0; m < MFR; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[AC::Syx + m * R + r] = Syx[m][r];
#pragma unroll
                for (int k = 0; k < D; ++k) acc[AC::XX + (m * R + r) * D + k] = XX[m][r][k];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// statistics of sequence b for the rows [m0, m0 + MG) of C from the stored <x>, <x x^T>:
// XX_m += mask_mbt P_bt, Syx_m += y_mbt x_bt (y is zero where masked).  For the rows the backward
// sweep does not carry, and after a re-observe (given = 2).
// acc (this lane's rows): XX[MG][R][D] | Syx[MG][R]
// ---------------------------------------------------------------------------------------------
template <int D, int G, int MG>
VMP_HD void lssmm_stats_seq(const lssmm_seq_args &A, int64_t b, int lane, int m0, double *acc)
{
    constexpr int R = (D + G - 1) / G, NS = D * (D + 1) / 2;
    const lssmm_rows<D, G> rw(lane);
    const int M = A.M, T = A.T;
    const int64_t BL = A.BL;
    double xx[MG][R][D], yx[MG][R];
#pragma unroll
    for (int g = 0; g < MG; ++g)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            yx[g][r] = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) xx[g][r][k] = 0.0;
        }
    double pn[R][D], xn[R], yn[MG];
    uint64_t wn = A.Mw[b];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < D; ++k) pn[r][k] = A.P[(int64_t)rw.sym(r, k) * BL + b];
        xn[r] = A.Z[(int64_t)rw.rowc[r] * BL + b];
    }
#pragma unroll
    for (int g = 0; g < MG; ++g) yn[g] = (m0 + g < M) ? A.Yt[(int64_t)(m0 + g) * BL + b] : 0.0;
    for (int t = 0; t < T; ++t) {
        const uint64_t w = wn >> m0;
        double p[R][D], x[R], y[MG];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            x[r] = xn[r];
#pragma unroll
            for (int k = 0; k < D; ++k) p[r][k] = pn[r][k];
        }
#pragma unroll
        for (int g = 0; g < MG; ++g) y[g] = yn[g];
        if (t + 1 < T) {
            wn = A.Mw[(int64_t)(t + 1) * BL + b];
            const double *pp = A.P + (int64_t)(t + 1) * NS * BL + b;
            const double *zp = A.Z + (int64_t)(t + 1) * D * BL + b;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < D; ++k) pn[r][k] = pp[(int64_t)rw.sym(r, k) * BL];
                xn[r] = zp[(int64_t)rw.rowc[r] * BL];
            }
#pragma unroll
            for (int g = 0; g < MG; ++g)
                yn[g] = (m0 + g < M) ? A.Yt[((int64_t)(t + 1) * M + m0 + g) * BL + b] : 0.0;
        }
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            if (m0 + g < M) {
                const double bd = (double)((w >> g) & 1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int k = 0; k < D; ++k) xx[g][r][k] = fma(bd, p[r][k], xx[g][r][k]);
                    yx[g][r] = fma(y[g], x[r], yx[g][r]);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < MG; ++g)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[MG * R * D + g * R + r] = yx[g][r];
#pragma unroll
            for (int k = 0; k < D; ++k) acc[(g * R + r) * D + k] = xx[g][r][k];
        }
}

// ---------------------------------------------------------------------------------------------
// a lane's row accumulators -> the packed layout of the raw sums (lssmm_raw); ``put(slot, value)``
// stores or adds.  chain: acc = lssmm_acc chain part -> slots of [sumP | Snp | P0 | PT | x0];
// stats: acc = XX[MG][R][D] | Syx[MG][R] -> slots relative to XX_0 (XX (M, NS) then Syx (M, D)).
// ---------------------------------------------------------------------------------------------
template <int D, int G, typename PUT>
VMP_HD void lssmm_put_chain(int lane, const double *acc, PUT put)
{
    using AC = lssmm_acc<D, G, 0>;
    constexpr int R = (D + G - 1) / G, NS = D * (D + 1) / 2;
    const lssmm_rows<D, G> rw(lane);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!rw.valid(r)) continue;
        const int i = rw.row[r];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k <= i) {
                put(rw.tri[r] + k, acc[AC::sumP + r * D + k]);
                put(NS + D * D + rw.tri[r] + k, acc[AC::P0 + r * D + k]);
                put(2 * NS + D * D + rw.tri[r] + k, acc[AC::PT + r * D + k]);
            }
            put(NS + k * D + i, acc[AC::Snp + r * D + k]);            // Snp[k][i]: column i
        }
        put(3 * NS + D * D + i, acc[AC::x0 + r]);
    }
}

template <int D, int G, int MG, typename PUT>
VMP_HD void lssmm_put_stats(int lane, int m0, int M, const double *xx, const double *yx, PUT put)
{
    constexpr int R = (D + G - 1) / G, NS = D * (D + 1) / 2;
    const lssmm_rows<D, G> rw(lane);
#pragma unroll
    for (int g = 0; g < MG; ++g) {
        const int m = m0 + g;
        if (m >= M) continue;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!rw.valid(r)) continue;
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (k <= rw.row[r]) put(m * NS + rw.tri[r] + k, xx[(g * R + r) * D + k]);
            put(M * NS + m * D + rw.row[r], yx[g * R + r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// <x x^T> <- R <x x^T> R^T of one (step, sequence) on the packed triangle (the rotations of
// inference/transformations.py applied to the plate array; reference: gaussian.py:1693-1741)
// ---------------------------------------------------------------------------------------------
template <int D>
VMP_HD void lssmm_rotate_packed(const double *R, double *p, int64_t stride)
{
    constexpr int NS = D * (D + 1) / 2;
    double a[NS], t[D][D];
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = p[(int64_t)s * stride];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int l = 0; l < D; ++l) {
            double u = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) u = fma(R[i * D + k], a[sym_ix(k, l)], u);
            t[i][l] = u;
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double u = 0.0;
#pragma unroll
            for (int l = 0; l < D; ++l) u = fma(t[i][l], R[j * D + l], u);
            p[(int64_t)sym_ix(i, j) * stride] = u;
        }
}

// ---------------------------------------------------------------------------------------------
// replicated nodes and the bound: a few 10^4 flops of scalar code on the state vector (one thread)
// ---------------------------------------------------------------------------------------------
struct lssmm_small_args {
    vmp_lssmm_layout L;
    int D, M, T, nops;
    int ops[12];
    double pri[8];            // Gamma priors (a0, b0) of tau, gamma, alpha, nu
    int nu_latent;
};

// in-place inverse of the SPD matrix A (D x D, row-major) by Gauss-Jordan; returns log|A|
VMP_HD double lssmm_serial_inverse(double *A, int D, int *bad)
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
            for (int j = 0; j < D; ++j)
                if (j != p) A[i * D + j] -= c * A[p * D + j];
            A[i * D + p] = -c * d;
        }
    }
    return ld;
}

template <typename DG>
VMP_HD void lssmm_set_gamma(double *g, int n, int k, double a, double b, DG digamma_fn)
{
    g[0 * n + k] = a;
    g[1 * n + k] = b;
    g[2 * n + k] = a / b;
    g[3 * n + k] = digamma_fn(a) - log(b);
}

template <typename LG>
VMP_HD double lssmm_gamma_term(double a0, double b0, const double *g, int n, int k, LG lgamma_fn)
{
    const double a = g[0 * n + k], b = g[1 * n + k];
    return (a0 * log(b0) - lgamma_fn(a0)) - (a * log(b) - lgamma_fn(a)) + (b - b0) * g[2 * n + k]
           + (a0 - a) * g[3 * n + k];
}

// scratch of the replicated-node routine (doubles): one D x D work matrix per thread, then shared
// cells: innov[DMAX] | rowsum[MMAX] | gterm[3 DMAX + 1] | resid, nobs, Mobs, bad
VMP_HD constexpr int lssmm_small_scratch(int nthr)
{
    return nthr * LSSMM_DMAX * LSSMM_DMAX + LSSMM_DMAX + LSSMM_MMAX + 3 * LSSMM_DMAX + 1 + 4;
}

// DG / LG: digamma and log-gamma (vmp_common.h: vmp_digamma, vmp_lgamma), passed in so that this
// header stays free of the HIP runtime.
// Cooperative form: ``nthr`` threads (one wavefront on the device, ONE on the host) run it
// together; a section is row-parallel (rows of C / A dealt to the threads), entry-parallel
// (matrix entries dealt to the threads) or serial (thread 0), ``sync()`` stands between dependent
// sections.  Every sum over rows goes through a per-row cell and is added by thread 0 in row order,
// so the result does not depend on nthr (the host build with one thread == the device with 64).
// The state sits in LDS on the device: a single thread pays one LDS round trip per dependent access
// (126 us per launch of ~5000 of them); dealt over the wavefront the launch is a few round trips deep.
template <typename DG, typename LG, typename SYNC>
VMP_HD void lssmm_small_body(const lssmm_small_args &A, double *st, double *scratch, int tid, int nthr,
                             SYNC sync, DG digamma_fn, LG lgamma_fn)
{
    const vmp_lssmm_layout &L = A.L;
    const int D = A.D, M = A.M, T = A.T, DD = D * D, NS = D * (D + 1) / 2;
    const lssmm_tab to = lssmm_tab_offsets(D, M);
    const lssmm_raw ro = lssmm_raw_offsets(D, M);
    double *tau = st + L.off_tau, *gam = st + L.off_gamma, *alp = st + L.off_alpha, *nu = st + L.off_nu;
    double *Cm = st + L.off_Cm, *CovC = st + L.off_CovC, *ldC = st + L.off_ldC, *SCC = st + L.off_SCC;
    double *Am = st + L.off_Am, *AA = st + L.off_AA, *ldA = st + L.off_ldA;
    double *tab = st + L.off_tab;
    const double *setup = st + L.off_setup;      // [0] sum mask y^2 [1] sequences with data [2..] n_m
    const double *raw = st + L.off_raw;
    double *sc = st + L.off_scal;
    const double *nm = setup + 2;
    const double Beff = setup[1];
    double *tmp = scratch + tid * (LSSMM_DMAX * LSSMM_DMAX);       // this thread's work matrix
    double *innov = scratch + nthr * (LSSMM_DMAX * LSSMM_DMAX);
    double *rowsum = innov + LSSMM_DMAX;
    double *gterm = rowsum + LSSMM_MMAX;
    double *cell = gterm + 3 * LSSMM_DMAX + 1;                     // [0] resid [1] nobs [2] Mobs [3] bad
    if (tid == 0) {
        double nobs = 0.0, Mobs = 0.0;
        for (int m = 0; m < M; ++m) {
            nobs += nm[m];
            Mobs += nm[m] > 0.0 ? 1.0 : 0.0;
        }
        cell[1] = nobs;
        cell[2] = Mobs;
        cell[3] = 0.0;
    }
    sync();
    int bad = 0;
    // chain statistics from the raw sums (sequences with data only)
    const double *sumP = raw + ro.sumP, *Snp = raw + ro.Snp, *P0 = raw + ro.P0, *PT = raw + ro.PT;
    const double *s0 = raw + ro.x0, *XX = raw + ro.XX, *Syx = raw + ro.Syx;
    for (int oi = 0; oi < A.nops; ++oi) {
        const int op = A.ops[oi];
        const double nobs = cell[1], Mobs = cell[2];
        if (op == VMP_LSSM_OP_C) {
            // Lam_m = diag<gamma> + <tau> XX_m;  c_m = Cov_m <tau> Syx_m;  a row without data gets
            // its prior-only posterior (expfamily.py:343-366 updates ignored plates too)
            for (int m = tid; m < M; m += nthr) {
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j)
                        tmp[i * D + j] = tau[2] * XX[m * NS + sym_ix(i, j)] + (i == j ? gam[2 * D + i] : 0.0);
                const double ld = lssmm_serial_inverse(tmp, D, &bad);
                ldC[m] = -ld;
                for (int e = 0; e < DD; ++e) CovC[m * DD + e] = tmp[e];
                for (int i = 0; i < D; ++i) {
                    double s = 0.0;
                    for (int k = 0; k < D; ++k) s += tmp[i * D + k] * tau[2] * Syx[m * D + k];
                    Cm[m * D + i] = s;
                }
            }
            sync();
        }
        if (op == VMP_LSSM_OP_C || op == VMP_LSSM_OP_GAMMA || op == VMP_LSSM_OP_ELBO) {
            // sum over the observed rows of <c_m c_m^T> (the message to gamma: node.py:570-655)
            for (int e = tid; e < DD; e += nthr) {
                const int i = e / D, j = e % D;
                double s = 0.0;
                for (int m = 0; m < M; ++m)
                    if (nm[m] > 0.0) s += CovC[m * DD + e] + Cm[m * D + i] * Cm[m * D + j];
                SCC[e] = s;
            }
            sync();
        }
        if (op == VMP_LSSM_OP_GAMMA) {
            for (int j = tid; j < D; j += nthr)
                lssmm_set_gamma(gam, D, j, A.pri[2] + 0.5 * Mobs, A.pri[3] + 0.5 * SCC[j * D + j],
                                digamma_fn);
            sync();
        } else if (op == VMP_LSSM_OP_XPREP) {
            // shared parts of the chain precision (gaussian_markov_chain.py:270-441), h_0, and the
            // per-row observation terms tau c_m, tau <c_m c_m^T>
            const double *Lam0 = st + L.off_Lam0, *mu0 = st + L.off_mu0;
            for (int e = tid; e < DD; e += nthr) {
                const int j = e / D, k = e % D;
                if (k <= j) {
                    // symmetric blocks, stored in full: entry (j, k) and its mirror
                    double anua = 0.0;
                    for (int i = 0; i < D; ++i) anua += nu[2 * D + i] * AA[(i * D + j) * D + k];
                    const double dn = (j == k) ? nu[2 * D + j] : 0.0;
                    const int s = j * D + k, sT = k * D + j;
                    tab[to.base + s] = tab[to.base + sT] = Lam0[j * D + k] + (T > 1 ? anua : 0.0);
                    tab[to.base + DD + s] = tab[to.base + DD + sT] = dn + anua;
                    tab[to.base + 2 * DD + s] = tab[to.base + 2 * DD + sT] = (T > 1 ? dn : Lam0[j * D + k]);
                }
                tab[to.E + j * D + k] = -nu[2 * D + k] * Am[k * D + j];       // Phi[t, t+1][j][k]
            }
            for (int i = tid; i < D; i += nthr) {
                double s = 0.0;
                for (int k = 0; k < D; ++k) s += Lam0[i * D + k] * mu0[k];
                tab[to.h0 + i] = s;
            }
            for (int m = tid; m < M; m += nthr) {
                for (int i = 0; i < D; ++i) {
                    tab[to.C + m * D + i] = tau[2] * Cm[m * D + i];
                    for (int j = 0; j <= i; ++j)
                        tab[to.CC + (m * D + i) * D + j] = tab[to.CC + (m * D + j) * D + i] =
                            tau[2] * (CovC[m * DD + i * D + j] + Cm[m * D + i] * Cm[m * D + j]);
                }
            }
            if (tid == 0) sc[1] = tau[2];
            sync();
        } else if (op == VMP_LSSM_OP_A) {
            // Spp = sum_{t<T-1} P = sumP - P_T-1;  Snp[i] = sum <x_t+1,i x_t>
            for (int i = tid; i < D; i += nthr) {
                for (int j = 0; j < D; ++j)
                    for (int k = 0; k < D; ++k)
                        tmp[j * D + k] = nu[2 * D + i] * (sumP[sym_ix(j, k)] - PT[sym_ix(j, k)])
                                         + (j == k ? alp[2 * D + j] : 0.0);
                const double ld = lssmm_serial_inverse(tmp, D, &bad);
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
            sync();
        } else if (op == VMP_LSSM_OP_ALPHA) {
            for (int j = tid; j < D; j += nthr) {
                double s = 0.0;
                for (int i = 0; i < D; ++i) s += AA[(i * D + j) * D + j];
                lssmm_set_gamma(alp, D, j, A.pri[4] + 0.5 * D, A.pri[5] + 0.5 * s, digamma_fn);
            }
            sync();
        } else if (op == VMP_LSSM_OP_TAU || op == VMP_LSSM_OP_NU || op == VMP_LSSM_OP_ELBO) {
            // residual sum mask <(y - c.x)^2> = sum mask y^2 + sum_m (<c c^T>_m : XX_m - 2 c_m . Syx_m)
            // and the innovation sums from the statistics
            for (int m = tid; m < M; m += nthr) {
                double syf = 0.0, sff = 0.0;
                for (int k = 0; k < D; ++k) syf += Cm[m * D + k] * Syx[m * D + k];
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j)
                        sff += (CovC[m * DD + i * D + j] + Cm[m * D + i] * Cm[m * D + j])
                               * XX[m * NS + sym_ix(i, j)];
                rowsum[m] = sff - 2.0 * syf;
            }
            for (int i = tid; i < D; i += nthr) {
                double s = sumP[sym_ix(i, i)] - P0[sym_ix(i, i)];              // Snn[i][i]
                for (int j = 0; j < D; ++j) s -= 2.0 * Am[i * D + j] * Snp[i * D + j];
                for (int j = 0; j < D; ++j)
                    for (int k = 0; k < D; ++k)
                        s += AA[(i * D + j) * D + k] * (sumP[sym_ix(j, k)] - PT[sym_ix(j, k)]);
                innov[i] = s;
            }
            sync();
            if (tid == 0) {
                double r = setup[0];
                for (int m = 0; m < M; ++m) r += rowsum[m];
                cell[0] = r;
            }
            sync();
            const double resid = cell[0];
            if (op == VMP_LSSM_OP_TAU) {
                if (tid == 0) {
                    lssmm_set_gamma(tau, 1, 0, A.pri[0] + 0.5 * nobs, A.pri[1] + 0.5 * resid, digamma_fn);
                    if (!(tau[1] > 0.0)) sc[0] = (double)VMP_ERR_FLOATING;
                }
                sync();
            } else if (op == VMP_LSSM_OP_NU) {
                for (int i = tid; i < D; i += nthr)
                    lssmm_set_gamma(nu, D, i, A.pri[6] + 0.5 * Beff * (T - 1), A.pri[7] + 0.5 * innov[i],
                                    digamma_fn);
                sync();
            } else {
                // the Gamma terms (two log-gammas each) dealt to the threads: gamma | alpha | nu | tau
                for (int e = tid; e < 3 * D + 1; e += nthr) {
                    double g = 0.0;
                    if (e < D) g = lssmm_gamma_term(A.pri[2], A.pri[3], gam, D, e, lgamma_fn);
                    else if (e < 2 * D) g = lssmm_gamma_term(A.pri[4], A.pri[5], alp, D, e - D, lgamma_fn);
                    else if (e < 3 * D) {
                        if (A.nu_latent) g = lssmm_gamma_term(A.pri[6], A.pri[7], nu, D, e - 2 * D, lgamma_fn);
                    } else g = lssmm_gamma_term(A.pri[0], A.pri[1], tau, 1, 0, lgamma_fn);
                    gterm[e] = g;
                }
                sync();
                if (tid == 0) {
                    const double LOG2PI = 1.8378770664093453;
                    double *Lo = st + L.off_L;
                    const double *Lam0 = st + L.off_Lam0, *mu0 = st + L.off_mu0;
                    Lo[0] = nobs * (-0.5 * LOG2PI + 0.5 * tau[3]) - 0.5 * tau[2] * resid;           // Y
                    double lc = 0.0;
                    for (int m = 0; m < M; ++m)
                        if (nm[m] > 0.0) lc += 0.5 * ldC[m] + 0.5 * D;
                    for (int j = 0; j < D; ++j)
                        lc += 0.5 * Mobs * gam[3 * D + j] - 0.5 * gam[2 * D + j] * SCC[j * D + j];
                    Lo[1] = lc;                                                                     // C
                    double la = 0.5 * D * D;
                    for (int i = 0; i < D; ++i) la += 0.5 * ldA[i];
                    for (int j = 0; j < D; ++j) {
                        double s = 0.0;
                        for (int i = 0; i < D; ++i) s += AA[(i * D + j) * D + j];
                        la += 0.5 * D * alp[3 * D + j] - 0.5 * alp[2 * D + j] * s;
                    }
                    Lo[2] = la;                                                                     // A
                    double slognu = 0.0;
                    for (int i = 0; i < D; ++i) slognu += nu[3 * D + i];
                    double lx = Beff * (0.5 * T * D + 0.5 * st[L.off_ldLam0] + 0.5 * (T - 1) * slognu)
                                - 0.5 * raw[ro.ld];
                    for (int i = 0; i < D; ++i)
                        for (int j = 0; j < D; ++j)
                            lx -= 0.5 * Lam0[i * D + j]
                                  * (P0[sym_ix(i, j)] - s0[i] * mu0[j] - mu0[i] * s0[j]
                                     + Beff * mu0[i] * mu0[j]);
                    for (int i = 0; i < D; ++i) lx -= 0.5 * nu[2 * D + i] * innov[i];
                    Lo[3] = lx;                                                                     // X
                    double lg = 0.0, lal = 0.0, lnu = 0.0;
                    for (int j = 0; j < D; ++j) {
                        lg += gterm[j];
                        lal += gterm[D + j];
                        lnu += gterm[2 * D + j];
                    }
                    Lo[4] = lg;
                    Lo[5] = lal;
                    Lo[6] = gterm[3 * D];
                    Lo[7] = lnu;
                    Lo[8] = Lo[0] + Lo[1] + Lo[2] + Lo[3] + Lo[4] + Lo[5] + Lo[6] + Lo[7];
                }
                sync();
            }
        }
    }
    if (bad) cell[3] = 1.0;
    sync();
    if (tid == 0 && cell[3] != 0.0) sc[0] = (double)VMP_ERR_NOT_POSDEF;
}
