// vmp_lssmm_dev.h -- per-sequence and replicated-node arithmetic of the state-space block with
// ARRAY masks (vmp_lssmm.hip).  Every function here is plain scalar code on the registers of ONE
// thread (one sequence, or the single thread of the replicated-node kernel), written host+device:
// the kernels of vmp_lssmm.hip call them with b = the thread's sequence, and tests/host/
// lssmm_host.cpp compiles the SAME text with g++ and loops over b -- the CPU suite checks this
// arithmetic against oracle/lssm.py (MaskedLSSMOracle, pinned on the live reference) without a GPU.
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

constexpr int LSSMM_DMAX = 4;
constexpr int LSSMM_MMAX = 64;

// packed lower triangle: (i, j), i >= j  ->  i (i + 1) / 2 + j
VMP_HD constexpr int sym_ix(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

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
    L->len_tab = 3 * NS + DD + D + (int64_t)M * D + (int64_t)M * NS;
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

// offsets inside the table block (state + off_tab)
struct lssmm_tab {
    int base, E, h0, C, CC, len;
};
VMP_HD lssmm_tab lssmm_tab_offsets(int D, int M)
{
    const int NS = D * (D + 1) / 2;
    lssmm_tab t;
    t.base = 0;
    t.E = 3 * NS;
    t.h0 = t.E + D * D;
    t.C = t.h0 + D;
    t.CC = t.C + M * D;
    t.len = t.CC + M * NS;
    return t;
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

// in-place inverse of the SPD matrix a (packed lower triangle, D <= 4) by the symmetric
// Gauss-Jordan sweep on the full matrix; pivots go into (prod, ld); a non-positive pivot sets *bad.
template <int D>
VMP_HD void lssmm_spd_inverse(double *a, double &prod, double &ld, int &bad)
{
    double m[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) m[i][j] = a[sym_ix(i, j)];
#pragma unroll
    for (int p = 0; p < D; ++p) {
        const double piv = m[p][p];
        if (!(piv > 0.0)) bad = 1;
        lssmm_ld_acc(piv, prod, ld);
        const double d = 1.0 / piv;
#pragma unroll
        for (int j = 0; j < D; ++j) m[p][j] *= d;
        m[p][p] = d;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if (i == p) continue;
            const double c = m[i][p];
#pragma unroll
            for (int j = 0; j < D; ++j)
                if (j != p) m[i][j] -= c * m[p][j];
            m[i][p] = -c * d;
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[sym_ix(i, j)] = 0.5 * (m[i][j] + m[j][i]);
}

// arrays of one sequence: element f of time step t at base[(t * nf + f) * BL + b]
struct lssmm_seq_args {
    const double *Yt;       // (T, M, BL)
    const uint64_t *Mw;     // (T, BL)
    double *F;              // (T, NS + D, BL)
    double *Z;              // (T, D, BL)
    double *P;              // (T, NS, BL)
    const double *tab;      // tables (lssmm_tab), LDS on the device
    int M, T;
    int64_t BL;
};

// ---------------------------------------------------------------------------------------------
// forward sweep of sequence b: S_t = Dg_t - E^T S_t-1^-1 E with Dg_t = base_t + sum_m mask tau<cc>_m,
// z_t = h_t - (S_t-1^-1 E)^T z_t-1 with h_t = sum_m y tau c_m (+ Lam0 mu0 at t = 0); S_t^-1 and z_t
// go to F.  Returns log|Phi_b| = sum_t log|S_t|; *bad on a non-positive pivot.
// ---------------------------------------------------------------------------------------------
template <int D>
VMP_HD double lssmm_forward_seq(const lssmm_seq_args &A, int64_t b, int &bad)
{
    constexpr int NS = D * (D + 1) / 2;
    const lssmm_tab to = lssmm_tab_offsets(D, A.M);
    const double *tab = A.tab;
    const int M = A.M, T = A.T;
    const int64_t BL = A.BL;
    double E[D][D];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int k = 0; k < D; ++k) E[j][k] = tab[to.E + j * D + k];
    double Sinv[NS], z[D];
#pragma unroll
    for (int s = 0; s < NS; ++s) Sinv[s] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = 0.0;
    double prod = 1.0, ld = 0.0;
    // The operands of a step do not depend on the recursion: they are requested one chunk of MC
    // observed dimensions (for M <= MC: one time step) ahead of their use, so that the HBM
    // latency runs beside the arithmetic of the current step (a sequence's thread has the SIMD
    // almost to itself: 1e4 sequences are 157 wavefronts for 1024 SIMDs).
    constexpr int MC = 8;
    const int nch = (M + MC - 1) / MC;
    double yn[MC];
    uint64_t wn = A.Mw[b];
#pragma unroll
    for (int g = 0; g < MC; ++g) yn[g] = (g < M) ? A.Yt[(int64_t)g * BL + b] : 0.0;
    for (int t = 0; t < T; ++t) {
        const uint64_t w = wn;
        if (t + 1 < T) wn = A.Mw[(int64_t)(t + 1) * BL + b];
        const double *bs = tab + to.base + (t == 0 ? 0 : (t < T - 1 ? NS : 2 * NS));
        double S[NS], h[D];
#pragma unroll
        for (int s = 0; s < NS; ++s) S[s] = bs[s];
#pragma unroll
        for (int i = 0; i < D; ++i) h[i] = (t == 0) ? tab[to.h0 + i] : 0.0;
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
                const int m = c * MC + g;
                if (m < M) {
                    const double *cm = tab + to.C + m * D;
#pragma unroll
                    for (int i = 0; i < D; ++i) h[i] += y[g] * cm[i];
                    if ((w >> m) & 1) {
                        const double *cc = tab + to.CC + m * NS;
#pragma unroll
                        for (int s = 0; s < NS; ++s) S[s] += cc[s];
                    }
                }
            }
        }
        if (t > 0) {
            // J = S_t-1^-1 E;  S -= E^T J;  z = h - J^T z_prev
            double J[D][D];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) s += Sinv[sym_ix(i, j)] * E[j][k];
                    J[i][k] = s;
                }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k <= i; ++k) {
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) s += E[j][i] * J[j][k];
                    S[sym_ix(i, k)] -= s;
                }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) s += J[j][i] * z[j];
                h[i] -= s;
            }
        }
        lssmm_spd_inverse<D>(S, prod, ld, bad);
        double *fp = A.F + (int64_t)t * (NS + D) * BL + b;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            Sinv[s] = S[s];
            fp[(int64_t)s * BL] = S[s];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            z[i] = h[i];
            fp[(int64_t)(NS + i) * BL] = h[i];
        }
    }
    return ld + log(prod);
}

// ---------------------------------------------------------------------------------------------
// backward sweep of sequence b: x_t = S_t^-1 z_t - J_t x_t+1, Cov(x_t, x_t+1) = -J_t V_t+1,
// V_t = S_t^-1 - Cov(x_t, x_t+1) J_t^T  (J_t = S_t^-1 E), P_t = V_t + x_t x_t^T; x -> Z, P -> P.
// given: the <x> in Z are point masses (V = 0), no recursion.
// acc (lssmm_raw chain part, chain_len doubles, ld NOT included): sum_t P | sum <x_t+1 x_t^T> |
// P_0 | P_T-1 | x_0.
// ---------------------------------------------------------------------------------------------
template <int D>
VMP_HD void lssmm_backward_seq(const lssmm_seq_args &A, int64_t b, int given, double *acc)
{
    constexpr int NS = D * (D + 1) / 2;
    const lssmm_tab to = lssmm_tab_offsets(D, A.M);
    const lssmm_raw ro = lssmm_raw_offsets(D, A.M);
    const int T = A.T;
    const int64_t BL = A.BL;
    double E[D][D];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int k = 0; k < D; ++k) E[j][k] = A.tab[to.E + j * D + k];
    double sumP[NS], Snp[D][D], Vn[NS], xn[D], PT[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) sumP[s] = Vn[s] = PT[s] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        xn[i] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) Snp[i][j] = 0.0;
    }
    double Pc[NS], x[D];
#pragma unroll
    for (int s = 0; s < NS; ++s) Pc[s] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = 0.0;
    double fn[NS + D];
#pragma unroll
    for (int s = 0; s < NS + D; ++s)
        fn[s] = given ? 0.0 : A.F[((int64_t)(T - 1) * (NS + D) + s) * BL + b];
    for (int t = T - 1; t >= 0; --t) {
        double V[NS];
        double *zp = A.Z + (int64_t)t * D * BL + b;
        if (given) {
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = zp[(int64_t)i * BL];
#pragma unroll
            for (int s = 0; s < NS; ++s) V[s] = 0.0;
            if (t < T - 1) {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) Snp[i][j] += xn[i] * x[j];
            }
        } else {
            double Sinv[NS], z[D];
#pragma unroll
            for (int s = 0; s < NS; ++s) Sinv[s] = fn[s];
#pragma unroll
            for (int i = 0; i < D; ++i) z[i] = fn[NS + i];
            if (t > 0) {
                // the forward quantities of the step before: requested now, used next round
                const double *fp = A.F + (int64_t)(t - 1) * (NS + D) * BL + b;
#pragma unroll
                for (int s = 0; s < NS + D; ++s) fn[s] = fp[(int64_t)s * BL];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) s += Sinv[sym_ix(i, j)] * z[j];
                x[i] = s;
            }
            if (t == T - 1) {
#pragma unroll
                for (int s = 0; s < NS; ++s) V[s] = Sinv[s];
            } else {
                double J[D][D], Cn[D][D];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        double s = 0.0;
#pragma unroll
                        for (int j = 0; j < D; ++j) s += Sinv[sym_ix(i, j)] * E[j][k];
                        J[i][k] = s;
                    }
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) s += J[i][k] * xn[k];
                    x[i] -= s;
                }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        double s = 0.0;
#pragma unroll
                        for (int l = 0; l < D; ++l) s += J[i][l] * Vn[sym_ix(l, k)];
                        Cn[i][k] = -s;
                    }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        // symmetrised like the reference (utils/linalg.py:572): 1/2 (V + V^T)
                        double s1 = 0.0, s2 = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            s1 += Cn[i][k] * J[j][k];
                            s2 += Cn[j][k] * J[i][k];
                        }
                        V[sym_ix(i, j)] = Sinv[sym_ix(i, j)] - 0.5 * (s1 + s2);
                    }
                // <x_t+1 x_t^T> = Cov(x_t, x_t+1)^T + means
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) Snp[i][j] += Cn[j][i] + xn[i] * x[j];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) zp[(int64_t)i * BL] = x[i];
        }
        double *pp = A.P + (int64_t)t * NS * BL + b;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const int s = sym_ix(i, j);
                Pc[s] = V[s] + x[i] * x[j];
                pp[(int64_t)s * BL] = Pc[s];
                sumP[s] += Pc[s];
            }
        if (t == T - 1) {
#pragma unroll
            for (int s = 0; s < NS; ++s) PT[s] = Pc[s];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) Vn[s] = V[s];
#pragma unroll
        for (int i = 0; i < D; ++i) xn[i] = x[i];
    }
    // the loop ends on t = 0: Pc = P_0, x = x_0
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc[ro.sumP + s] = sumP[s];
        acc[ro.P0 + s] = Pc[s];
        acc[ro.PT + s] = PT[s];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        acc[ro.x0 + i] = x[i];
#pragma unroll
        for (int j = 0; j < D; ++j) acc[ro.Snp + i * D + j] = Snp[i][j];
    }
}

// ---------------------------------------------------------------------------------------------
// statistics of sequence b for the rows [m0, m0 + MG) of C: XX_m += mask_mbt P_bt (packed),
// Syx_m += y_mbt x_bt (y is zero where masked).  acc: MG * (NS + D) doubles, [g][NS | D].
// ---------------------------------------------------------------------------------------------
template <int D, int MG>
VMP_HD void lssmm_stats_seq(const lssmm_seq_args &A, int64_t b, int m0, double *acc)
{
    constexpr int NS = D * (D + 1) / 2;
    const int M = A.M, T = A.T;
    const int64_t BL = A.BL;
    double xx[MG][NS], yx[MG][D];
#pragma unroll
    for (int g = 0; g < MG; ++g) {
#pragma unroll
        for (int s = 0; s < NS; ++s) xx[g][s] = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) yx[g][i] = 0.0;
    }
    double pn[NS], xn[D], yn[MG];
    uint64_t wn = A.Mw[b];
#pragma unroll
    for (int s = 0; s < NS; ++s) pn[s] = A.P[(int64_t)s * BL + b];
#pragma unroll
    for (int i = 0; i < D; ++i) xn[i] = A.Z[(int64_t)i * BL + b];
#pragma unroll
    for (int g = 0; g < MG; ++g) yn[g] = (m0 + g < M) ? A.Yt[(int64_t)(m0 + g) * BL + b] : 0.0;
    for (int t = 0; t < T; ++t) {
        const uint64_t w = wn >> m0;
        double p[NS], x[D], y[MG];
#pragma unroll
        for (int s = 0; s < NS; ++s) p[s] = pn[s];
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = xn[i];
#pragma unroll
        for (int g = 0; g < MG; ++g) y[g] = yn[g];
        if (t + 1 < T) {
            wn = A.Mw[(int64_t)(t + 1) * BL + b];
            const double *pp = A.P + (int64_t)(t + 1) * NS * BL + b;
            const double *zp = A.Z + (int64_t)(t + 1) * D * BL + b;
#pragma unroll
            for (int s = 0; s < NS; ++s) pn[s] = pp[(int64_t)s * BL];
#pragma unroll
            for (int i = 0; i < D; ++i) xn[i] = zp[(int64_t)i * BL];
#pragma unroll
            for (int g = 0; g < MG; ++g)
                yn[g] = (m0 + g < M) ? A.Yt[((int64_t)(t + 1) * M + m0 + g) * BL + b] : 0.0;
        }
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            if (m0 + g < M) {
                const double bit = ((w >> g) & 1) ? 1.0 : 0.0;
#pragma unroll
                for (int s = 0; s < NS; ++s) xx[g][s] += bit * p[s];
#pragma unroll
                for (int i = 0; i < D; ++i) yx[g][i] += y[g] * x[i];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < MG; ++g) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[g * (NS + D) + s] = xx[g][s];
#pragma unroll
        for (int i = 0; i < D; ++i) acc[g * (NS + D) + NS + i] = yx[g][i];
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
                    double anua = 0.0;
                    for (int i = 0; i < D; ++i) anua += nu[2 * D + i] * AA[(i * D + j) * D + k];
                    const double dn = (j == k) ? nu[2 * D + j] : 0.0;
                    const int s = sym_ix(j, k);
                    tab[to.base + s] = Lam0[j * D + k] + (T > 1 ? anua : 0.0);
                    tab[to.base + NS + s] = dn + anua;
                    tab[to.base + 2 * NS + s] = (T > 1 ? dn : Lam0[j * D + k]);
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
                        tab[to.CC + m * NS + sym_ix(i, j)] =
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
