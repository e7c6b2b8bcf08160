// TEST INFRASTRUCTURE: the per-sequence / replicated-node arithmetic of the masked state-space
// block (bayespy_amd/csrc/vmp_lssmm_dev.h -- the very text the HIP kernels run) compiled for the
// host with g++, behind the SAME C ABI as libvmp_hip.so's vmp_lssmm_* entry points (the context is
// ignored).  tests/host_build.py compiles it; the CPU suite checks it against oracle/lssm.py and
// uses it as the kernel double of plans/lssm_masked.py (no GPU needed).  Only the plate sums are
// combined differently here (sequentially over the sequences instead of wavefront trees).
// vmp_digamma / vmp_lgamma: the host+device text of vmp_common.h, force-included by the build.
#include <string.h>
#include <vector>

#include "../../bayespy_amd/csrc/vmp_lssmm_dev.h"

namespace {
struct dg_fn { double operator()(double x) const { return vmp_digamma(x); } };
struct lg_fn { double operator()(double x) const { return vmp_lgamma(x); } };
constexpr int MG = 4;
}

template <int D>
static void x_update_impl(int given, const lssmm_seq_args &S, const double *seqobs, int64_t B,
                          double *state, const vmp_lssmm_layout &L)
{
    constexpr int NS = D * (D + 1) / 2;
    const int M = S.M;
    const lssmm_raw ro = lssmm_raw_offsets(D, M);
    double *raw = state + L.off_raw;
    for (int e = (given == 2 ? ro.XX : 0); e < ro.len; ++e) raw[e] = 0.0;
    int bad = 0;
    if (given != 2) {
        std::vector<double> ld(B > 0 ? B : 1, 0.0);
        if (!given)
            for (int64_t b = 0; b < B; ++b) ld[b] = lssmm_forward_seq<D>(S, b, bad);
        for (int64_t b = 0; b < B; ++b) {
            double acc[3 * NS + D * D + D];
            lssmm_backward_seq<D>(S, b, given, acc);
            for (int e = 0; e < 3 * NS + D * D + D; ++e) raw[e] += seqobs[b] * acc[e];
            raw[ro.ld] += seqobs[b] * ld[b];
        }
    }
    for (int m0 = 0; m0 < M; m0 += MG)
        for (int64_t b = 0; b < B; ++b) {
            double acc[MG * (NS + D)];
            lssmm_stats_seq<D, MG>(S, b, m0, acc);
            for (int g = 0; g < MG && m0 + g < M; ++g) {
                for (int s = 0; s < NS; ++s) raw[ro.XX + (m0 + g) * NS + s] += acc[g * (NS + D) + s];
                for (int i = 0; i < D; ++i) raw[ro.Syx + (m0 + g) * D + i] += acc[g * (NS + D) + NS + i];
            }
        }
    if (bad) state[L.off_scal] = (double)VMP_ERR_NOT_POSDEF;
}

extern "C" {

int32_t vmp_lssmm_limits(int32_t *max_D, int32_t *max_M)
{
    if (max_D) *max_D = LSSMM_DMAX;
    if (max_M) *max_M = LSSMM_MMAX;
    return VMP_OK;
}

int32_t vmp_lssmm_get_layout(int32_t D, int32_t M, vmp_lssmm_layout *out)
{
    if (!out || D < 1 || D > LSSMM_DMAX || M < 1 || M > LSSMM_MMAX) return VMP_ERR_INVALID;
    lssmm_fill_layout(D, M, out);
    return VMP_OK;
}

int32_t vmp_lssmm_workspace_doubles(int32_t D, int32_t M, int64_t B, int32_t T, int64_t *n)
{
    (void)T;
    if (!n || D < 1 || D > LSSMM_DMAX || M < 1 || M > LSSMM_MMAX || B < 0) return VMP_ERR_INVALID;
    *n = 64;
    return VMP_OK;
}

int32_t vmp_lssmm_prepare(vmp_ctx *, const double *Y, const uint8_t *mask, int64_t sm, int64_t sb,
                          int64_t st, int32_t M, int64_t B, int32_t T, int64_t BL, int32_t D,
                          double *Yt, uint64_t *Mw, double *seqobs, double *state, void *)
{
    vmp_lssmm_layout L;
    lssmm_fill_layout(D, M, &L);
    memset(Yt, 0, sizeof(double) * (size_t)T * M * BL);
    memset(Mw, 0, sizeof(uint64_t) * (size_t)T * BL);
    double *setup = state + L.off_setup;
    double syy = 0.0;
    for (int m = 0; m < M; ++m) {
        double c = 0.0;
        for (int64_t b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t)
                if (mask[m * sm + b * sb + (int64_t)t * st]) {
                    const double v = Y[((int64_t)m * B + b) * T + t];
                    Yt[((int64_t)t * M + m) * BL + b] = v;
                    Mw[(int64_t)t * BL + b] |= 1ull << m;
                    syy += v * v;
                    c += 1.0;
                }
        setup[2 + m] = c;
    }
    double nb = 0.0;
    for (int64_t b = 0; b < BL; ++b) {
        uint64_t any = 0;
        if (b < B)
            for (int t = 0; t < T; ++t) any |= Mw[(int64_t)t * BL + b];
        seqobs[b] = any ? 1.0 : 0.0;
        nb += seqobs[b];
    }
    setup[0] = syy;
    setup[1] = nb;
    return VMP_OK;
}

int32_t vmp_lssmm_x_update(vmp_ctx *, int32_t given, const double *Yt, const uint64_t *Mw,
                           const double *seqobs, int32_t M, int64_t B, int32_t T, int64_t BL,
                           int32_t D, double *state, double *F, double *Z, double *P, void *)
{
    if (D < 1 || D > LSSMM_DMAX || M < 1 || M > LSSMM_MMAX) return VMP_ERR_INVALID;
    vmp_lssmm_layout L;
    lssmm_fill_layout(D, M, &L);
    lssmm_seq_args S;
    S.Yt = Yt; S.Mw = Mw; S.F = F; S.Z = Z; S.P = P; S.tab = state + L.off_tab;
    S.M = M; S.T = T; S.BL = BL;
    switch (D) {
    case 1: x_update_impl<1>(given, S, seqobs, B, state, L); break;
    case 2: x_update_impl<2>(given, S, seqobs, B, state, L); break;
    case 3: x_update_impl<3>(given, S, seqobs, B, state, L); break;
    default: x_update_impl<4>(given, S, seqobs, B, state, L); break;
    }
    return VMP_OK;
}

int32_t vmp_lssmm_small_ops(vmp_ctx *, int32_t D, int32_t M, int32_t T, const double *priors,
                            int32_t nu_latent, int32_t nops, const int32_t *ops, double *state)
{
    if (D < 1 || D > LSSMM_DMAX || M < 1 || M > LSSMM_MMAX || nops < 1 || nops > 12)
        return VMP_ERR_INVALID;
    lssmm_small_args A;
    lssmm_fill_layout(D, M, &A.L);
    A.D = D; A.M = M; A.T = T; A.nops = nops;
    for (int i = 0; i < nops; ++i) A.ops[i] = ops[i];
    for (int i = 0; i < 8; ++i) A.pri[i] = priors[i];
    A.nu_latent = nu_latent;
    double scratch[lssmm_small_scratch(1)];
    lssmm_small_body(A, state, scratch, 0, 1, [] {}, dg_fn(), lg_fn());
    return VMP_OK;
}

}  // extern "C"
