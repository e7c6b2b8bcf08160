// TEST INFRASTRUCTURE: the per-sequence / replicated-node arithmetic of the masked state-space
// block (bayespy_amd/csrc/vmp_lssmm_dev.h -- the very text the HIP kernels run) compiled for the
// host with g++, behind the SAME C ABI as libvmp_hip.so's vmp_lssmm_* entry points (the context is
// ignored).  tests/host_build.py compiles it; the CPU suite checks it against oracle/lssm.py and
// uses it as the kernel double of plans/lssm_masked.py (no GPU needed).  The sweeps run as lane
// groups of ONE by default (the device deals a sequence over four lanes: same entries, same sums,
// the operands of other rows by lane moves; LSSMM_HOST_LANES=4 runs that form on four threads) and
// the plate sums are combined sequentially over the sequences instead of wavefront trees.
// vmp_digamma / vmp_lgamma: the host+device text of vmp_common.h, force-included by the build.
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>

#include "../../bayespy_amd/csrc/vmp_lssmm_dev.h"

// Lane groups of FOUR on the host: the four "lanes" of a sequence are four threads that run the
// sweep text in lock step; bc(v, j) -- on the device a DPP quad permute -- is an exchange through a
// shared cell between two barriers.  Slow, exact, and it exercises the row ownership, the operand
// moves and the packing of the lane accumulators exactly as the kernels have them
// (LSSMM_HOST_LANES=4; tests/test_lssm_masked_host.py).
namespace {
struct quad_exchange {
    std::atomic<int> count{0}, sense{0};
    double cell[4];
    void barrier(int &local)
    {
        local ^= 1;
        if (count.fetch_add(1) == 3) {
            count.store(0);
            sense.store(local);
        } else {
            while (sense.load() != local) std::this_thread::yield();
        }
    }
};
quad_exchange g_quad;
thread_local int t_lane = 0, t_sense = 0;
}

template <>
struct lssmm_lanes<4> {
    static double bc(double v, int j)
    {
        g_quad.cell[t_lane] = v;
        g_quad.barrier(t_sense);
        const double r = g_quad.cell[j];
        g_quad.barrier(t_sense);
        return r;
    }
};

namespace {
struct dg_fn { double operator()(double x) const { return vmp_digamma(x); } };
struct lg_fn { double operator()(double x) const { return vmp_lgamma(x); } };
constexpr int MG = 4;

int host_lanes()
{
    const char *e = getenv("LSSMM_HOST_LANES");
    return (e && atoi(e) == 4) ? 4 : 1;
}

// run body(lane) on G threads in lock step (G = 1: on the caller)
template <int G, typename F>
void run_lanes(F body)
{
    if (G == 1) {
        t_lane = 0;
        body(0);
        return;
    }
    g_quad.count.store(0);
    g_quad.sense.store(0);
    std::vector<std::thread> th;
    for (int l = 0; l < G; ++l)
        th.emplace_back([l, &body] {
            t_lane = l;
            t_sense = 0;
            body(l);
        });
    for (auto &t : th) t.join();
}
}

// The text of the sweeps as lane groups of G (1: every row on the calling thread).  The backward
// sweep carries the statistics under the rule of the device's default form.  Every lane adds its
// own slots of the raw sums (the slots of different lanes are disjoint).
template <int D, int G>
static void x_update_lanes(int given, const lssmm_seq_args &S, const double *seqobs, int64_t B,
                           double *state, const vmp_lssmm_layout &L)
{
    constexpr int MF = LSSMM_MFUSE, R = (D + G - 1) / G;
    const int M = S.M;
    const lssmm_raw ro = lssmm_raw_offsets(D, M);
    double *raw = state + L.off_raw;
    for (int e = (given == 2 ? ro.XX : 0); e < ro.len; ++e) raw[e] = 0.0;
    const bool fuse = D <= 4 && M <= MF && given != 2;
    std::atomic<int> bad{0};
    run_lanes<G>([&](int lane) {
        int lbad = 0;
        if (given != 2) {
            std::vector<double> ld(B > 0 ? B : 1, 0.0);
            if (!given)
                for (int64_t b = 0; b < B; ++b)
                    ld[b] = (M <= 8 && D <= 4 && S.nib) ? lssmm_forward_seq<D, G, 8>(S, b, lane, lbad)
                                              : lssmm_forward_seq<D, G, 64>(S, b, lane, lbad);
            for (int64_t b = 0; b < B; ++b) {
                const double w = seqobs[b];
                auto put_chain = [&](int slot, double v) { raw[slot] += w * v; };
                auto put_stats = [&](int slot, double v) { raw[ro.XX + slot] += v; };
                if (fuse) {
                    using AC = lssmm_acc<D, G, MF>;
                    double acc[AC::len];
                    if (given) lssmm_backward_seq<D, G, MF, true>(S, b, lane, acc);
                    else lssmm_backward_seq<D, G, MF, false>(S, b, lane, acc);
                    lssmm_put_chain<D, G>(lane, acc, put_chain);
                    lssmm_put_stats<D, G, MF>(lane, 0, M, acc + AC::XX, acc + AC::Syx, put_stats);
                } else {
                    double acc[lssmm_acc<D, G, 0>::len];
                    if (given) lssmm_backward_seq<D, G, 0, true>(S, b, lane, acc);
                    else lssmm_backward_seq<D, G, 0, false>(S, b, lane, acc);
                    lssmm_put_chain<D, G>(lane, acc, put_chain);
                }
                if (lane == 0) raw[ro.ld] += w * ld[b];
            }
        }
        if (!fuse)
            for (int m0 = 0; m0 < M; m0 += MG)
                for (int64_t b = 0; b < B; ++b) {
                    double acc[MG * R * (D + 1)];
                    lssmm_stats_seq<D, G, MG>(S, b, lane, m0, acc);
                    lssmm_put_stats<D, G, MG>(lane, m0, M, acc, acc + MG * R * D,
                                              [&](int slot, double v) { raw[ro.XX + slot] += v; });
                }
        if (lbad) bad.store(1);
    });
    if (bad.load()) state[L.off_scal] = (double)VMP_ERR_NOT_POSDEF;
}

template <int D>
static void x_update_impl(int given, const lssmm_seq_args &S, const double *seqobs, int64_t B,
                          double *state, const vmp_lssmm_layout &L)
{
    if (host_lanes() == 4) x_update_lanes<D, 4>(given, S, seqobs, B, state, L);
    else x_update_lanes<D, 1>(given, S, seqobs, B, state, L);
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
    if (!out || !lssmm_dims_ok(D, M)) return VMP_ERR_INVALID;
    lssmm_fill_layout(D, M, out);
    return VMP_OK;
}

int32_t vmp_lssmm_workspace_doubles(int32_t D, int32_t M, int64_t B, int32_t T, int64_t *n)
{
    (void)T;
    if (!n || !lssmm_dims_ok(D, M) || B < 0) return VMP_ERR_INVALID;
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
    if (!lssmm_dims_ok(D, M)) return VMP_ERR_INVALID;
    vmp_lssmm_layout L;
    lssmm_fill_layout(D, M, &L);
    lssmm_seq_args S;
    S.Yt = Yt; S.Mw = Mw; S.F = F; S.Z = Z; S.P = P; S.tab = state + L.off_tab;
    S.M = M; S.T = T; S.BL = BL;
    // the nibble tables of the forward sweep, under the rule of the device (LSSMM_HOST_NIBBLES=0: off)
    const char *e = getenv("LSSMM_HOST_NIBBLES");
    std::vector<double> nib((e && atoi(e) == 0) ? 0 : lssmm_nibble_len(D, M));
    S.nib = nullptr;
    if (!nib.empty()) {
        lssmm_build_nibbles(D, M, S.tab, nib.data(), 0, 1);
        S.nib = nib.data();
    }
    switch (D) {
    case 1: x_update_impl<1>(given, S, seqobs, B, state, L); break;
    case 2: x_update_impl<2>(given, S, seqobs, B, state, L); break;
    case 3: x_update_impl<3>(given, S, seqobs, B, state, L); break;
    case 4: x_update_impl<4>(given, S, seqobs, B, state, L); break;
    case 5: x_update_impl<5>(given, S, seqobs, B, state, L); break;
    case 6: x_update_impl<6>(given, S, seqobs, B, state, L); break;
    case 7: x_update_impl<7>(given, S, seqobs, B, state, L); break;
    default: x_update_impl<8>(given, S, seqobs, B, state, L); break;
    }
    return VMP_OK;
}

int32_t vmp_lssmm_rotate_p(vmp_ctx *, int32_t D, int32_t T, int64_t B, int64_t BL, const double *R,
                           double *P)
{
    if (D < 1 || D > LSSMM_DMAX) return VMP_ERR_INVALID;
    const int NS = D * (D + 1) / 2;
    for (int t = 0; t < T; ++t)
        for (int64_t b = 0; b < B; ++b) {
            double *p = P + (int64_t)t * NS * BL + b;
            switch (D) {
            case 1: lssmm_rotate_packed<1>(R, p, BL); break;
            case 2: lssmm_rotate_packed<2>(R, p, BL); break;
            case 3: lssmm_rotate_packed<3>(R, p, BL); break;
            case 4: lssmm_rotate_packed<4>(R, p, BL); break;
            case 5: lssmm_rotate_packed<5>(R, p, BL); break;
            case 6: lssmm_rotate_packed<6>(R, p, BL); break;
            case 7: lssmm_rotate_packed<7>(R, p, BL); break;
            default: lssmm_rotate_packed<8>(R, p, BL); break;
            }
        }
    return VMP_OK;
}

int32_t vmp_lssmm_small_ops(vmp_ctx *, int32_t D, int32_t M, int32_t T, const double *priors,
                            int32_t nu_latent, int32_t nops, const int32_t *ops, double *state)
{
    if (!lssmm_dims_ok(D, M) || nops < 1 || nops > 12)
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
