// vmp_generic.hip -- generic plate-broadcast kernels behind bayespy_amd.utils:
//
//   vmp_sum_multiply    misc.sum_multiply / sum_multiply_to_plates (utils/misc.py:805-933,
//                       np.einsum(optimize=False) call site :906) -- E14/E15
//   vmp_ewise           fused broadcast elementwise formulas (the ufunc chains inside the
//                       five VMP formulas of every Distribution), one pass over HBM
//   vmp_spd_batched     linalg.chol / chol_inv / chol_logdet (utils/linalg.py:31-223),
//                       one workgroup (or wavefront) per matrix instead of a Python loop -- E13
//   vmp_softmax_moments misc.normalized_exp / logsumexp (utils/misc.py:1366-1401) -- E20
//   vmp_onehot_i64      CategoricalMoments.compute_fixed_moments (categorical.py:30-46),
//                       integer scatter, bit-exact -- E26
//
// All arrays are fp64 with explicit element strides (stride 0 = broadcast axis), i.e.
// the reference's broadcast-compressed plate semantics are preserved on the device.
#include "vmp_common.h"

#include <stdint.h>

namespace {

constexpr int NT = 256;
constexpr int MAXD = VMP_MAX_DIMS;
constexpr int MAXIN = VMP_MAX_OPERANDS;

struct Iter {
    int nk, nr, nin;                 // kept dims, reduced dims, operands
    int64_t ksize[MAXD], rsize[MAXD];
    int64_t kstride[MAXIN][MAXD], rstride[MAXIN][MAXD];
    int64_t okstride[MAXD];
    const double *in[MAXIN];
    int64_t nkeep, nred;
    int i32;                         // every flat index fits 32 bits: cheaper divisions
};

// (IT: Iter as a kernel argument, or the record of a queued operation in constant memory)
template <class IT>
__device__ inline void decode_keep(const IT &it, int64_t o, int64_t *off, int64_t &ooff)
{
    for (int i = 0; i < it.nin; ++i) off[i] = 0;
    ooff = 0;
    if (it.i32) {
        uint32_t t = (uint32_t)o;
        for (int d = it.nk - 1; d >= 0; --d) {
            const uint32_t sz = (uint32_t)it.ksize[d];
            const uint32_t q = t / sz, c = t - q * sz;
            t = q;
            for (int i = 0; i < it.nin; ++i) off[i] += (int64_t)c * it.kstride[i][d];
            ooff += (int64_t)c * it.okstride[d];
        }
        return;
    }
    for (int d = it.nk - 1; d >= 0; --d) {
        const int64_t q = o / it.ksize[d];
        const int64_t c = o - q * it.ksize[d];
        o = q;
        for (int i = 0; i < it.nin; ++i) off[i] += c * it.kstride[i][d];
        ooff += c * it.okstride[d];
    }
}

template <class IT>
__device__ inline double product_at(const IT &it, const int64_t *base, int64_t r)
{
    int64_t off[MAXIN];
    for (int i = 0; i < it.nin; ++i) off[i] = base[i];
    if (it.nr == 1) {
        for (int i = 0; i < it.nin; ++i) off[i] += r * it.rstride[i][0];
    } else if (it.i32) {
        uint32_t t = (uint32_t)r;
        for (int d = it.nr - 1; d >= 0; --d) {
            const uint32_t sz = (uint32_t)it.rsize[d];
            const uint32_t q = t / sz, c = t - q * sz;
            t = q;
            for (int i = 0; i < it.nin; ++i) off[i] += (int64_t)c * it.rstride[i][d];
        }
    } else if (it.nr == 2) {
        const int64_t q = r / it.rsize[1];
        const int64_t c = r - q * it.rsize[1];
        for (int i = 0; i < it.nin; ++i) off[i] += q * it.rstride[i][0] + c * it.rstride[i][1];
    } else
    for (int d = it.nr - 1; d >= 0; --d) {
        const int64_t q = r / it.rsize[d];
        const int64_t c = r - q * it.rsize[d];
        r = q;
        for (int i = 0; i < it.nin; ++i) off[i] += c * it.rstride[i][d];
    }
    double p = it.in[0][off[0]];
    for (int i = 1; i < it.nin; ++i) p *= it.in[i][off[i]];
    return p;
}

// One thread per output element, sequential reduction (many outputs / short sums).
__global__ void __launch_bounds__(NT)
sum_multiply_thread_kernel(Iter it, double scale, double *__restrict__ out)
{
    for (int64_t o = (int64_t)blockIdx.x * NT + threadIdx.x; o < it.nkeep;
         o += (int64_t)gridDim.x * NT) {
        int64_t base[MAXIN], ooff;
        decode_keep(it, o, base, ooff);
        double acc = 0.0;
        for (int64_t r = 0; r < it.nred; ++r) acc += product_at(it, base, r);
        out[ooff] = scale * acc;
    }
}

// Many outputs, each a short reduction along a dense axis (row dot products, traces, ...):
// a group of G lanes owns one output and walks the reduced index together, so consecutive
// lanes read consecutive addresses; the group's partial sums meet by xor-shuffles.
template <int G>
__global__ void __launch_bounds__(NT)
sum_multiply_rowgroup_kernel(Iter it, double scale, double *__restrict__ out)
{
    const int gl = threadIdx.x % G;
    const int64_t ngroups = (int64_t)gridDim.x * (NT / G);
    const int64_t first = (int64_t)blockIdx.x * (NT / G) + threadIdx.x / G;
    // the same trip count for every lane: all lanes of a wavefront take part in the shuffles
    const int64_t trips = (it.nkeep + ngroups - 1) / ngroups;
    for (int64_t t = 0; t < trips; ++t) {
        const int64_t o = first + t * ngroups;
        const bool act = o < it.nkeep;
        int64_t base[MAXIN], ooff;
        decode_keep(it, act ? o : 0, base, ooff);
        double acc = 0.0;
        if (act)
            for (int64_t r = gl; r < it.nred; r += G) acc += product_at(it, base, r);
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (act && gl == 0) out[ooff] = scale * acc;
    }
}

// One workgroup per (output element, slice of the reduction domain): wavefront
// reductions, fixed-order combination of the slices => deterministic.
__global__ void __launch_bounds__(NT)
sum_multiply_block_kernel(Iter it, int nsplit, double *__restrict__ partial, double scale,
                          double *__restrict__ out)
{
    __shared__ double red[NT / 64];
    const int64_t o = blockIdx.x;
    const int sp = blockIdx.y;
    int64_t base[MAXIN], ooff;
    decode_keep(it, o, base, ooff);
    const int64_t chunk = (it.nred + nsplit - 1) / nsplit;
    const int64_t r0 = sp * chunk;
    const int64_t r1 = (r0 + chunk < it.nred) ? r0 + chunk : it.nred;
    double acc = 0.0;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += NT) acc += product_at(it, base, r);
    acc = block_sum<NT>(acc, red);
    if (threadIdx.x == 0) {
        if (nsplit == 1) out[ooff] = scale * acc;          // single slice: no combine pass
        else partial[(int64_t)sp * it.nkeep + o] = acc;
    }
}

// Up to 16 kept elements, long reduction, kept axes not one dense axis: every thread walks reduce
// positions and accumulates ALL kept elements of its position in registers, so each 128-byte
// line of the big operand is consumed by one lane and the index arithmetic is paid once per
// position (2.1 TB/s on (N,16)x(N,1)->(16,); the lane-group form below reaches 1.7 there but
// wins beyond 16 kept elements, where this form runs out of registers).
template <int NK>
__global__ void __launch_bounds__(NT)
sum_multiply_fatthread_kernel(Iter it, int nsplit, double *__restrict__ partial)
{
    __shared__ int64_t koff[MAXIN][NK];
    __shared__ double wsum[NT / 64][NK];
    const int tid = threadIdx.x;
    const int nkeep = (int)it.nkeep;
    if (tid < NK) {
        int64_t off[MAXIN], ooff;
        decode_keep(it, tid < nkeep ? tid : 0, off, ooff);
        for (int i = 0; i < it.nin; ++i) koff[i][tid] = off[i];
    }
    __syncthreads();
    double acc[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) acc[k] = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * NT + tid; r < it.nred; r += (int64_t)gridDim.x * NT) {
        int64_t roff[MAXIN];
        for (int i = 0; i < it.nin; ++i) roff[i] = 0;
        if (it.i32) {
            uint32_t q = (uint32_t)r;
            for (int d = it.nr - 1; d >= 0; --d) {
                const uint32_t sz = (uint32_t)it.rsize[d];
                const uint32_t q2 = q / sz, c = q - q2 * sz;
                q = q2;
                for (int i = 0; i < it.nin; ++i) roff[i] += (int64_t)c * it.rstride[i][d];
            }
        } else {
            int64_t q = r;
            for (int d = it.nr - 1; d >= 0; --d) {
                const int64_t q2 = q / it.rsize[d];
                const int64_t c = q - q2 * it.rsize[d];
                q = q2;
                for (int i = 0; i < it.nin; ++i) roff[i] += c * it.rstride[i][d];
            }
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (k < nkeep) {
                double p = it.in[0][roff[0] + koff[0][k]];
                for (int i = 1; i < it.nin; ++i) p *= it.in[i][roff[i] + koff[i][k]];
                acc[k] += p;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const double v = wave_sum(acc[k]);
        if ((tid & 63) == 0) wsum[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < nkeep) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) v += wsum[w][tid];
        partial[(int64_t)blockIdx.x * it.nkeep + tid] = v;
    }
}

// Few outputs (<= 64 kept elements), long reduction, kept axes not one dense axis (e.g. the sum
// over sequences and time of (B, T, D, D) blocks against a broadcast factor): a group of NK lanes
// owns a reduce position, lane k its kept element k, so the kept elements of one position --
// contiguous in the big operand -- are read by neighbouring lanes (coalesced), every lane keeps
// ONE accumulator, and a thread runs U positions per round to keep loads in flight.  (The first
// form of this kernel gave every thread all kept elements of its position: 64 cache lines per
// load instruction, 0.85 TB/s.)
template <int NK>
__global__ void __launch_bounds__(NT)
sum_multiply_fat_kernel(Iter it, int nsplit, double *__restrict__ partial)
{
    constexpr int RPB = NT / NK;          // reduce positions per workgroup and step
    constexpr int U = 8;
    __shared__ double red[NT];
    const int tid = threadIdx.x;
    const int k = tid % NK, rs = tid / NK;
    const int nkeep = (int)it.nkeep;
    const bool kact = k < nkeep;
    int64_t koff[MAXIN], ooff;
    decode_keep(it, kact ? k : 0, koff, ooff);
    double acc = 0.0;
    const int64_t step = (int64_t)gridDim.x * RPB;
    for (int64_t r0 = (int64_t)blockIdx.x * RPB + rs; r0 < it.nred; r0 += step * U) {
        double p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * step;
            const bool ok = kact && r < it.nred;
            int64_t roff[MAXIN];
            for (int i = 0; i < it.nin; ++i) roff[i] = 0;
            if (it.i32) {
                uint32_t q = ok ? (uint32_t)r : 0u;
                for (int d = it.nr - 1; d >= 0; --d) {
                    const uint32_t sz = (uint32_t)it.rsize[d];
                    const uint32_t q2 = q / sz, c = q - q2 * sz;
                    q = q2;
                    for (int i = 0; i < it.nin; ++i) roff[i] += (int64_t)c * it.rstride[i][d];
                }
            } else {
                int64_t q = ok ? r : 0;
                for (int d = it.nr - 1; d >= 0; --d) {
                    const int64_t q2 = q / it.rsize[d];
                    const int64_t c = q - q2 * it.rsize[d];
                    q = q2;
                    for (int i = 0; i < it.nin; ++i) roff[i] += c * it.rstride[i][d];
                }
            }
            double v = ok ? it.in[0][roff[0] + koff[0]] : 0.0;
            for (int i = 1; i < it.nin; ++i) v *= ok ? it.in[i][roff[i] + koff[i]] : 0.0;
            p[u] = v;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += p[u];
    }
    // fixed-order combination of the RPB position-lanes of every kept element
    red[tid] = acc;
    __syncthreads();
    for (int half = RPB / 2; half >= 1; half >>= 1) {
        if (rs < half) red[tid] += red[tid + half * NK];
        __syncthreads();
    }
    if (rs == 0 && kact) partial[(int64_t)blockIdx.x * it.nkeep + k] = red[k];
}

// Column reduce: the innermost kept axis is dense in the operands that carry it, so
// lanes run along it (coalesced) and the workgroup's row-lanes split the reduction.
//   grid = (kept-inner blocks, kept-outer, nsplit); block = KX x (NT/KX) threads.
template <int KX>
__global__ void __launch_bounds__(NT)
sum_multiply_column_kernel(Iter it, int nsplit, double *__restrict__ partial)
{
    constexpr int RY = NT / KX;
    __shared__ double tile[RY][KX + 1];
    const int kx = threadIdx.x % KX, ry = threadIdx.x / KX;
    const int64_t kin = it.ksize[it.nk - 1];
    const int64_t ki = (int64_t)blockIdx.x * KX + kx;        // index along the inner kept axis
    const int64_t ko = blockIdx.y;                           // flattened outer kept index
    const int sp = blockIdx.z;
    const int64_t o = ko * kin + (ki < kin ? ki : 0);
    int64_t base[MAXIN], ooff;
    decode_keep(it, o, base, ooff);
    const int64_t chunk = (it.nred + nsplit - 1) / nsplit;
    const int64_t r0 = sp * chunk;
    const int64_t r1 = (r0 + chunk < it.nred) ? r0 + chunk : it.nred;
    double acc = 0.0;
    if (ki < kin)
        for (int64_t r = r0 + ry; r < r1; r += RY) acc += product_at(it, base, r);
    tile[ry][kx] = acc;
    __syncthreads();
    if (ry == 0 && ki < kin) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < RY; ++j) s += tile[j][kx];
        partial[(int64_t)sp * it.nkeep + o] = s;
    }
}

// ---------------------------------------------------------------------------
// Dense two-axis forms.  After merging, many plate sums and row contractions of the engine are
// a product of operands over a [outer][inner] index space in which every operand is either
// dense ([outer][inner] contiguous), a vector along one of the two axes, or a scalar, and
// `inner` is a power of two: the messages to plate-less parents (sum over N of (N, K) or
// (N, K, K) arrays times per-plate weights), traces and row dot products.  Lanes then run over
// the flat index in PAIRS (16-byte loads, four in flight per lane), a lane's inner position
// never changes, and no index is ever divided.
//   class 0: dense   in[o * inner + i]      class 1: per-outer vector  in[o]
//   class 2: per-inner vector  in[i]        class 3: scalar            in[0]
// ---------------------------------------------------------------------------
struct Dense2D {
    const double *in[MAXIN];
    int cls[MAXIN];
    int nin, inner;
    int64_t outer;
};

constexpr int D2_U = 4;

// reduce over OUTER (column sums): partial[block][inner]
__global__ void __launch_bounds__(NT)
sum_multiply_colsum_kernel(Dense2D a, double *__restrict__ partial)
{
    __shared__ double red[NT][2];
    const int tid = threadIdx.x;
    const int half = a.inner >> 1;                 // pairs per row, a power of two <= NT
    const int kp = tid & (half - 1);
    const int rb = NT / half;                      // rows a workgroup covers per load round
    const int rt = tid / half;
    // factors that do not depend on the row
    double c0 = 1.0, c1 = 1.0;
    const double *pf[MAXIN];
#pragma unroll
    for (int i = 0; i < MAXIN; ++i) {
        pf[i] = nullptr;
        if (i < a.nin) {
            if (a.cls[i] == 2) {
                const v2f64 q = *reinterpret_cast<const v2f64 *>(a.in[i] + 2 * kp);
                c0 *= q[0];
                c1 *= q[1];
            } else if (a.cls[i] == 3) {
                c0 *= a.in[i][0];
                c1 *= a.in[i][0];
            } else {
                pf[i] = a.in[i] + (a.cls[i] == 0 ? 2 * kp : 0);
            }
        }
    }
    double acc0 = 0.0, acc1 = 0.0;
    const int64_t rstep = (int64_t)gridDim.x * rb * D2_U;
    for (int64_t r0 = (int64_t)blockIdx.x * rb * D2_U + rt; r0 < a.outer; r0 += rstep) {
        double p0[D2_U], p1[D2_U];
#pragma unroll
        for (int u = 0; u < D2_U; ++u) {
            const int64_t r = r0 + (int64_t)u * rb;
            const bool ok = r < a.outer;
            const int64_t rr = ok ? r : 0;
            double x0 = ok ? 1.0 : 0.0, x1 = x0;
#pragma unroll
            for (int i = 0; i < MAXIN; ++i) {
                if (i < a.nin && pf[i]) {
                    if (a.cls[i] == 0) {
                        const v2f64 q = __builtin_nontemporal_load(
                            reinterpret_cast<const v2f64 *>(pf[i] + rr * a.inner));
                        x0 *= q[0];
                        x1 *= q[1];
                    } else {
                        const double w = pf[i][rr];
                        x0 *= w;
                        x1 *= w;
                    }
                }
            }
            p0[u] = x0;
            p1[u] = x1;
        }
#pragma unroll
        for (int u = 0; u < D2_U; ++u) {
            acc0 += p0[u];
            acc1 += p1[u];
        }
    }
    red[tid][0] = acc0 * c0;
    red[tid][1] = acc1 * c1;
    __syncthreads();
    for (int k = tid; k < a.inner; k += NT) {
        double v = 0.0;
        for (int j = 0; j < rb; ++j) v += red[j * half + (k >> 1)][k & 1];   // fixed order
        partial[(int64_t)blockIdx.x * a.inner + k] = v;
    }
}

// reduce over INNER (row sums): out[o * ostride] = scale * sum_i prod
__global__ void __launch_bounds__(NT)
sum_multiply_rowsum_kernel(Dense2D a, double scale, int64_t ostride, double *__restrict__ out)
{
    const int tid = threadIdx.x;
    const int half = a.inner >> 1;                 // lanes per row, a power of two <= 64
    const int ip = tid & (half - 1);
    const int rb = NT / half;
    const int rt = tid / half;
    double c0 = 1.0, c1 = 1.0, cs = 1.0;
    const double *pf[MAXIN];
#pragma unroll
    for (int i = 0; i < MAXIN; ++i) {
        pf[i] = nullptr;
        if (i < a.nin) {
            if (a.cls[i] == 2) {
                const v2f64 q = *reinterpret_cast<const v2f64 *>(a.in[i] + 2 * ip);
                c0 *= q[0];
                c1 *= q[1];
            } else if (a.cls[i] == 3) {
                cs *= a.in[i][0];
            } else {
                pf[i] = a.in[i] + (a.cls[i] == 0 ? 2 * ip : 0);
            }
        }
    }
    cs *= scale;
    const int64_t rstep = (int64_t)gridDim.x * rb * D2_U;
    // every lane of a wavefront runs the same number of rounds (the shuffles need all lanes)
    for (int64_t rbase = (int64_t)blockIdx.x * rb * D2_U; rbase < a.outer; rbase += rstep) {
        double acc[D2_U];
#pragma unroll
        for (int u = 0; u < D2_U; ++u) {
            const int64_t r = rbase + rt + (int64_t)u * rb;
            const bool ok = r < a.outer;
            const int64_t rr = ok ? r : 0;
            double x0 = c0, x1 = c1;
#pragma unroll
            for (int i = 0; i < MAXIN; ++i) {
                if (i < a.nin && pf[i]) {
                    if (a.cls[i] == 0) {
                        const v2f64 q = __builtin_nontemporal_load(
                            reinterpret_cast<const v2f64 *>(pf[i] + rr * a.inner));
                        x0 *= q[0];
                        x1 *= q[1];
                    } else {
                        const double w = pf[i][rr];
                        x0 *= w;
                        x1 *= w;
                    }
                }
            }
            acc[u] = x0 + x1;
        }
#pragma unroll
        for (int u = 0; u < D2_U; ++u) {
            double v = acc[u];
            for (int off = half >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            const int64_t r = rbase + rt + (int64_t)u * rb;
            if (ip == 0 && r < a.outer) out[r * ostride] = cs * v;
        }
    }
}

__global__ void __launch_bounds__(NT)
sum_multiply_finish_kernel(Iter it, int nsplit, double scale, const double *__restrict__ partial,
                           double *__restrict__ out)
{
    for (int64_t o = (int64_t)blockIdx.x * NT + threadIdx.x; o < it.nkeep;
         o += (int64_t)gridDim.x * NT) {
        int64_t base[MAXIN], ooff;
        decode_keep(it, o, base, ooff);
        double acc = 0.0;
        // slice-major partials: consecutive outputs are consecutive addresses
        for (int s = 0; s < nsplit; ++s) acc += partial[(int64_t)s * it.nkeep + o];
        out[ooff] = scale * acc;
    }
}

// Few outputs, many slices: with a thread per output the slices are summed by a handful of
// threads one dependent load after the other (2048 slices x ~0.4 us measured: longer than the
// reduction pass itself).  Here 16 row-lanes split the slices of 16 neighbouring outputs, eight
// loads in flight each, and meet in LDS in fixed order.
__global__ void __launch_bounds__(NT)
sum_multiply_finish_few_kernel(Iter it, int nsplit, double scale,
                               const double *__restrict__ partial, double *__restrict__ out)
{
    __shared__ double tile[16][17];
    const int kx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int64_t o = (int64_t)blockIdx.x * 16 + kx;
    const bool act = o < it.nkeep;
    double acc = 0.0;
    if (act) {
        int sidx = ry;
        for (; sidx + 7 * 16 < nsplit; sidx += 8 * 16) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(sidx + 16 * u) * it.nkeep + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; sidx < nsplit; sidx += 16) acc += partial[(int64_t)sidx * it.nkeep + o];
    }
    tile[ry][kx] = acc;
    __syncthreads();
    if (ry == 0 && act) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += tile[j][kx];
        int64_t base[MAXIN], ooff;
        decode_keep(it, o, base, ooff);
        out[ooff] = scale * v;
    }
}

// ---------------------------------------------------------------------------
// Elementwise expression VM: a postfix program over <= 6 broadcast operands,
// evaluated with a 4-deep register stack, one HBM pass.
// ---------------------------------------------------------------------------
struct EwiseArgs {
    int ndim, nin, nops;
    int64_t shape[MAXD];
    int64_t stride[MAXIN][MAXD];
    const double *in[MAXIN];
    int32_t ops[VMP_EWISE_MAX_OPS];
    double consts[VMP_EWISE_MAX_CONSTS];
    int64_t total;
    int pairmask;       // VEC instance: operands read as aligned pairs (the others broadcast)
};

// the postfix program of a fused formula at the operand offsets of one element
// (UNI: the program comes from LDS -- vector registers -- although it is the same for all lanes: the
// dispatch is kept on the scalar unit by reading the words back through the first lane)
template <bool UNI, class EA, class LD>
__device__ inline double ewise_program_impl(const EA &a, const int64_t *off, LD ld)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#define PUSH(v) do { s3 = s2; s2 = s1; s1 = s0; s0 = (v); } while (0)
#define BIN(expr) do { const double y = s0, x = s1; s0 = (expr); s1 = s2; s2 = s3; } while (0)
    const int nops = UNI ? __builtin_amdgcn_readfirstlane(a.nops) : a.nops;
    for (int p = 0; p < nops; ++p) {
        const int word = UNI ? __builtin_amdgcn_readfirstlane(a.ops[p]) : a.ops[p];
        const int op = word & 0xff, arg = word >> 8;
        if (UNI) {
            // the interpreter of the queue: a word is dispatched by taken scalar branches; the frequent
            // cheap words first (the compiler's balanced tree over 20 cases costs every word ~5 of them)
            if (op == VMP_OP_IN) { PUSH(ld(arg, (int64_t)0)); continue; }
            if (op == VMP_OP_CONST) { PUSH(a.consts[arg]); continue; }
            if (op == VMP_OP_MUL) { BIN(x * y); continue; }
            if (op == VMP_OP_ADD) { BIN(x + y); continue; }
            if (op == VMP_OP_SUB) { BIN(x - y); continue; }
        }
        switch (op) {
        case VMP_OP_IN:      PUSH(ld(arg, UNI ? (int64_t)0 : off[arg])); break;
        case VMP_OP_CONST:   PUSH(a.consts[arg]); break;
        case VMP_OP_ADD:     BIN(x + y); break;
        case VMP_OP_SUB:     BIN(x - y); break;
        case VMP_OP_MUL:     BIN(x * y); break;
        case VMP_OP_DIV:     BIN(x / y); break;
        case VMP_OP_NEG:     s0 = -s0; break;
        case VMP_OP_LOG:     s0 = log(s0); break;
        case VMP_OP_EXP:     s0 = exp(s0); break;
        case VMP_OP_SQR:     s0 = s0 * s0; break;
        case VMP_OP_SQRT:    s0 = sqrt(s0); break;
        case VMP_OP_RECIP:   s0 = 1.0 / s0; break;
        case VMP_OP_DIGAMMA: s0 = vmp_digamma(s0); break;
        case VMP_OP_LGAMMA:  s0 = vmp_lgamma(s0); break;
        case VMP_OP_TRIGAMMA: s0 = vmp_trigamma(s0); break;
        case VMP_OP_MAX:     BIN(fmax(x, y)); break;
        case VMP_OP_MIN:     BIN(fmin(x, y)); break;
        case VMP_OP_WHERE_NZ: BIN((x != 0.0) ? y : 0.0); break;   // 0 * -inf guard
        case VMP_OP_DUP:     PUSH(s0); break;
        case VMP_OP_SWAP:    { const double tmp = s0; s0 = s1; s1 = tmp; } break;
        default: break;
        }
    }
#undef PUSH
#undef BIN
    return s0;
}

template <class EA>
__device__ inline double ewise_program(const EA &a, const int64_t *off)
{
    return ewise_program_impl<false>(a, off, [&](int arg, int64_t o) { return a.in[arg][o]; });
}

// one output element of a fused formula: flat index -> operand offsets, then the postfix program
__device__ inline double ewise_element(const EwiseArgs &a, int64_t e)
{
    int64_t off[MAXIN];
    for (int i = 0; i < a.nin; ++i) off[i] = 0;
    int64_t t = e;
    for (int d = a.ndim - 1; d >= 0; --d) {
        const int64_t q = t / a.shape[d];
        const int64_t c = t - q * a.shape[d];
        t = q;
        for (int i = 0; i < a.nin; ++i) off[i] += c * a.stride[i][d];
    }
    return ewise_program(a, off);
}

__global__ void __launch_bounds__(NT)
ewise_kernel(EwiseArgs a, double *__restrict__ out)
{
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < a.total;
         e += (int64_t)gridDim.x * NT)
        out[e] = ewise_element(a, e);
}

// Fast path: after dimension merging the iteration space has one or two axes.  A thread
// evaluates NE elements at once: all operand loads of the NE elements are issued before the
// program is interpreted (they overlap instead of serialising inside the dispatch loop), and
// the per-operation dispatch is paid once per NE elements.
constexpr int EW_NE = 4;

// One shared body per special function for the fast instances: inlined into the four-element
// interpreter they cost ~100 VGPRs, i.e. half of the wavefronts that keep loads in flight.
__device__ __noinline__ double ew_digamma(double x) { return vmp_digamma(x); }
__device__ __noinline__ double ew_lgamma(double x) { return vmp_lgamma(x); }
__device__ __noinline__ double ew_trigamma(double x) { return vmp_trigamma(x); }

// VEC: the innermost axis is contiguous (or broadcast) in every operand and holds an even number
// of elements: the iteration space is in PAIRS (a.shape / a.total / the innermost strides are
// set up that way by the launcher), every load and store moves 16 bytes per lane and the index
// arithmetic is paid once per pair.
// NI: operand slots compiled in (2, 4 or MAXIN): registers, and with them the number of
// wavefronts that keep loads in flight, scale with it.
template <int NDIM, bool I32, bool VEC = false, int NI = MAXIN>
__global__ void __launch_bounds__(NT)
ewise_small_kernel(EwiseArgs a, double *__restrict__ out)
{
    constexpr int NJ = VEC ? EW_NE / 2 : EW_NE;       // index decodes per thread and round
    // a workgroup owns NJ * NT consecutive elements (pairs) per round: its NJ loads per operand
    // are NT apart, rounds are a whole grid apart
    constexpr int64_t span = NT;
    for (int64_t e0 = (int64_t)blockIdx.x * (NT * NJ) + threadIdx.x; e0 < a.total;
         e0 += (int64_t)gridDim.x * (NT * NJ)) {
        double v[NI][EW_NE];
        bool ok[EW_NE];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int64_t e = e0 + j * span;
            const bool okj = e < a.total;
            if constexpr (!VEC) ok[j] = okj;
            const int64_t ee = okj ? e : 0;
            // flat index -> one coordinate per (merged) axis; 32-bit divisions when they fit
            int64_t off[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) off[i] = 0;
            if (I32) {
                uint32_t t = (uint32_t)ee;
#pragma unroll
                for (int d = NDIM - 1; d >= 1; --d) {
                    const uint32_t sz = (uint32_t)a.shape[d];
                    const uint32_t q = t / sz, c = t - q * sz;
                    t = q;
#pragma unroll
                    for (int i = 0; i < NI; ++i) off[i] += (int64_t)c * a.stride[i][d];
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) off[i] += (int64_t)t * a.stride[i][0];
            } else {
                int64_t t = ee;
#pragma unroll
                for (int d = NDIM - 1; d >= 1; --d) {
                    const int64_t q = t / a.shape[d], c = t - q * a.shape[d];
                    t = q;
#pragma unroll
                    for (int i = 0; i < NI; ++i) off[i] += c * a.stride[i][d];
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) off[i] += t * a.stride[i][0];
            }
            if constexpr (VEC) {
                ok[2 * j] = ok[2 * j + 1] = okj;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if (i < a.nin) {
                        if ((a.pairmask >> i) & 1) {
                            const v2f64 q = *reinterpret_cast<const v2f64 *>(a.in[i] + off[i]);
                            v[i][2 * j] = q[0];
                            v[i][2 * j + 1] = q[1];
                        } else {
                            v[i][2 * j] = v[i][2 * j + 1] = a.in[i][off[i]];
                        }
                    } else {
                        v[i][2 * j] = v[i][2 * j + 1] = 0.0;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) v[i][j] = (i < a.nin) ? a.in[i][off[i]] : 0.0;
            }
        }
        double s0[EW_NE], s1[EW_NE], s2[EW_NE], s3[EW_NE];
#pragma unroll
        for (int j = 0; j < EW_NE; ++j) s0[j] = s1[j] = s2[j] = s3[j] = 0.0;
#define EACH for (int j = 0; j < EW_NE; ++j)
#define PUSH(val) _Pragma("unroll") EACH { s3[j] = s2[j]; s2[j] = s1[j]; s1[j] = s0[j]; s0[j] = (val); }
#define BIN(expr) _Pragma("unroll") EACH { const double y = s0[j], x = s1[j]; s0[j] = (expr); \
                                           s1[j] = s2[j]; s2[j] = s3[j]; }
#define UNA(expr) _Pragma("unroll") EACH { const double x = s0[j]; s0[j] = (expr); }
        for (int p = 0; p < a.nops; ++p) {
            const int op = a.ops[p] & 0xff, arg = a.ops[p] >> 8;
            switch (op) {
            case VMP_OP_IN:
                // operand index is wave-uniform: a short chain of selects keeps v[] in registers
                if constexpr (NI <= 2) {
                    PUSH(arg == 0 ? v[0][j] : v[1][j]);
                } else if constexpr (NI <= 4) {
                    PUSH(arg == 0 ? v[0][j] : arg == 1 ? v[1][j] : arg == 2 ? v[2][j] : v[3][j]);
                } else {
                    PUSH(arg == 0 ? v[0][j] : arg == 1 ? v[1][j] : arg == 2 ? v[2][j]
                         : arg == 3 ? v[3][j] : arg == 4 ? v[4][j] : v[5][j]);
                }
                break;
            case VMP_OP_CONST:   PUSH(a.consts[arg]); break;
            case VMP_OP_ADD:     BIN(x + y); break;
            case VMP_OP_SUB:     BIN(x - y); break;
            case VMP_OP_MUL:     BIN(x * y); break;
            case VMP_OP_DIV:     BIN(x / y); break;
            case VMP_OP_NEG:     UNA(-x); break;
            case VMP_OP_LOG:     UNA(log(x)); break;
            case VMP_OP_EXP:     UNA(exp(x)); break;
            case VMP_OP_SQR:     UNA(x * x); break;
            case VMP_OP_SQRT:    UNA(sqrt(x)); break;
            case VMP_OP_RECIP:   UNA(1.0 / x); break;
            case VMP_OP_DIGAMMA: UNA(ew_digamma(x)); break;
            case VMP_OP_LGAMMA:  UNA(ew_lgamma(x)); break;
            case VMP_OP_TRIGAMMA: UNA(ew_trigamma(x)); break;
            case VMP_OP_MAX:     BIN(fmax(x, y)); break;
            case VMP_OP_MIN:     BIN(fmin(x, y)); break;
            case VMP_OP_WHERE_NZ: BIN((x != 0.0) ? y : 0.0); break;
            case VMP_OP_DUP:     PUSH(s0[j]); break;
            case VMP_OP_SWAP:
                _Pragma("unroll") EACH { const double tmp = s0[j]; s0[j] = s1[j]; s1[j] = tmp; }
                break;
            default: break;
            }
        }
#undef EACH
#undef PUSH
#undef BIN
#undef UNA
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (ok[2 * j]) {
                    v2f64 q;
                    q[0] = s0[2 * j];
                    q[1] = s0[2 * j + 1];
                    *reinterpret_cast<v2f64 *>(out + 2 * (e0 + j * span)) = q;
                }
        } else {
#pragma unroll
            for (int j = 0; j < EW_NE; ++j)
                if (ok[j]) out[e0 + j * span] = s0[j];
        }
    }
}

// ---------------------------------------------------------------------------
// Batched SPD inverse / log-determinant by Gauss-Jordan sweeps in LDS.
// ---------------------------------------------------------------------------
constexpr int SPD_MAXN = 64;
constexpr int SPD_LD = SPD_MAXN + 1;

// one workgroup of NTH threads per matrix (n <= 64): Gauss-Jordan sweeps in LDS.  The arithmetic of
// an element does not depend on NTH (the stand-alone kernel runs it with 256 threads, the
// interpreter of queued small operations with its 1024).
template <int NTH>
__device__ inline void spd_block_body(double *M, int *bad, int tid, int n,
                                      const double *__restrict__ a, double *__restrict__ ainv,
                                      double *__restrict__ logdet, int32_t *__restrict__ info)
{
    if (tid == 0) *bad = 0;
    for (int e = tid; e < n * n; e += NTH) {
        const int i = e / n, j = e - i * n;
        M[i * SPD_LD + j] = 0.5 * (a[i * n + j] + a[j * n + i]);
    }
    double ld = 0.0, prod = 1.0;
    constexpr int EPT = SPD_MAXN * SPD_MAXN / NTH;
    for (int p = 0; p < n; ++p) {
        __syncthreads();
        const double piv = M[p * SPD_LD + p];
        double ci[EPT], rj[EPT], me[EPT];
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int e = tid + m * NTH;
            if (e < n * n) {
                const int i = e / n, j = e - i * n;
                ci[m] = M[i * SPD_LD + p];
                rj[m] = M[p * SPD_LD + j];
                me[m] = M[i * SPD_LD + j];
            }
        }
        if (tid == 0) {
            if (!(piv > 0.0)) *bad = 1;
            logdet_accumulate(piv, prod, ld);
        }
        const double d = fast_recip(piv);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int e = tid + m * NTH;
            if (e < n * n) {
                const int i = e / n, j = e - i * n;
                double v;
                if (i == p) v = (j == p) ? d : rj[m] * d;
                else if (j == p) v = -ci[m] * d;
                else v = me[m] - ci[m] * rj[m] * d;
                M[i * SPD_LD + j] = v;
            }
        }
    }
    __syncthreads();
    if (ainv)
        for (int e = tid; e < n * n; e += NTH) {
            const int i = e / n, j = e - i * n;
            ainv[e] = M[i * SPD_LD + j];
        }
    if (tid == 0) {
        if (logdet) *logdet = logdet_finish(prod, ld);
        if (info) *info = *bad;
    }
}

__global__ void __launch_bounds__(NT)
spd_batched_block_kernel(int n, int64_t batch, const double *__restrict__ A,
                         double *__restrict__ Ainv, double *__restrict__ logdet,
                         int32_t *__restrict__ info)
{
    __shared__ double M[SPD_MAXN * SPD_LD];
    __shared__ int bad;
    const int64_t b = blockIdx.x;
    spd_block_body<NT>(M, &bad, threadIdx.x, n, A + b * n * n, Ainv ? Ainv + b * n * n : nullptr,
                       logdet ? logdet + b : nullptr, info ? info + b : nullptr);
}

// one wavefront per matrix (n <= 8): n*n <= 64 elements, one per lane
__global__ void __launch_bounds__(NT)
spd_batched_wave_kernel(int n, int64_t batch, const double *__restrict__ A,
                        double *__restrict__ Ainv, double *__restrict__ logdet,
                        int32_t *__restrict__ info)
{
    __shared__ double Ms[4][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + w;
    const bool act = (b < batch) && (l < n * n);
    const int i = act ? l / n : 0, j = act ? l - i * n : 0;
    double *M = Ms[w];
    double v = 0.0;
    if (act) v = 0.5 * (A[b * n * n + i * n + j] + A[b * n * n + j * n + i]);
    double ld = 0.0, prod = 1.0;
    int bad = 0;
    for (int p = 0; p < n; ++p) {
        M[l] = v;
        __syncthreads();
        const double piv = M[p * n + p];
        const double ci = M[i * n + p], rj = M[p * n + j];
        if (!(piv > 0.0)) bad = 1;
        logdet_accumulate(piv, prod, ld);
        const double d = fast_recip(piv);
        if (i == p) v = (j == p) ? d : rj * d;
        else if (j == p) v = -ci * d;
        else v = v - ci * rj * d;
        __syncthreads();
    }
    if (act && Ainv) Ainv[b * n * n + l] = v;
    if (b < batch && l == 0) {
        if (logdet) logdet[b] = logdet_finish(prod, ld);
        if (info) info[b] = bad;
    }
}

// Throughput form for 8 < n <= 16 and large batches: one matrix per group of NP = 16 lanes
// (exactly one DPP row), lane i holds row i in registers, so a wavefront inverts four matrices
// with no workgroup barrier.  Each Gauss-Jordan step needs the pivot row in every lane of the
// group: v_mov_b64_dpp row_newbcast (pure VALU, the control is an immediate -> static
// unrolling).  Matrices are staged through LDS both ways so the global accesses stay fully
// coalesced.  (16 < n <= 32 runs on the matrix cores: vmp_spd_mfma.hip.)
template <int P>
__device__ __forceinline__ double group_bcast(double x)
{
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + P, 0xf, 0xf, false);
}

template <int NP, int P>
__device__ __forceinline__ void gj_rows_steps(double (&m)[NP], int gl, double &prod, double &ld,
                                              int &bad)
{
    static_assert(NP == 16, "one DPP row per matrix");
    if constexpr (P < NP) {
        double row[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) row[j] = group_bcast<P>(m[j]);
        const double piv = row[P];
        if (!(piv > 0.0)) bad = 1;
        logdet_accumulate(piv, prod, ld);
        const double d = fast_recip(piv);
        const double ci = m[P];
        const double f = (gl == P) ? 0.0 : -ci * d;      // row i: m_i -= (m_ip / piv) row_p
        const double sc = (gl == P) ? d : 0.0;           // pivot row: row_p / piv
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const double base = (gl == P) ? 0.0 : m[j];
            m[j] = (j == P) ? ((gl == P) ? d : f) : base + (f + sc) * row[j];
        }
        gj_rows_steps<NP, P + 1>(m, gl, prod, ld, bad);
    }
}

// x_i = sum_P m[P] * v_P and friends: every lane of the group needs the value held by lane P,
// for all P.
template <int NP, int P>
__device__ __forceinline__ void rows_matvec(const double (&m)[NP], double v, double &acc)
{
    if constexpr (P < NP) {
        acc += m[P] * group_bcast<P>(v);
        rows_matvec<NP, P + 1>(m, v, acc);
    }
}

template <int NP, int P>
__device__ __forceinline__ void rows_rank1(double (&m)[NP], double xi)
{
    if constexpr (P < NP) {
        m[P] += xi * group_bcast<P>(xi);
        rows_rank1<NP, P + 1>(m, xi);
    }
}

template <int NP, int P>
__device__ __forceinline__ void rows_sum(double v, double &acc)
{
    if constexpr (P < NP) {
        acc += group_bcast<P>(v);
        rows_sum<NP, P + 1>(v, acc);
    }
}

// MOMENTS = false: A -> A^-1, log|A| (vmp_spd_batched).
// MOMENTS = true : natural parameters (phi0 = `rhs`, phi1 = `A`) of a Gaussian with full
// covariance per plate -> moments and log-normaliser in the same pass
// (GaussianARDDistribution.compute_moments_and_cgf, gaussian.py:680-706):
//   Cov = (-2 phi1)^-1 ; u0 = Cov phi0 ; u1 = u0 u0^T + Cov ; g = -1/2 u0.phi0 + 1/2 log|-2 phi1|
// with u0 -> `vec_out`, u1 -> `Ainv`, g -> `logdet`.
template <int NP, int NTB, bool MOMENTS>
__global__ void __launch_bounds__(NTB)
spd_batched_rows_kernel(int n, int64_t batch, const double *__restrict__ A,
                        const double *__restrict__ rhs, double *__restrict__ Ainv,
                        double *__restrict__ vec_out, double *__restrict__ logdet,
                        int32_t *__restrict__ info)
{
    constexpr int MPW = 64 / NP;             // matrices per wavefront
    constexpr int MPB = MPW * (NTB / 64);     // matrices per workgroup
    constexpr int LDP = NP + 1;
    __shared__ double Ms[MPB * NP * LDP];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int g = l / NP, gl = l % NP;
    const int nn = n * n;
    for (int64_t b0 = (int64_t)blockIdx.x * MPB; b0 < batch; b0 += (int64_t)gridDim.x * MPB) {
        const int nb = (int)((batch - b0) < MPB ? (batch - b0) : MPB);
        __syncthreads();
        for (int e = tid; e < nb * nn; e += NTB) {
            const int mb = e / nn, r = e - mb * nn;
            const int i = r / n, j = r - i * n;
            Ms[(mb * NP + i) * LDP + j] = __builtin_nontemporal_load(A + b0 * nn + e);
        }
        __syncthreads();
        const int mb = w * MPW + g;
        const bool act = mb < nb;
        double *M = Ms + mb * NP * LDP;
        constexpr double SC = MOMENTS ? -1.0 : 0.5;      // -2 phi1, symmetrised / symmetrise
        double m[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j)
            m[j] = (act && gl < n && j < n) ? SC * (M[gl * LDP + j] + M[j * LDP + gl])
                                            : ((gl == j) ? 1.0 : 0.0);
        double ld = 0.0, prod = 1.0;
        int bad = 0;
        gj_rows_steps<NP, 0>(m, gl, prod, ld, bad);
        double lg = logdet_finish(prod, ld);
        if constexpr (MOMENTS) {
            const double p0 = (act && gl < n) ? rhs[(b0 + mb) * n + gl] : 0.0;
            double x = 0.0;
            rows_matvec<NP, 0>(m, p0, x);
            double s = 0.0;
            rows_sum<NP, 0>(x * p0, s);
            rows_rank1<NP, 0>(m, x);
            if (act && gl < n) vec_out[(b0 + mb) * n + gl] = x;
            lg = -0.5 * s + 0.5 * lg;
        }
        __syncthreads();
        if (act && gl < n) {
#pragma unroll
            for (int j = 0; j < NP; ++j)
                if (j < n) M[gl * LDP + j] = m[j];
        }
        if (act && gl == 0) {
            if (logdet) logdet[b0 + mb] = lg;
            if (info) info[b0 + mb] = bad;
        }
        __syncthreads();
        if (Ainv)
            for (int e = tid; e < nb * nn; e += NTB) {
                const int mb2 = e / nn, r = e - mb2 * nn;
                const int i = r / n, j = r - i * n;
                Ainv[b0 * nn + e] = Ms[(mb2 * NP + i) * LDP + j];
            }
    }
}

// ---------------------------------------------------------------------------
// Row softmax with the reference's recipe: stable log-sum-exp, exp, then a second
// renormalisation (utils/misc.py:1388-1401).
// K <= 256: LPR lanes per row, four elements per lane held in registers: ONE pass over HBM
// (16-byte accesses when K is even), one exponential per element (exp(x - lse) =
// exp(x - max) / sum), reductions over the LPR lanes by xor-shuffles.
// ---------------------------------------------------------------------------
template <int LPR, bool VEC>
__global__ void __launch_bounds__(NT)
softmax_rows_kernel(int64_t rows, int K, const double *__restrict__ phi, double *__restrict__ p,
                    double *__restrict__ lse)
{
    constexpr int RPB = NT / LPR;                  // rows per workgroup and round
    const int lr = threadIdx.x % LPR, rt = threadIdx.x / LPR;
    const int64_t rounds = (rows + (int64_t)gridDim.x * RPB - 1) / ((int64_t)gridDim.x * RPB);
    for (int64_t t = 0; t < rounds; ++t) {
        const int64_t r = (t * gridDim.x + blockIdx.x) * RPB + rt;
        const bool rok = r < rows;
        const double *x = phi + (rok ? r : 0) * K;
        double v[4];
        bool ok[4];
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = 2 * (lr + LPR * j);
                ok[2 * j] = ok[2 * j + 1] = rok && k < K;
                v2f64 q = {-INFINITY, -INFINITY};
                if (ok[2 * j]) q = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(x + k));
                v[2 * j] = q[0];
                v[2 * j + 1] = q[1];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lr + LPR * j;
                ok[j] = rok && k < K;
                v[j] = ok[j] ? x[k] : -INFINITY;
            }
        }
        double mx = fmax(fmax(v[0], v[1]), fmax(v[2], v[3]));
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
        if (!isfinite(mx)) mx = 0.0;                       // misc.py:1375-1378
        double e[4], s = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = ok[j] ? exp_nonpos(v[j] - mx) : 0.0;
            s += e[j];
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const double ls = log(s) + mx;
        const double is = 1.0 / s;
        double s2 = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] *= is;
            s2 += e[j];
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) s2 += __shfl_xor(s2, off, 64);
        const double inv = 1.0 / s2;
        double *o = p + (rok ? r : 0) * K;
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (ok[2 * j]) {
                    v2f64 q;
                    q[0] = e[2 * j] * inv;
                    q[1] = e[2 * j + 1] * inv;
                    __builtin_nontemporal_store(q, reinterpret_cast<v2f64 *>(o + 2 * (lr + LPR * j)));
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ok[j]) o[lr + LPR * j] = e[j] * inv;
        }
        if (rok && lr == 0 && lse) lse[r] = ls;
    }
}

// K > 256: one wavefront per row, the row is re-read from cache.
__global__ void __launch_bounds__(NT)
softmax_kernel(int64_t rows, int K, const double *__restrict__ phi, double *__restrict__ p,
               double *__restrict__ lse)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < rows; r += (int64_t)gridDim.x * 4) {
        const double *x = phi + r * K;
        double mx = -INFINITY;
        for (int k = l; k < K; k += 64) mx = fmax(mx, x[k]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
        if (!isfinite(mx)) mx = 0.0;                       // misc.py:1375-1378
        double s = 0.0;
        for (int k = l; k < K; k += 64) s += exp_nonpos(x[k] - mx);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const double ls = log(s) + mx;
        const double is = 1.0 / s;
        double s2 = 0.0;
        for (int k = l; k < K; k += 64) s2 += exp_nonpos(x[k] - mx) * is;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off, 64);
        const double inv = is / s2;
        for (int k = l; k < K; k += 64) p[r * K + k] = exp_nonpos(x[k] - mx) * inv;
        if (l == 0 && lse) lse[r] = ls;
    }
}

__global__ void __launch_bounds__(NT)
onehot_kernel(int64_t n, int K, const int64_t *__restrict__ labels, double *__restrict__ out,
              int32_t *__restrict__ info)
{
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n * K;
         e += (int64_t)gridDim.x * NT) {
        const int64_t r = e / K;
        const int k = (int)(e - r * K);
        const int64_t lab = labels[r];
        if (k == 0 && (lab < 0 || lab >= K)) atomicOr(info, 1);
        out[e] = (lab == k) ? 1.0 : 0.0;
    }
}

int64_t grid_for(vmp_ctx *ctx, int64_t work_items, int per_block)
{
    int64_t g = (work_items + per_block - 1) / per_block;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return g;
}

void launch_finish(vmp_ctx *ctx, const Iter &it, int nsplit, double scale, const double *partial,
                   double *out)
{
    if (it.nkeep < 4096 && nsplit >= 32)
        hipLaunchKernelGGL(sum_multiply_finish_few_kernel, dim3((unsigned)((it.nkeep + 15) / 16)),
                           dim3(NT), 0, ctx->stream, it, nsplit, scale, partial, out);
    else
        hipLaunchKernelGGL(sum_multiply_finish_kernel,
                           dim3((unsigned)grid_for(ctx, it.nkeep, NT)), dim3(NT), 0, ctx->stream,
                           it, nsplit, scale, partial, out);
}

// ---------------------------------------------------------------------------
// The queue of small operations.  A sweep of the generic engine issues ~90 formulas and plate
// sums on scalars and K x K arrays between its few plate-sized kernels; launched one by one they
// are 5-30 us each of launch latency for < 1 us of work.  While the queue is open such calls are
// recorded on the host (the very EwiseArgs / Iter the kernels take) and ONE launch of
// small_ops_kernel -- a single workgroup that walks the records in order, a barrier between two
// of them -- runs them when something else needs the stream (any other entry point of this
// library flushes first), when the queue is full, or at vmp_queue_flush / vmp_queue_end.
// Results do not depend on how the operations are grouped into launches.
// Measured (bench.py --config generic_pca / generic_gmm, N = 1e6): eager sweeps 4.22 -> 3.83 ms /
// 3.07 -> 3.04 ms; sweeps replayed from a HIP graph get SLOWER (2.10 -> 2.14, 1.76 -> 1.95 ms: a
// graph node costs what one record costs the interpreter, a dependent round trip to memory), so
// the recorder of the generic engine switches the queue off (tune keys small_queue_ew / _sm) and
// the formulas -- whose interpreter arithmetic is that of ewise_kernel, bit for bit -- are the
// default; queued plate sums use another summation order than the stand-alone kernels and stay
// an opt-in (small_queue_sm = 1).
// ---------------------------------------------------------------------------
enum { SMALL_EWISE = 0, SMALL_SUM = 1, SMALL_SPD = 2 };
struct SmallSpd {
    int n;
    int64_t batch;
    const double *A;
    double *Ainv, *logdet;
    int32_t *info;
};
struct alignas(16) SmallOp {
    int32_t kind, pad;
    double scale;
    double *out;
    union {
        EwiseArgs ew;
        Iter it;
        SmallSpd spd;
    };
    // filled in by the flush (queue_place): where in the interpreter's LDS arena operand i / the
    // result live (element offsets; -1: in global memory only)
    int32_t lin[MAXIN];
    int32_t lout;
    int32_t fence;       // the record reads MEMORY that an earlier record of the launch wrote
};
// an array from OUTSIDE the launch that its records read: copied into the arena before the first record
struct SmallPre {
    const double *src;
    int32_t off, n;
};
constexpr int QNT = 512;                         // threads of the interpreter (256 VGPRs each: no spills)
constexpr int QUEUE_CAP = 128;                   // records per launch
constexpr int64_t SMALL_EW_MAX = 2048;           // elements of a queued formula
constexpr int64_t SMALL_SM_KEEP = 2048, SMALL_SM_WORK = 32768;   // outputs, products of a queued sum
constexpr int SMALL_SPD_MAXN = 32;               // a queued inverse / log-determinant: n x n, n <= 32
constexpr int64_t SMALL_SPD_BATCH = 4;

constexpr int QUEUE_SLOTS = 16;                  // staging buffers in rotation (eager flushes)
constexpr int ARENA_RECORDS = 16384;             // records of flushes recorded into HIP graphs
constexpr int QLDS = 14336;                      // doubles of the interpreter's LDS arena (112 KB)
constexpr int PRE_CAP = 256;                     // arrays copied in per launch
constexpr int ARENA_PRE = 2 * ARENA_RECORDS;

struct small_queue {
    int open;
    int n;                    // records collected in the current slot
    int cur;                  // slot being filled
    SmallOp *host;            // pinned: QUEUE_SLOTS x QUEUE_CAP records
    SmallOp *dev;             // the device copies the kernel reads
    hipEvent_t done[QUEUE_SLOTS];
    int pending[QUEUE_SLOTS];
    // a flush inside a stream capture is replayed with the graph: its records must stay where they
    // are for as long as the graph lives, so they are moved into an arena that is never reused
    // (allocated with the queue: nothing can be allocated while a stream records).  Their device
    // copy is made ONCE, by vmp_queue_commit after the recording -- not by a copy node that every
    // replay would pay for
    SmallOp *arena_host, *arena_dev;
    int arena_used, arena_committed;
    int64_t launches, ops;
    // the tables of arrays copied into the LDS arena, one per flush, kept like the records
    SmallPre *pre_host, *pre_dev, *pre_arena_host, *pre_arena_dev;
    int pre_arena_used, pre_arena_committed;
    int64_t cached_in, global_in;      // operands of the records read from the arena / from memory
};

// The records of a launch are staged through LDS, RC at a time, by the whole workgroup (one memory
// latency per RC records).  Read one field at a time where the program flow needs it -- as scalar
// loads from the constant address space, the form of round 5 -- every record cost ~18 dependent cache
// misses, 3.8 us per record in a recorded sweep (profiles/r06/queue_record_cost.txt), more than the graph
// node it replaced.  Values that steer the control flow come back through the first lane.
#define VMP_CONST_AS __attribute__((address_space(3)))
typedef const VMP_CONST_AS SmallOp CSmallOp;

__device__ inline CSmallOp *uniform_record(CSmallOp *op) { return op; }
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// one element of a queued formula: everything fits 32 bits here (total <= SMALL_EW_MAX), the
// arithmetic is ewise_element's
template <class EA>
__device__ inline double ewise_element32(const EA &a, uint32_t e)
{
    int64_t off[MAXIN];
    const int nin = a.nin, ndim = a.ndim;
    for (int i = 0; i < nin; ++i) off[i] = 0;
    uint32_t t = e;
    for (int d = ndim - 1; d >= 1; --d) {
        const uint32_t sz = (uint32_t)a.shape[d];
        const uint32_t q = t / sz, c = t - q * sz;
        t = q;
        for (int i = 0; i < nin; ++i) off[i] += (int64_t)c * a.stride[i][d];
    }
    if (ndim >= 1)
        for (int i = 0; i < nin; ++i) off[i] += (int64_t)t * a.stride[i][0];
    return ewise_program(a, off);
}

__device__ inline double wave_sum(double v)
{
    for (int st = 32; st > 0; st >>= 1) v += __shfl_down(v, st, 64);
    return v;
}

// the three kinds of records, each a function of its own (their register needs differ widely; as
// one body the interpreter spilled)
// Operand i of a record: in the LDS arena when the flush placed it there (lin >= 0), else where the
// record says.  Both are reached through one generic pointer (a flat access resolves the aperture).
__device__ inline const double *small_operand(CSmallOp *op, const double *glob, int i, const double *lds)
{
    const int32_t li = op->lin[i];
    return li >= 0 ? lds + li : glob;
}

// The program of a queued formula held ACROSS THE LANES of a wavefront -- lane p has word p, lane c
// constant c -- and read back with v_readlane: a word costs a few cycles instead of a dependent LDS
// round trip (0.17 us per word before: a 33-word formula on a scalar was 6.8 us).
struct ProgWords {
    int w;
    __device__ int operator[](int p) const { return __builtin_amdgcn_readlane(w, p); }
};
struct ProgConsts {
    double c;
    __device__ double operator[](int a) const
    {
        const int lo = __builtin_amdgcn_readlane(__double2loint(c), a);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(c), a);
        return __hiloint2double(hi, lo);
    }
};
struct ProgRegs {
    int nops;
    ProgWords ops;
    ProgConsts consts;
};
static_assert(VMP_EWISE_MAX_OPS <= 64 && VMP_EWISE_MAX_CONSTS <= 64, "one lane per program word");

// (no private array is indexed with a run-time value here: `off[arg]` with arg from the program put the
// offsets into scratch MEMORY -- a vector-memory round trip per operand read, and with it a wait for
// every store in flight: 2.5 us for a scalar formula)
__device__ __forceinline__ void small_ewise(CSmallOp *op, int tid, double *lds)
{
    const uint32_t total = (uint32_t)uni((int)op->ew.total);
    double *out = op->out;
    const int nin = uni(op->ew.nin), ndim = uni(op->ew.ndim);
    static_assert(MAXIN == 6, "operand selection below is written for six operands");
    const double *base[MAXIN];
#pragma unroll
    for (int i = 0; i < MAXIN; ++i) base[i] = small_operand(op, op->ew.in[i], i, lds);
    const int32_t lout = op->lout;
    ProgRegs prog;
    {
        const int lane = tid & 63;
        prog.nops = uni(op->ew.nops);
        prog.ops.w = lane < VMP_EWISE_MAX_OPS ? op->ew.ops[lane] : 0;
        prog.consts.c = lane < VMP_EWISE_MAX_CONSTS ? op->ew.consts[lane] : 0.0;
    }
    for (uint32_t e = tid; e < total; e += QNT) {
        int64_t off[MAXIN];
#pragma unroll
        for (int i = 0; i < MAXIN; ++i) off[i] = 0;
        uint32_t t = e;
        for (int d = ndim - 1; d >= 1; --d) {
            const uint32_t sz = (uint32_t)op->ew.shape[d];
            const uint32_t q = t / sz, c = t - q * sz;
            t = q;
#pragma unroll
            for (int i = 0; i < MAXIN; ++i)
                if (i < nin) off[i] += (int64_t)c * op->ew.stride[i][d];
        }
        if (ndim >= 1) {
#pragma unroll
            for (int i = 0; i < MAXIN; ++i)
                if (i < nin) off[i] += (int64_t)t * op->ew.stride[i][0];
        }
        const double *p0 = base[0] + off[0], *p1 = base[1] + off[1], *p2 = base[2] + off[2];
        const double *p3 = base[3] + off[3], *p4 = base[4] + off[4], *p5 = base[5] + off[5];
        auto ld = [&](int arg, int64_t) {
            const double *p = arg == 0 ? p0 : (arg == 1 ? p1 : (arg == 2 ? p2 : (arg == 3 ? p3 : (arg == 4 ? p4 : p5))));
            return *p;
        };
        const double v = ewise_program_impl<true>(prog, nullptr, ld);
        out[e] = v;
        if (lout >= 0) lds[lout + e] = v;
    }
}

__device__ __forceinline__ void small_spd(CSmallOp *op, double *M, int *bad, int tid)
{
    op = uniform_record(op);
    const int n = op->spd.n, nn = n * n;
    const int64_t batch = op->spd.batch;
    const double *A = op->spd.A;
    double *Ainv = op->spd.Ainv, *logdet = op->spd.logdet;
    int32_t *info = op->spd.info;
    for (int64_t b = 0; b < batch; ++b) {
        spd_block_body<QNT>(M, bad, tid, n, A + b * nn, Ainv ? Ainv + b * nn : nullptr,
                            logdet ? logdet + b : nullptr, info ? info + b : nullptr);
        __syncthreads();
    }
}

// A queued sum of products.  The latency of one dependent load (~1 us from L2 / HBM) is what a
// small operation costs, so the products of a lane are formed EIGHT at a time (their loads in
// flight together), G lanes share an output (G a power of two <= 64 chosen so that the workgroup
// is filled), and the partial sums of a group meet by xor-shuffles.  The order of the additions
// is fixed by the shape of the operation alone.
template <int NIN>
__device__ inline double small_sum_lane(const VMP_CONST_AS Iter &it, const double *const *in,
                                        const int64_t *base, int64_t first, int64_t step,
                                        int64_t nred)
{
    const int nin = it.nin, nr = it.nr;
    double acc = 0.0;
    for (int64_t r0 = first; r0 < nred; r0 += 8 * step) {
        double p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = r0 + u * step;
            const bool ok = r < nred;
            int64_t off[NIN];
#pragma unroll
            for (int i = 0; i < NIN; ++i) off[i] = base[i];
            uint32_t t = ok ? (uint32_t)r : 0u;        // (nred <= 32768: 32 bits)
            for (int d = nr - 1; d >= 1; --d) {
                const uint32_t sz = (uint32_t)it.rsize[d];
                const uint32_t q = t / sz, c = t - q * sz;
                t = q;
#pragma unroll
                for (int i = 0; i < NIN; ++i)
                    if (i < nin) off[i] += (int64_t)c * it.rstride[i][d];
            }
            if (nr >= 1) {
#pragma unroll
                for (int i = 0; i < NIN; ++i)
                    if (i < nin) off[i] += (int64_t)t * it.rstride[i][0];
            }
            double v = in[0][off[0]];
#pragma unroll
            for (int i = 1; i < NIN; ++i)
                if (i < nin) v *= in[i][off[i]];
            p[u] = ok ? v : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += p[u];
    }
    return acc;
}

template <int NIN>
__device__ inline void small_sum_body(CSmallOp *op, double *red, int tid, double *lds)
{
    const VMP_CONST_AS Iter &it = op->it;
    double *out = op->out;
    const double scale = op->scale;
    const int64_t nkeep = it.nkeep, nred = it.nred;
    const int32_t lout = op->lout;
    const double *in[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int k = i < it.nin ? i : 0;
        in[i] = small_operand(op, it.in[k], k, lds);
    }
    if (nkeep * 64 < QNT && nred > 1024) {
        // a few long sums: the workgroup per output, wavefront sums combined in a fixed order
        const int lane = tid & 63, wave = tid >> 6;
        for (int64_t o = 0; o < nkeep; ++o) {
            int64_t base[MAXIN], ooff;
            decode_keep(it, o, base, ooff);
            double acc = small_sum_lane<NIN>(it, in, base, tid, QNT, nred);
            acc = wave_sum(acc);
            if (lane == 0) red[wave] = acc;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int w = 0; w < QNT / 64; ++w) t += red[w];
                out[ooff] = scale * t;
                if (lout >= 0) lds[lout + ooff] = scale * t;
            }
            __syncthreads();
        }
        return;
    }
    int G = 1;
    while (G < 64 && (int64_t)(2 * G) * nkeep <= QNT && 2 * G <= nred) G <<= 1;
    const int per = QNT / G, gl = tid & (G - 1), slot = tid / G;
    for (int64_t o0 = 0; o0 < nkeep; o0 += per) {
        const int64_t o = o0 + slot;
        const bool act = o < nkeep;
        int64_t base[MAXIN], ooff;
        decode_keep(it, act ? o : 0, base, ooff);
        double acc = small_sum_lane<NIN>(it, in, base, gl, G, nred);
        for (int st = G >> 1; st > 0; st >>= 1) acc += __shfl_xor(acc, st, 64);
        if (act && gl == 0) {
            out[ooff] = scale * acc;
            if (lout >= 0) lds[lout + ooff] = scale * acc;
        }
    }
}

__device__ __forceinline__ void small_sum(CSmallOp *op, double *red, int tid, double *lds)
{
    op = uniform_record(op);
    if (op->it.nin <= 2) small_sum_body<2>(op, red, tid, lds);
    else small_sum_body<MAXIN>(op, red, tid, lds);
}

// The interpreter keeps the small arrays of its launch in LDS: arrays from outside that its records
// read are copied in first (all of them in flight together: one memory latency for the launch), a
// record's result is written to memory AND to the arena, and a record reads whatever an earlier
// record of the launch produced -- or what was copied in -- from the arena.  A record then costs a
// barrier and LDS latency instead of a dependent round trip through memory (2-3 us, which was also
// what a node of a HIP graph costs: the reason the queue did not pay inside recorded sweeps).
__global__ void __launch_bounds__(QNT)
small_ops_kernel(const SmallOp *__restrict__ ops, int n, const SmallPre *__restrict__ pre, int npre,
                 long long *__restrict__ prof)
{
    extern __shared__ double qlds[];
    __shared__ double red[QNT / 64];
    __shared__ double M[SMALL_SPD_MAXN * SPD_LD];
    __shared__ int bad;
    constexpr int RC = 16, RW = (int)(sizeof(SmallOp) / sizeof(double));
    __shared__ double recbuf[RC * RW];
    const int tid = threadIdx.x;
    {
        // a wavefront per array, eight arrays in flight per round
        const int w = tid >> 6, l = tid & 63;
        for (int p0 = 0; p0 < npre; p0 += QNT / 64) {
            const int p = p0 + w;
            if (p < npre) {
                const double *src = pre[p].src;
                const int off = pre[p].off, cnt = pre[p].n;
                for (int e = l; e < cnt; e += 64) qlds[off + e] = src[e];
            }
        }
        __syncthreads();
    }
    const double *words = reinterpret_cast<const double *>(ops);
    // (measurement only, tune key small_queue_prof: cycles of lane 0 per stage of a record)
    long long pc[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long long c0 = prof ? (long long)__builtin_readcyclecounter() : 0;
        if (i % RC == 0) {
            const int cnt = (n - i < RC ? n - i : RC) * RW;
            for (int e = tid; e < cnt; e += QNT) recbuf[e] = words[(int64_t)i * RW + e];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        CSmallOp *op = (CSmallOp *)(recbuf + (i % RC) * RW);          // (C cast: generic -> LDS address space)
        const int kind = uni(op->kind);
        if (uni(op->fence)) {
            // it reads memory an earlier record wrote: those stores have to have arrived (a round
            // trip of ~2 us -- which EVERY record paid while this was unconditional)
            __threadfence_block();
            __syncthreads();
        }
        const long long c1 = prof ? (long long)__builtin_readcyclecounter() : 0;
        if (kind == SMALL_EWISE) small_ewise(op, tid, qlds);
        else if (kind == SMALL_SPD) small_spd(op, M, &bad, tid);
        else small_sum(op, red, tid, qlds);
        // what this record wrote INTO THE ARENA is visible to the next one: LDS operations done,
        // then the barrier -- without waiting for the stores to memory (see above)
        const long long c2 = prof ? (long long)__builtin_readcyclecounter() : 0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (prof) {
            const long long c3 = (long long)__builtin_readcyclecounter();
            pc[0] += c1 - c0;
            pc[1] += c2 - c1;
            pc[2] += c3 - c2;
            pc[3] += 1;
        }
    }
    if (prof && tid == 0) {
        prof[0] = pc[0];
        prof[1] = pc[1];
        prof[2] = pc[2];
        prof[3] = pc[3];
    }
}

// ---- placement of a launch's small arrays in the interpreter's LDS arena (host, at the flush) ----
struct ByteRange {
    const char *lo, *hi;           // [lo, hi)
    int32_t off;                   // arena offset of lo (elements), or -1: not in the arena
};

// elements an operand touches: [lo, hi] relative to its base pointer
inline void span_of(int nd, const int64_t *size, const int64_t *stride, int64_t &lo, int64_t &hi)
{
    for (int d = 0; d < nd; ++d) {
        const int64_t ext = (size[d] - 1) * stride[d];
        if (ext < 0) lo += ext; else hi += ext;
    }
}

// Fills lin / lout of the n records and the table of arrays to copy in.  Rules: an operand that lies
// inside the result of an earlier record of the launch is read from that result's place in the arena
// (from memory if the result has no place there: the barrier + fence between records makes the write
// visible); an operand that overlaps no earlier result is an array from outside -- copied in when it
// is small and there is room; results of inverses (SMALL_SPD) are never placed.
inline void queue_place(SmallOp *h, int n, SmallPre *pre, int *npre_out, int64_t *cached, int64_t *uncached)
{
    ByteRange outs[2 * QUEUE_CAP + 8];      // results of the records so far (inverses: two each)
    ByteRange ext[PRE_CAP];
    int nout = 0, next = 0, used = 0;
    const int64_t PRE_MAX = 2048;
    auto alloc = [&](int64_t cnt) -> int32_t {
        const int64_t c = (cnt + 1) & ~(int64_t)1;       // 16-byte granules
        if (used + c > QLDS) return -1;
        const int32_t o = used;
        used += (int)c;
        return o;
    };
    for (int i = 0; i < n; ++i) {
        SmallOp &op = h[i];
        for (int k = 0; k < MAXIN; ++k) op.lin[k] = -1;
        op.lout = -1;
        op.fence = 0;
        if (op.kind != SMALL_SPD) {
            const int nin = op.kind == SMALL_EWISE ? op.ew.nin : op.it.nin;
            for (int k = 0; k < nin; ++k) {
                const double *base = op.kind == SMALL_EWISE ? op.ew.in[k] : op.it.in[k];
                int64_t lo = 0, hi = 0;
                if (op.kind == SMALL_EWISE) {
                    span_of(op.ew.ndim, op.ew.shape, op.ew.stride[k], lo, hi);
                } else {
                    span_of(op.it.nk, op.it.ksize, op.it.kstride[k], lo, hi);
                    span_of(op.it.nr, op.it.rsize, op.it.rstride[k], lo, hi);
                }
                const char *blo = reinterpret_cast<const char *>(base + lo);
                const char *bhi = reinterpret_cast<const char *>(base + hi + 1);
                bool overlaps = false;
                int32_t place = -1;
                for (int j = nout - 1; j >= 0; --j) {
                    if (blo < outs[j].hi && outs[j].lo < bhi) {
                        overlaps = true;
                        if (outs[j].off >= 0 && outs[j].lo <= blo && bhi <= outs[j].hi)
                            place = outs[j].off + (int32_t)((reinterpret_cast<const char *>(base) - outs[j].lo) / 8);
                        break;                        // the latest writer decides
                    }
                }
                if (!overlaps) {
                    for (int j = 0; j < next; ++j)
                        if (ext[j].lo <= blo && bhi <= ext[j].hi) {
                            place = ext[j].off + (int32_t)((reinterpret_cast<const char *>(base) - ext[j].lo) / 8);
                            break;
                        }
                    const int64_t cnt = hi - lo + 1;
                    if (place < 0 && cnt <= PRE_MAX && next < PRE_CAP) {
                        const int32_t o = alloc(cnt);
                        if (o >= 0) {
                            ext[next] = ByteRange{blo, bhi, o};
                            pre[next] = SmallPre{base + lo, o, (int32_t)cnt};
                            ++next;
                            place = o + (int32_t)(-lo);
                        }
                    }
                }
                op.lin[k] = place;
                if (overlaps && place < 0) op.fence = 1;
                if (place >= 0) *cached += 1; else *uncached += 1;
            }
        }
        // the record's result(s)
        if (op.kind == SMALL_SPD) {
            const int64_t nn = (int64_t)op.spd.n * op.spd.n * op.spd.batch;
            {
                const char *blo = reinterpret_cast<const char *>(op.spd.A);
                const char *bhi = reinterpret_cast<const char *>(op.spd.A + nn);
                for (int j = 0; j < nout; ++j)
                    if (blo < outs[j].hi && outs[j].lo < bhi) op.fence = 1;
            }
            if (op.spd.Ainv)
                outs[nout++] = ByteRange{reinterpret_cast<const char *>(op.spd.Ainv),
                                         reinterpret_cast<const char *>(op.spd.Ainv + nn), -1};
            if (op.spd.logdet)
                outs[nout++] = ByteRange{reinterpret_cast<const char *>(op.spd.logdet),
                                         reinterpret_cast<const char *>(op.spd.logdet + op.spd.batch), -1};
        } else {
            int64_t cnt;
            if (op.kind == SMALL_EWISE) {
                cnt = op.ew.total;
            } else {
                int64_t lo = 0, hi = 0;
                span_of(op.it.nk, op.it.ksize, op.it.okstride, lo, hi);
                cnt = lo < 0 ? -1 : hi + 1;
            }
            int32_t o = -1;
            if (cnt > 0 && cnt <= PRE_MAX) o = alloc(cnt);
            op.lout = o;
            const int64_t bytes = (cnt > 0 ? cnt : 0) * 8;
            outs[nout++] = ByteRange{reinterpret_cast<const char *>(op.out),
                                     reinterpret_cast<const char *>(op.out) + bytes, o};
        }
    }
    *npre_out = next;
}

inline small_queue *queue_of(vmp_ctx *ctx) { return reinterpret_cast<small_queue *>(ctx->queue); }

inline bool stream_records(vmp_ctx *ctx)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &cap);
    return cap == hipStreamCaptureStatusActive;
}

// room for one more record: *slot, or nullptr when the call has to launch on its own (the queue
// is closed, or the arena of recorded flushes is used up)
inline int32_t queue_slot(vmp_ctx *ctx, SmallOp **slot)
{
    *slot = nullptr;
    small_queue *q = queue_of(ctx);
    if (!q || q->open <= 0) return VMP_OK;
    if (q->n == QUEUE_CAP) {
        const int32_t rc = vmp_queue_flush(ctx);
        if (rc != VMP_OK) return rc;
    }
    if (q->arena_used + q->n + 1 > ARENA_RECORDS && stream_records(ctx)) {
        const int32_t rc = vmp_queue_flush(ctx);     // what is collected still fits; then on its own
        if (rc != VMP_OK) return rc;
        return VMP_OK;
    }
    *slot = q->host + (size_t)q->cur * QUEUE_CAP + q->n;
    return VMP_OK;
}

}  // namespace

const char *vmp_flush_cause = nullptr;

int32_t destroy_small_queue(vmp_ctx *ctx)
{
    if (!ctx || !ctx->queue) return VMP_OK;
    small_queue *q = queue_of(ctx);
    for (int i = 0; i < QUEUE_SLOTS; ++i)
        if (q->done[i]) (void)hipEventDestroy(q->done[i]);
    if (q->host) (void)hipHostFree(q->host);
    if (q->dev) (void)hipFree(q->dev);
    if (q->arena_host) (void)hipHostFree(q->arena_host);
    if (q->arena_dev) (void)hipFree(q->arena_dev);
    if (q->pre_host) (void)hipHostFree(q->pre_host);
    if (q->pre_dev) (void)hipFree(q->pre_dev);
    if (q->pre_arena_host) (void)hipHostFree(q->pre_arena_host);
    if (q->pre_arena_dev) (void)hipFree(q->pre_arena_dev);
    delete q;
    ctx->queue = nullptr;
    return VMP_OK;
}

extern "C" {

int32_t vmp_queue_begin(vmp_ctx *ctx)
{
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null context");
    if (vmp_tune_get("small_queue", 1) == 0) return VMP_OK;
    small_queue *q = queue_of(ctx);
    if (!q) {
        VMP_REQUIRE(ctx, !stream_records(ctx), VMP_ERR_INVALID,
                    "the first vmp_queue_begin of a context allocates: not while its stream records");
        q = new (std::nothrow) small_queue();
        VMP_REQUIRE(ctx, q, VMP_ERR_HIP, "out of host memory");
        memset(q, 0, sizeof(*q));
        const size_t ring = (size_t)QUEUE_SLOTS * QUEUE_CAP * sizeof(SmallOp);
        const size_t arena = (size_t)ARENA_RECORDS * sizeof(SmallOp);
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&q->host), ring, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&q->dev), ring);
        if (e == hipSuccess)
            e = hipHostMalloc(reinterpret_cast<void **>(&q->arena_host), arena, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&q->arena_dev), arena);
        const size_t pring = (size_t)QUEUE_SLOTS * PRE_CAP * sizeof(SmallPre);
        const size_t parena = (size_t)ARENA_PRE * sizeof(SmallPre);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&q->pre_host), pring, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&q->pre_dev), pring);
        if (e == hipSuccess)
            e = hipHostMalloc(reinterpret_cast<void **>(&q->pre_arena_host), parena, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&q->pre_arena_dev), parena);
        for (int i = 0; i < QUEUE_SLOTS && e == hipSuccess; ++i)
            e = hipEventCreateWithFlags(&q->done[i], hipEventDisableTiming);
        // (the interpreter's LDS arena is beyond the 64 KB a kernel gets without asking; per device)
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(small_ops_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, QLDS * (int)sizeof(double));

        ctx->queue = q;
        if (e != hipSuccess) {
            (void)destroy_small_queue(ctx);
            VMP_HIP_CHECK(ctx, e);
        }
    }
    q->open += 1;
    return VMP_OK;
}

int32_t vmp_queue_end(vmp_ctx *ctx)
{
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null context");
    small_queue *q = queue_of(ctx);
    if (!q || q->open <= 0) return VMP_OK;
    q->open -= 1;
    return q->open == 0 ? vmp_queue_flush(ctx) : VMP_OK;
}

int32_t vmp_queue_flush(vmp_ctx *ctx)
{
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null context");
    small_queue *q = queue_of(ctx);
    const char *cause = vmp_flush_cause ? vmp_flush_cause : "caller";
    vmp_flush_cause = nullptr;
    if (!q || q->n == 0) return VMP_OK;
    const int n = q->n;
    static const bool trace = getenv("VMP_QUEUE_TRACE") != nullptr;
    if (trace) {
        fprintf(stderr, "[vmp queue] flush of %d records for %s\n", n, cause);
        const SmallOp *h = q->host + (size_t)q->cur * QUEUE_CAP;
        for (int i = 0; i < n; ++i)
            if (h[i].kind == SMALL_EWISE)
                fprintf(stderr, "    ew  total=%lld nin=%d nops=%d ndim=%d\n", (long long)h[i].ew.total,
                        h[i].ew.nin, h[i].ew.nops, h[i].ew.ndim);
            else if (h[i].kind == SMALL_SUM)
                fprintf(stderr, "    sum nkeep=%lld nred=%lld nin=%d nk=%d nr=%d\n",
                        (long long)h[i].it.nkeep, (long long)h[i].it.nred, h[i].it.nin, h[i].it.nk,
                        h[i].it.nr);
            else
                fprintf(stderr, "    spd n=%d batch=%lld\n", h[i].spd.n, (long long)h[i].spd.batch);
    }
    q->n = 0;
    SmallOp *host = q->host + (size_t)q->cur * QUEUE_CAP, *dev = q->dev + (size_t)q->cur * QUEUE_CAP;
    SmallPre *phost = q->pre_host + (size_t)q->cur * PRE_CAP, *pdev = q->pre_dev + (size_t)q->cur * PRE_CAP;
    int npre = 0;
    if (vmp_tune_get("small_queue_lds", 1) != 0) {
        queue_place(host, n, phost, &npre, &q->cached_in, &q->global_in);
    } else {
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < MAXIN; ++k) host[i].lin[k] = -1;
            host[i].lout = -1;
            host[i].fence = 1;
        }
    }
    const bool rec = stream_records(ctx);
    if (rec && q->pre_arena_used + npre > ARENA_PRE) {
        // no room left to keep this launch's table of copied-in arrays for the life of the graph:
        // the launch reads everything from memory (correct, slower)
        npre = 0;
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < MAXIN; ++k) host[i].lin[k] = -1;
            host[i].lout = -1;
            host[i].fence = 1;
        }
    }
    if (rec) {
        // replayed with the graph: the records move into the arena (queue_slot made sure they
        // fit); vmp_queue_commit copies them to the device once, after the recording
        VMP_REQUIRE(ctx, q->arena_used + n <= ARENA_RECORDS && q->pre_arena_used + npre <= ARENA_PRE,
                    VMP_ERR_UNSUPPORTED, "arena of recorded small operations exhausted");
        memcpy(q->arena_host + q->arena_used, host, (size_t)n * sizeof(SmallOp));
        dev = q->arena_dev + q->arena_used;
        q->arena_used += n;
        memcpy(q->pre_arena_host + q->pre_arena_used, phost, (size_t)npre * sizeof(SmallPre));
        pdev = q->pre_arena_dev + q->pre_arena_used;
        q->pre_arena_used += npre;
    } else {
        VMP_HIP_CHECK(ctx, hipMemcpyAsync(dev, host, (size_t)n * sizeof(SmallOp),
                                          hipMemcpyHostToDevice, ctx->stream));
        if (npre > 0)
            VMP_HIP_CHECK(ctx, hipMemcpyAsync(pdev, phost, (size_t)npre * sizeof(SmallPre),
                                              hipMemcpyHostToDevice, ctx->stream));
    }
    long long *prof = nullptr;
    if (vmp_tune_get("small_queue_prof", 0) != 0 && !rec) {
        static long long *prof_dev = nullptr;
        if (!prof_dev) (void)hipMalloc(reinterpret_cast<void **>(&prof_dev), 8 * sizeof(long long));
        prof = prof_dev;
    }
    hipLaunchKernelGGL(small_ops_kernel, dim3(1), dim3(QNT), QLDS * sizeof(double), ctx->stream, dev, n,
                       pdev, npre, prof);
    if (prof) {
        long long h[4] = {0, 0, 0, 0};
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[vmp queue] %lld records: staging + dispatch %.0f, body %.0f, barrier %.0f cycles per record\n",
                h[3], (double)h[0] / (double)(h[3] ? h[3] : 1), (double)h[1] / (double)(h[3] ? h[3] : 1),
                (double)h[2] / (double)(h[3] ? h[3] : 1));
    }
    VMP_HIP_CHECK(ctx, hipGetLastError());
    q->launches += 1;
    q->ops += n;
    if (!rec) {
        // the slot is reused QUEUE_SLOTS flushes from now: by then its copy must have left the
        // pinned buffer (the event is waited for only if it has not fired, i.e. practically never)
        VMP_HIP_CHECK(ctx, hipEventRecord(q->done[q->cur], ctx->stream));
        q->pending[q->cur] = 1;
        q->cur = (q->cur + 1) % QUEUE_SLOTS;
        if (q->pending[q->cur]) {
            VMP_HIP_CHECK(ctx, hipEventSynchronize(q->done[q->cur]));
            q->pending[q->cur] = 0;
        }
    }
    return VMP_OK;
}

int32_t vmp_queue_commit(vmp_ctx *ctx)
{
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null context");
    small_queue *q = queue_of(ctx);
    if (!q || (q->arena_committed == q->arena_used && q->pre_arena_committed == q->pre_arena_used)) return VMP_OK;
    VMP_REQUIRE(ctx, !stream_records(ctx), VMP_ERR_INVALID,
                "vmp_queue_commit belongs after the recording, not into it");
    VMP_HIP_CHECK(ctx, hipMemcpy(q->arena_dev + q->arena_committed,
                                 q->arena_host + q->arena_committed,
                                 (size_t)(q->arena_used - q->arena_committed) * sizeof(SmallOp),
                                 hipMemcpyHostToDevice));
    q->arena_committed = q->arena_used;
    if (q->pre_arena_committed != q->pre_arena_used) {
        VMP_HIP_CHECK(ctx, hipMemcpy(q->pre_arena_dev + q->pre_arena_committed,
                                     q->pre_arena_host + q->pre_arena_committed,
                                     (size_t)(q->pre_arena_used - q->pre_arena_committed) * sizeof(SmallPre),
                                     hipMemcpyHostToDevice));
        q->pre_arena_committed = q->pre_arena_used;
    }
    return VMP_OK;
}

int32_t vmp_queue_stats(vmp_ctx *ctx, int64_t *launches, int64_t *ops)
{
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null context");
    small_queue *q = queue_of(ctx);
    if (launches) *launches = q ? q->launches : 0;
    if (ops) *ops = q ? q->ops : 0;
    return VMP_OK;
}

int32_t vmp_sum_multiply(vmp_ctx *ctx, int32_t ndim, const int64_t *shape, int32_t nin,
                         const double *const *in, const int64_t *in_strides,
                         const int64_t *out_strides, uint32_t reduce_mask, double scale,
                         double *out, void *workspace, size_t workspace_bytes)
{
    VMP_REQUIRE(ctx, ctx && shape && in && in_strides && out_strides && out, VMP_ERR_INVALID,
                "null argument");
    VMP_REQUIRE(ctx, ndim >= 0 && ndim <= MAXD && nin >= 1 && nin <= MAXIN, VMP_ERR_UNSUPPORTED,
                "sum_multiply supports <= %d dims and <= %d operands", MAXD, MAXIN);
    Iter it;
    memset(&it, 0, sizeof(it));
    it.nin = nin;
    it.nkeep = 1;
    it.nred = 1;
    for (int i = 0; i < nin; ++i) {
        VMP_REQUIRE(ctx, in[i] != nullptr, VMP_ERR_INVALID, "null operand %d", i);
        it.in[i] = in[i];
    }
    for (int d = 0; d < ndim; ++d) {
        VMP_REQUIRE(ctx, shape[d] >= 0, VMP_ERR_INVALID, "negative extent");
        if (shape[d] == 1) continue;
        if (reduce_mask & (1u << d)) {
            const int k = it.nr++;
            it.rsize[k] = shape[d];
            for (int i = 0; i < nin; ++i) it.rstride[i][k] = in_strides[i * ndim + d];
            it.nred *= shape[d];
        } else {
            const int k = it.nk++;
            it.ksize[k] = shape[d];
            for (int i = 0; i < nin; ++i) it.kstride[i][k] = in_strides[i * ndim + d];
            it.okstride[k] = out_strides[d];
            it.nkeep *= shape[d];
        }
    }
    if (it.nkeep == 0) return VMP_OK;
    // merge neighbouring axes of the same kind that are jointly contiguous (or jointly
    // broadcast) in every operand: fewer index divisions, and the dense forms below apply
    for (int k = it.nk - 2; k >= 0; --k) {
        bool ok = it.okstride[k] == it.okstride[k + 1] * it.ksize[k + 1];
        for (int i = 0; i < nin && ok; ++i)
            ok = it.kstride[i][k] == it.kstride[i][k + 1] * it.ksize[k + 1];
        if (!ok) continue;
        it.ksize[k] *= it.ksize[k + 1];
        it.okstride[k] = it.okstride[k + 1];
        for (int i = 0; i < nin; ++i) it.kstride[i][k] = it.kstride[i][k + 1];
        for (int j = k + 1; j < it.nk - 1; ++j) {
            it.ksize[j] = it.ksize[j + 1];
            it.okstride[j] = it.okstride[j + 1];
            for (int i = 0; i < nin; ++i) it.kstride[i][j] = it.kstride[i][j + 1];
        }
        --it.nk;
    }
    for (int k = it.nr - 2; k >= 0; --k) {
        bool ok = true;
        for (int i = 0; i < nin && ok; ++i)
            ok = it.rstride[i][k] == it.rstride[i][k + 1] * it.rsize[k + 1];
        if (!ok) continue;
        it.rsize[k] *= it.rsize[k + 1];
        for (int i = 0; i < nin; ++i) it.rstride[i][k] = it.rstride[i][k + 1];
        for (int j = k + 1; j < it.nr - 1; ++j) {
            it.rsize[j] = it.rsize[j + 1];
            for (int i = 0; i < nin; ++i) it.rstride[i][j] = it.rstride[i][j + 1];
        }
        --it.nr;
    }
    it.i32 = (it.nkeep < ((int64_t)1 << 31) && it.nred < ((int64_t)1 << 31)) ? 1 : 0;
    if (it.nkeep <= SMALL_SM_KEEP
        && it.nkeep * (it.nred > 0 ? it.nred : 1) <= vmp_tune_get("small_queue_sm_work", SMALL_SM_WORK)
        && vmp_tune_get("small_queue_sm", 1)) {
        SmallOp *slot = nullptr;
        const int32_t rc = queue_slot(ctx, &slot);
        if (rc != VMP_OK) return rc;
        if (slot) {
            slot->kind = SMALL_SUM;
            slot->scale = scale;
            slot->out = out;
            slot->it = it;
            queue_of(ctx)->n += 1;
            return VMP_OK;
        }
    }
    VMP_FLUSH_SMALL(ctx);
    hipStream_t s = ctx->stream;
    // ---- dense two-axis forms -------------------------------------------------------------
    if (it.nk == 1 && it.nr == 1 && it.nred > 0) {
        auto pow2 = [](int64_t v) { return v >= 2 && (v & (v - 1)) == 0; };
        // column sums: reduce the outer axis
        const int64_t kin = it.ksize[0], rin = it.rsize[0];
        for (int mode = 0; mode < 2; ++mode) {
            const int64_t inner = mode == 0 ? kin : rin, outer = mode == 0 ? rin : kin;
            if (!pow2(inner) || inner > (mode == 0 ? 2 * NT : 128)) continue;
            if (mode == 0 ? (outer < 4096 || !workspace) : (outer < 4096)) continue;
            Dense2D a;
            memset(&a, 0, sizeof(a));
            a.nin = nin;
            a.inner = (int)inner;
            a.outer = outer;
            bool ok = true, dense = false;
            for (int i = 0; i < nin && ok; ++i) {
                const int64_t si = mode == 0 ? it.kstride[i][0] : it.rstride[i][0];
                const int64_t so = mode == 0 ? it.rstride[i][0] : it.kstride[i][0];
                a.in[i] = in[i];
                const bool al = reinterpret_cast<uintptr_t>(in[i]) % 16 == 0;
                if (si == 1 && so == inner && al) { a.cls[i] = 0; dense = true; }
                else if (si == 0 && so == 1) a.cls[i] = 1;
                else if (si == 1 && so == 0 && al) a.cls[i] = 2;
                else if (si == 0 && so == 0) a.cls[i] = 3;
                else ok = false;
            }
            if (!ok || !dense) continue;
            const int rb = NT / (int)(inner / 2);
            int64_t nb = (outer + (int64_t)rb * D2_U - 1) / ((int64_t)rb * D2_U);
            const int64_t cap = (int64_t)ctx->num_cu * 8;
            if (nb > cap) nb = cap;
            if (mode == 0) {
                if (workspace_bytes < (size_t)(nb * inner) * sizeof(double)) continue;
                double *partial = reinterpret_cast<double *>(workspace);
                hipLaunchKernelGGL(sum_multiply_colsum_kernel, dim3((unsigned)nb), dim3(NT), 0, s,
                                   a, partial);
                launch_finish(ctx, it, (int)nb, scale, partial, out);
            } else {
                hipLaunchKernelGGL(sum_multiply_rowsum_kernel, dim3((unsigned)nb), dim3(NT), 0, s,
                                   a, scale, it.okstride[0], out);
            }
            VMP_HIP_CHECK(ctx, hipGetLastError());
            return VMP_OK;
        }
    }
    if (it.nred == 0) {
        // empty sum -> zeros
        it.nred = 0;
    }
    // long reductions: one workgroup per kept element (and slice), lanes along the reduced
    // axes -- also when there are many kept elements (e.g. per-sequence sums over T x D x D),
    // where a thread per output would read with a stride of the whole reduced extent
    const int64_t ws_doubles = workspace ? (int64_t)(workspace_bytes / sizeof(double)) : 0;
    // (a handful of kept elements over 64 .. 1023 products -- tr(A B) of two K x K matrices -- too:
    // one thread walking them alone is a chain of 256 dependent-latency loads, 0.12 ms)
    const bool use_block = (it.nred >= 1024 || (it.nred >= 64 && it.nkeep <= 256)) &&
                           (it.nkeep <= ws_doubles) && (it.nkeep <= 0x7fffffff);
    // column pattern: several kept elements along a dense innermost kept axis
    bool use_column = false;
    if (it.nred >= 512 && it.nk >= 1 && it.ksize[it.nk - 1] >= 4 && it.nkeep <= 65536) {
        use_column = true;
        for (int i = 0; i < nin; ++i) {
            const int64_t st = it.kstride[i][it.nk - 1];
            if (st != 0 && st != 1) use_column = false;
        }
    }
    // innermost reduced axis dense in some operand (then lanes along it coalesce)
    bool dense_reduce = false;
    if (it.nr >= 1)
        for (int i = 0; i < nin; ++i) dense_reduce = dense_reduce || it.rstride[i][it.nr - 1] == 1;
    // the column form needs >= 16 dense lanes to coalesce; below that the fat-thread form wins
    const bool use_fat = it.nkeep <= 64 && it.nkeep >= 2 && it.nred >= 65536 &&
                         !(use_column && it.ksize[it.nk - 1] >= 16) && workspace;
    if (use_fat) {
        int64_t nsplit = (int64_t)ctx->num_cu * 8;
        const int64_t maxsplit = (it.nred + NT - 1) / NT;
        if (nsplit > maxsplit) nsplit = maxsplit;
        VMP_REQUIRE(ctx, workspace_bytes >= (size_t)(it.nkeep * nsplit) * sizeof(double),
                    VMP_ERR_INVALID, "sum_multiply workspace too small (%lld doubles needed)",
                    (long long)(it.nkeep * nsplit));
        double *partial = reinterpret_cast<double *>(workspace);
#define VMP_FAT(nk)                                                                             \
    hipLaunchKernelGGL(sum_multiply_fat_kernel<nk>, dim3((unsigned)nsplit), dim3(NT), 0, s, it, \
                       (int)nsplit, partial)
        if (it.nkeep <= 16)
            hipLaunchKernelGGL(sum_multiply_fatthread_kernel<16>, dim3((unsigned)nsplit), dim3(NT),
                               0, s, it, (int)nsplit, partial);
        else if (it.nkeep <= 32) VMP_FAT(32);
        else VMP_FAT(64);
#undef VMP_FAT
        launch_finish(ctx, it, (int)nsplit, scale, partial, out);
    } else if (use_column) {
        const int64_t kin = it.ksize[it.nk - 1];
        const int64_t kouter = it.nkeep / kin;
        const int KX = kin >= 64 ? 64 : (kin >= 16 ? 16 : 4);
        const int64_t kblocks = (kin + KX - 1) / KX;
        int64_t want = ((int64_t)ctx->num_cu * 8 + kblocks * kouter - 1) / (kblocks * kouter);
        int64_t maxsplit = (it.nred + 8 * (NT / KX) - 1) / (8 * (NT / KX));
        int64_t nsplit = want < maxsplit ? want : maxsplit;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 1024) nsplit = 1024;
        VMP_REQUIRE(ctx, kouter <= 65535 && workspace
                             && workspace_bytes >= (size_t)(it.nkeep * nsplit) * sizeof(double),
                    VMP_ERR_INVALID, "sum_multiply workspace too small (%lld doubles needed)",
                    (long long)(it.nkeep * nsplit));
        double *partial = reinterpret_cast<double *>(workspace);
        const dim3 grid((unsigned)kblocks, (unsigned)kouter, (unsigned)nsplit);
        if (KX == 64)
            hipLaunchKernelGGL(sum_multiply_column_kernel<64>, grid, dim3(NT), 0, s, it,
                               (int)nsplit, partial);
        else if (KX == 16)
            hipLaunchKernelGGL(sum_multiply_column_kernel<16>, grid, dim3(NT), 0, s, it,
                               (int)nsplit, partial);
        else
            hipLaunchKernelGGL(sum_multiply_column_kernel<4>, grid, dim3(NT), 0, s, it,
                               (int)nsplit, partial);
        launch_finish(ctx, it, (int)nsplit, scale, partial, out);
    } else if (!use_block && it.nred >= 8 && it.nkeep >= 4096 && dense_reduce) {
        // short dense reductions: a lane group per output (coalesced), else a thread per output
        const int G = it.nred >= 64 ? 64 : (it.nred >= 32 ? 32 : (it.nred >= 16 ? 16 : 8));
        const dim3 grid((unsigned)grid_for(ctx, it.nkeep, NT / G));
        if (G == 64)
            hipLaunchKernelGGL(sum_multiply_rowgroup_kernel<64>, grid, dim3(NT), 0, s, it, scale, out);
        else if (G == 32)
            hipLaunchKernelGGL(sum_multiply_rowgroup_kernel<32>, grid, dim3(NT), 0, s, it, scale, out);
        else if (G == 16)
            hipLaunchKernelGGL(sum_multiply_rowgroup_kernel<16>, grid, dim3(NT), 0, s, it, scale, out);
        else
            hipLaunchKernelGGL(sum_multiply_rowgroup_kernel<8>, grid, dim3(NT), 0, s, it, scale, out);
    } else if (!use_block) {
        hipLaunchKernelGGL(sum_multiply_thread_kernel, dim3((unsigned)grid_for(ctx, it.nkeep, NT)),
                           dim3(NT), 0, s, it, scale, out);
    } else {
        int64_t want = ((int64_t)ctx->num_cu * 8 + it.nkeep - 1) / it.nkeep;
        int64_t maxsplit = (it.nred + 4 * NT - 1) / (4 * NT);
        int64_t nsplit = want < maxsplit ? want : maxsplit;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 4096) nsplit = 4096;
        VMP_REQUIRE(ctx, workspace && workspace_bytes >= (size_t)(it.nkeep * nsplit) * sizeof(double),
                    VMP_ERR_INVALID, "sum_multiply workspace too small (%lld doubles needed)",
                    (long long)(it.nkeep * nsplit));
        double *partial = reinterpret_cast<double *>(workspace);
        hipLaunchKernelGGL(sum_multiply_block_kernel, dim3((unsigned)it.nkeep, (unsigned)nsplit),
                           dim3(NT), 0, s, it, (int)nsplit, partial, scale, out);
        if (nsplit > 1)
            launch_finish(ctx, it, (int)nsplit, scale, partial, out);
    }
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

size_t vmp_sum_multiply_workspace_bytes(void) { return (size_t)64 << 20; }

int32_t vmp_ewise(vmp_ctx *ctx, int32_t ndim, const int64_t *shape, int32_t nin,
                  const double *const *in, const int64_t *in_strides, int32_t nops,
                  const int32_t *ops, int32_t nconsts, const double *consts, double *out)
{
    VMP_REQUIRE(ctx, ctx && out && (ndim == 0 || shape), VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, ndim >= 0 && ndim <= MAXD && nin >= 0 && nin <= MAXIN
                         && nops >= 1 && nops <= VMP_EWISE_MAX_OPS
                         && nconsts >= 0 && nconsts <= VMP_EWISE_MAX_CONSTS,
                VMP_ERR_UNSUPPORTED, "ewise program/operand count out of range");
    EwiseArgs a;
    memset(&a, 0, sizeof(a));
    a.nin = nin;
    a.nops = nops;
    a.total = 1;
    for (int i = 0; i < nin; ++i) a.in[i] = in[i];
    // drop unit dims and merge adjacent dims that are jointly contiguous in every operand
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] == 1) continue;
        a.total *= shape[d];
        bool merged = false;
        if (a.ndim > 0) {
            const int k = a.ndim - 1;
            bool ok = true;
            for (int i = 0; i < nin; ++i)
                if (a.stride[i][k] != in_strides[i * ndim + d] * shape[d]) ok = false;
            if (ok) {
                a.shape[k] *= shape[d];
                for (int i = 0; i < nin; ++i) a.stride[i][k] = in_strides[i * ndim + d];
                merged = true;
            }
        }
        if (!merged) {
            const int k = a.ndim++;
            a.shape[k] = shape[d];
            for (int i = 0; i < nin; ++i) a.stride[i][k] = in_strides[i * ndim + d];
        }
    }
    for (int p = 0; p < nops; ++p) {
        const int op = ops[p] & 0xff, arg = ops[p] >> 8;
        VMP_REQUIRE(ctx, op >= 0 && op < VMP_OP__COUNT, VMP_ERR_INVALID, "bad opcode %d", op);
        if (op == VMP_OP_IN) VMP_REQUIRE(ctx, arg >= 0 && arg < nin, VMP_ERR_INVALID, "bad operand");
        if (op == VMP_OP_CONST)
            VMP_REQUIRE(ctx, arg >= 0 && arg < nconsts, VMP_ERR_INVALID, "bad constant index");
        a.ops[p] = ops[p];
    }
    for (int c = 0; c < nconsts; ++c) a.consts[c] = consts[c];
    if (a.total == 0) return VMP_OK;
    if (a.total <= vmp_tune_get("small_queue_ew_max", SMALL_EW_MAX) && vmp_tune_get("small_queue_ew", 1)) {
        SmallOp *slot = nullptr;
        const int32_t rc = queue_slot(ctx, &slot);
        if (rc != VMP_OK) return rc;
        if (slot) {
            slot->kind = SMALL_EWISE;
            slot->scale = 1.0;
            slot->out = out;
            slot->ew = a;
            queue_of(ctx)->n += 1;
            return VMP_OK;
        }
    }
    VMP_FLUSH_SMALL(ctx);
    const bool i32 = a.total < ((int64_t)1 << 31);
    // 16-byte accesses: innermost axis contiguous or broadcast in every operand, even extent,
    // every pair aligned (even outer strides, 16-byte aligned bases)
    bool vec = i32 && a.ndim >= 1 && a.ndim <= 4 && (a.shape[a.ndim - 1] % 2 == 0) &&
               (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    int pairmask = 0;
    for (int i = 0; i < nin && vec; ++i) {
        const int64_t st = a.stride[i][a.ndim - 1];
        if (st == 1) {
            pairmask |= 1 << i;
            if (reinterpret_cast<uintptr_t>(a.in[i]) % 16 != 0) vec = false;
            for (int d = 0; d < a.ndim - 1; ++d)
                if (a.stride[i][d] % 2 != 0) vec = false;
        } else if (st != 0) {
            vec = false;
        }
    }
    if (vec) {
        const int k = a.ndim - 1;
        a.shape[k] /= 2;
        a.total /= 2;
        for (int i = 0; i < nin; ++i)
            if ((pairmask >> i) & 1) a.stride[i][k] = 2;
        a.pairmask = pairmask;
    }
    const dim3 grid((unsigned)grid_for(ctx, a.total, NT * (vec ? 2 : 4)));
#define VMP_EW_NI(nd, V)                                                                       \
    do {                                                                                       \
        if (nin <= 2)                                                                          \
            hipLaunchKernelGGL((ewise_small_kernel<nd, true, V, 2>), grid, dim3(NT), 0,        \
                               ctx->stream, a, out);                                           \
        else if (nin <= 4)                                                                     \
            hipLaunchKernelGGL((ewise_small_kernel<nd, true, V, 4>), grid, dim3(NT), 0,        \
                               ctx->stream, a, out);                                           \
        else                                                                                   \
            hipLaunchKernelGGL((ewise_small_kernel<nd, true, V, MAXIN>), grid, dim3(NT), 0,    \
                               ctx->stream, a, out);                                           \
    } while (0)
#define VMP_EW(nd)                                                                             \
    do {                                                                                       \
        if (vec)                                                                               \
            VMP_EW_NI(nd, true);                                                               \
        else if (i32)                                                                          \
            VMP_EW_NI(nd, false);                                                              \
        else                                                                                   \
            hipLaunchKernelGGL((ewise_small_kernel<nd, false>), grid, dim3(NT), 0,             \
                               ctx->stream, a, out);                                           \
    } while (0)
    if (a.ndim <= 1) VMP_EW(1);
    else if (a.ndim == 2) VMP_EW(2);
    else if (a.ndim == 3) VMP_EW(3);
    else if (a.ndim == 4) VMP_EW(4);
    else hipLaunchKernelGGL(ewise_kernel, grid, dim3(NT), 0, ctx->stream, a, out);
#undef VMP_EW
#undef VMP_EW_NI
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_spd_batched(vmp_ctx *ctx, int32_t n, int64_t batch, const double *A, double *Ainv,
                        double *logdet, int32_t *info)
{
    VMP_REQUIRE(ctx, ctx && A, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, n >= 1 && batch >= 0, VMP_ERR_INVALID, "bad dims");
    VMP_REQUIRE(ctx, n <= SPD_MAXN, VMP_ERR_UNSUPPORTED, "batched SPD kernels support n <= %d",
                SPD_MAXN);
    if (batch == 0) return VMP_OK;
    if (n > 8 && n <= SMALL_SPD_MAXN && batch <= SMALL_SPD_BATCH && vmp_tune_get("small_queue_sm", 1)
        && vmp_tune_get("small_queue_spd", 1)) {
        // (n <= 8 has its own wavefront-per-matrix arithmetic; the queued form is the block kernel's)
        SmallOp *slot = nullptr;
        const int32_t rc = queue_slot(ctx, &slot);
        if (rc != VMP_OK) return rc;
        if (slot) {
            slot->kind = SMALL_SPD;
            slot->scale = 1.0;
            slot->out = nullptr;
            slot->spd.n = n;
            slot->spd.batch = batch;
            slot->spd.A = A;
            slot->spd.Ainv = Ainv;
            slot->spd.logdet = logdet;
            slot->spd.info = info;
            queue_of(ctx)->n += 1;
            return VMP_OK;
        }
    }
    VMP_FLUSH_SMALL(ctx);
    const int64_t big = 4 * (int64_t)ctx->num_cu;      // enough matrices to fill the chip row-wise
    if (n > 8 && n <= 16 && batch >= big)
        hipLaunchKernelGGL((spd_batched_rows_kernel<16, 256, false>),
                           dim3((unsigned)grid_for(ctx, batch, 16)), dim3(256), 0, ctx->stream, n,
                           batch, A, nullptr, Ainv, nullptr, logdet, info);
    else if (n > 16 && n <= 32 && batch >= big)
        return vmp_launch_spd_mfma(ctx, false, n, batch, A, nullptr, Ainv, nullptr, logdet, info);
    else if (n <= 8)
        hipLaunchKernelGGL(spd_batched_wave_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(NT), 0,
                           ctx->stream, n, batch, A, Ainv, logdet, info);
    else
        hipLaunchKernelGGL(spd_batched_block_kernel, dim3((unsigned)batch), dim3(NT), 0,
                           ctx->stream, n, batch, A, Ainv, logdet, info);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_gaussian_moments(vmp_ctx *ctx, int32_t n, int64_t batch, const double *phi0,
                             const double *phi1, double *u0, double *u1, double *g, int32_t *info)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && phi0 && phi1 && u0 && u1 && g, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, batch >= 0, VMP_ERR_INVALID, "bad dims");
    VMP_REQUIRE(ctx, n > 8 && n <= 32, VMP_ERR_UNSUPPORTED,
                "fused Gaussian moments are built for 8 < n <= 32 (got %d)", n);
    if (batch == 0) return VMP_OK;
    if (n <= 16)
        hipLaunchKernelGGL((spd_batched_rows_kernel<16, 256, true>),
                           dim3((unsigned)grid_for(ctx, batch, 16)), dim3(256), 0, ctx->stream, n,
                           batch, phi1, phi0, u1, u0, g, info);
    else
        return vmp_launch_spd_mfma(ctx, true, n, batch, phi1, phi0, u1, u0, g, info);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_softmax_moments(vmp_ctx *ctx, int64_t rows, int32_t K, const double *phi, double *p,
                            double *lse)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && phi && p, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, rows >= 0 && K >= 1, VMP_ERR_INVALID, "bad dims");
    if (rows == 0) return VMP_OK;
    if (K <= 256) {
        int lpr = 1;
        while (lpr * 4 < K) lpr <<= 1;
        const bool vec = (K % 2 == 0) && reinterpret_cast<uintptr_t>(phi) % 16 == 0
                         && reinterpret_cast<uintptr_t>(p) % 16 == 0;
        const dim3 grid((unsigned)grid_for(ctx, rows, NT / lpr));
#define VMP_SM(L)                                                                              \
    do {                                                                                       \
        if (vec)                                                                               \
            hipLaunchKernelGGL((softmax_rows_kernel<L, true>), grid, dim3(NT), 0, ctx->stream, \
                               rows, K, phi, p, lse);                                          \
        else                                                                                   \
            hipLaunchKernelGGL((softmax_rows_kernel<L, false>), grid, dim3(NT), 0,             \
                               ctx->stream, rows, K, phi, p, lse);                             \
    } while (0)
        switch (lpr) {
        case 1: VMP_SM(1); break;
        case 2: VMP_SM(2); break;
        case 4: VMP_SM(4); break;
        case 8: VMP_SM(8); break;
        case 16: VMP_SM(16); break;
        case 32: VMP_SM(32); break;
        default: VMP_SM(64); break;
        }
#undef VMP_SM
    } else {
        hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)grid_for(ctx, rows, 4)), dim3(NT), 0,
                           ctx->stream, rows, K, phi, p, lse);
    }
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_onehot_i64(vmp_ctx *ctx, int64_t n, int32_t K, const int64_t *labels, double *out,
                       int32_t *info)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && labels && out && info, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, n >= 0 && K >= 1, VMP_ERR_INVALID, "bad dims");
    VMP_HIP_CHECK(ctx, hipMemsetAsync(info, 0, sizeof(int32_t), ctx->stream));
    if (n == 0) return VMP_OK;
    hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)grid_for(ctx, n * K, NT * 4)), dim3(NT), 0,
                       ctx->stream, n, K, labels, out, info);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // extern "C"
