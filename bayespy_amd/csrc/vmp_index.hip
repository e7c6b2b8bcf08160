// vmp_index.hip -- plate re-indexing kernels of the deterministic nodes that move plates:
//
//   vmp_take_axis         Take._compute_moments (take.py:72-81: np.take on a plate axis) and the
//                         block copies of Concatenate._compute_moments (concatenate.py:130-167)
//   vmp_segment_sum_axis  Take._compute_message_to_parent (take.py:83-94: misc.put_simple,
//                         utils/misc.py:549-585 -- np.add accumulation over repeated indices)
//
// Arrays are seen as (outer, axis, inner) with `inner` contiguous: lanes run along `inner`, so
// rows are moved with coalesced accesses; both kernels are HBM-bound data movement.  The
// accumulation visits the sources of every output row in a FIXED order (a CSR map built once
// on the host from the node's constant index array): results are bit-reproducible, no atomics.
#include "vmp_common.h"

namespace {

constexpr int NT = 256;

// dst[o, dst_off + j, i] = src[o, idx ? idx[j] : j, i]
__global__ __launch_bounds__(NT) void take_axis_kernel(
    int64_t outer, int64_t src_len, int64_t inner, const double *__restrict__ src, int64_t n,
    const int64_t *__restrict__ idx, double *__restrict__ dst, int64_t dst_len, int64_t dst_off)
{
    const int64_t total = outer * n * inner;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * NT) {
        const int64_t i = e % inner;
        const int64_t r = e / inner;
        const int64_t j = r % n;
        const int64_t o = r / n;
        const int64_t s = idx ? idx[j] : j;
        dst[(o * dst_len + dst_off + j) * inner + i] = src[(o * src_len + s) * inner + i];
    }
}

// dst[o, l, i] = sum_{t = ptr[l]}^{ptr[l+1]-1} src[o, perm[t], i]
__global__ __launch_bounds__(NT) void segment_sum_axis_kernel(
    int64_t outer, int64_t src_len, int64_t inner, const double *__restrict__ src,
    int64_t out_len, const int64_t *__restrict__ ptr, const int64_t *__restrict__ perm,
    double *__restrict__ dst)
{
    const int64_t total = outer * out_len * inner;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * NT) {
        const int64_t i = e % inner;
        const int64_t r = e / inner;
        const int64_t l = r % out_len;
        const int64_t o = r / out_len;
        double acc = 0.0;
        for (int64_t t = ptr[l]; t < ptr[l + 1]; ++t)
            acc += src[(o * src_len + perm[t]) * inner + i];
        dst[e] = acc;
    }
}

int64_t blocks_for(vmp_ctx *ctx, int64_t work_items)
{
    int64_t g = (work_items + NT - 1) / NT;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    return g > cap ? cap : (g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

int32_t vmp_take_axis(vmp_ctx *ctx, int64_t outer, int64_t src_len, int64_t inner,
                      const double *src, int64_t n, const int64_t *idx, double *dst,
                      int64_t dst_len, int64_t dst_off)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && src && dst, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, outer >= 0 && src_len >= 0 && inner >= 0 && n >= 0 && dst_off >= 0 &&
                     dst_off + n <= dst_len && (idx || n <= src_len),
                VMP_ERR_INVALID, "bad dims");
    if (outer * n * inner == 0) return VMP_OK;
    hipLaunchKernelGGL(take_axis_kernel, dim3((unsigned)blocks_for(ctx, outer * n * inner)),
                       dim3(NT), 0, ctx->stream, outer, src_len, inner, src, n, idx, dst, dst_len,
                       dst_off);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

int32_t vmp_segment_sum_axis(vmp_ctx *ctx, int64_t outer, int64_t src_len, int64_t inner,
                             const double *src, int64_t out_len, const int64_t *ptr,
                             const int64_t *perm, double *dst)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && src && dst && ptr && (perm || src_len == 0), VMP_ERR_INVALID,
                "null argument");
    VMP_REQUIRE(ctx, outer >= 0 && src_len >= 0 && inner >= 0 && out_len >= 0, VMP_ERR_INVALID,
                "bad dims");
    if (outer * out_len * inner == 0) return VMP_OK;
    hipLaunchKernelGGL(segment_sum_axis_kernel,
                       dim3((unsigned)blocks_for(ctx, outer * out_len * inner)), dim3(NT), 0,
                       ctx->stream, outer, src_len, inner, src, out_len, ptr, perm, dst);
    VMP_HIP_CHECK(ctx, hipGetLastError());
    return VMP_OK;
}

}  // extern "C"
