// vmp_comm.hip -- the collective of the VMP hot path: an RCCL communicator owned by the context.
//
// The reference sums every child->parent message over the plates the parent lacks
// (node.py:650 -> utils/misc.py:805) and sums each node's lower-bound term over its plates
// (expfamily.py:470-480).  With the observation plate sharded over one process per GPU those
// sums are: local partial sum (the plate-pass kernels) + vmp_allreduce_sum_f64, an fp64 sum
// all-reduce over xGMI, enqueued on the context's stream so that it is ordered with the kernels
// that produce and consume the statistics -- no host synchronisation, no Python in the path.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a process that already mapped RCCL (PyTorch
// does) shares that copy, and a single-GPU process never loads it.
#include "vmp_common.h"

#include <dlfcn.h>

namespace {

typedef int (*fn_get_unique_id)(void *);
typedef int (*fn_comm_init_rank)(void **, int, vmp_comm_id, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_error_string)(int);

struct rccl_api {
    void *handle;
    fn_get_unique_id get_unique_id;
    fn_comm_init_rank comm_init_rank;
    fn_comm_destroy comm_destroy;
    fn_all_reduce all_reduce;
    fn_error_string error_string;
};

rccl_api g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

constexpr int NCCL_FLOAT64 = 8;   // ncclFloat64 / ncclDouble (rccl.h)
constexpr int NCCL_SUM = 0;       // ncclSum

int32_t load_rccl(vmp_ctx *ctx)
{
    if (g_rccl.handle) return VMP_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    VMP_REQUIRE(ctx, h != nullptr, VMP_ERR_HIP, "cannot load RCCL (librccl.so.1): %s", dlerror());
    rccl_api a;
    a.handle = h;
    a.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    a.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    a.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    a.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    a.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    VMP_REQUIRE(ctx, a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_reduce
                && a.error_string, VMP_ERR_HIP, "librccl.so.1 lacks an expected symbol");
    g_rccl = a;
    return VMP_OK;
}

#define VMP_RCCL_CHECK(ctx, expr)                                                       \
    do {                                                                                \
        int r__ = (expr);                                                               \
        if (r__ != 0) {                                                                 \
            VMP_SET_ERR(ctx, "%s failed: %s (%s:%d)", #expr, g_rccl.error_string(r__),  \
                        __FILE__, __LINE__);                                            \
            return VMP_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

}  // namespace

extern "C" {

int32_t vmp_comm_unique_id(vmp_ctx *ctx, vmp_comm_id *id)
{
    VMP_REQUIRE(ctx, ctx && id, VMP_ERR_INVALID, "null argument");
    int32_t rc = load_rccl(ctx);
    if (rc != VMP_OK) return rc;
    VMP_RCCL_CHECK(ctx, g_rccl.get_unique_id(id));
    return VMP_OK;
}

int32_t vmp_comm_init_rank(vmp_ctx *ctx, const vmp_comm_id *id, int32_t rank, int32_t world)
{
    VMP_REQUIRE(ctx, ctx && id, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, VMP_ERR_INVALID,
                "bad rank %d of %d", rank, world);
    VMP_REQUIRE(ctx, ctx->comm == nullptr, VMP_ERR_INVALID,
                "the context already owns a communicator (vmp_comm_destroy first)");
    int32_t rc = load_rccl(ctx);
    if (rc != VMP_OK) return rc;
    VMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    void *comm = nullptr;
    VMP_RCCL_CHECK(ctx, g_rccl.comm_init_rank(&comm, world, *id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return VMP_OK;
}

int32_t vmp_comm_destroy(vmp_ctx *ctx)
{
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null argument");
    if (ctx->comm) {
        VMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        VMP_RCCL_CHECK(ctx, g_rccl.comm_destroy(ctx->comm));
        ctx->comm = nullptr;
        ctx->comm_rank = 0;
        ctx->comm_world = 1;
    }
    return VMP_OK;
}

int32_t vmp_comm_info(vmp_ctx *ctx, int32_t *rank, int32_t *world)
{
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null argument");
    if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
    if (world) *world = ctx->comm ? ctx->comm_world : 1;
    return VMP_OK;
}

int32_t vmp_allreduce_sum_f64(vmp_ctx *ctx, double *buf, int64_t count)
{
    VMP_FLUSH_SMALL(ctx);
    VMP_REQUIRE(ctx, ctx && (buf || count == 0) && count >= 0, VMP_ERR_INVALID, "bad argument");
    if (count == 0) return VMP_OK;
    // without a communicator the context is its own world: the sum over one rank
    if (!ctx->comm) return VMP_OK;
    VMP_RCCL_CHECK(ctx, g_rccl.all_reduce(buf, buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM,
                                          ctx->comm, ctx->stream));
    return VMP_OK;
}

}  // extern "C"
