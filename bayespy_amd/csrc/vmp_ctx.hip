// vmp_ctx.hip -- context, stream and raw-memory entry points of libvmp_hip.
#include "vmp_common.h"

#include <new>

static char g_err[512] = "no context";

// ---- measurement knobs ---------------------------------------------------------------------
namespace {
struct tune_entry { char key[32]; int value; };
tune_entry g_tune[128];
int g_ntune = 0;
bool g_tune_env = false;
}  // namespace

extern "C" char **environ;

// VMP_TUNE_<key>=<int> in the environment presets a key (read once, before the first lookup)
static void tune_from_env()
{
    g_tune_env = true;
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "VMP_TUNE_", 9) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (!eq) continue;
        const size_t len = (size_t)(eq - (*e + 9));
        if (len == 0 || len >= sizeof(g_tune[0].key) || g_ntune >= (int)(sizeof(g_tune) / sizeof(g_tune[0])))
            continue;
        memcpy(g_tune[g_ntune].key, *e + 9, len);
        g_tune[g_ntune].key[len] = 0;
        g_tune[g_ntune].value = atoi(eq + 1);
        g_ntune += 1;
    }
}

int vmp_tune_get(const char *key, int dflt)
{
    if (!g_tune_env) tune_from_env();
    for (int i = 0; i < g_ntune; ++i)
        if (strcmp(g_tune[i].key, key) == 0) return g_tune[i].value;
    return dflt;
}

extern "C" {

int32_t vmp_tune_set(const char *key, int32_t value)
{
    if (!key || strlen(key) >= sizeof(g_tune[0].key)) return VMP_ERR_INVALID;
    if (!g_tune_env) tune_from_env();
    for (int i = 0; i < g_ntune; ++i)
        if (strcmp(g_tune[i].key, key) == 0) {
            g_tune[i].value = value;
            return VMP_OK;
        }
    if (g_ntune >= (int)(sizeof(g_tune) / sizeof(g_tune[0]))) return VMP_ERR_INVALID;
    strcpy(g_tune[g_ntune].key, key);
    g_tune[g_ntune].value = value;
    g_ntune += 1;
    return VMP_OK;
}

#ifndef VMP_BUILD_ID
#define VMP_BUILD_ID "unknown"
#endif
const char *vmp_version(void) { return "libvmp_hip 0.2 (gfx950) build " VMP_BUILD_ID; }

int32_t vmp_ctx_create(int32_t device, void *stream, vmp_ctx **out)
{
    if (!out) return VMP_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device visible (%s)", hipGetErrorString(e));
        return VMP_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        snprintf(g_err, sizeof(g_err), "device %d out of range [0,%d)", device, ndev);
        return VMP_ERR_INVALID;
    }
    vmp_ctx *ctx = new (std::nothrow) vmp_ctx();
    if (!ctx) return VMP_ERR_HIP;
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    ctx->timing = 0;
    ctx->err[0] = 0;
    for (int i = 0; i < 3 * VMP_EV_RING; ++i) ctx->ev[i] = nullptr;
    ctx->ev_n = 0;
    ctx->xs = nullptr;
    ctx->ev_xfork = ctx->ev_xdone = nullptr;
    ctx->x_pending = 0;
    ctx->x_buf_pending[0] = ctx->x_buf_pending[1] = 0;
    ctx->x_count = 0;
    ctx->ev_xbuf[0] = ctx->ev_xbuf[1] = nullptr;
    ctx->xs_cus = 0;
    for (int i = 0; i < 3; ++i) ctx->ms[i] = nullptr;
    for (int i = 0; i < VMP_NME; ++i) ctx->me[i] = nullptr;
    ctx->comm = nullptr;
    ctx->queue = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    VMP_HIP_CHECK(ctx, hipSetDevice(device));
    int cu = 0;
    VMP_HIP_CHECK(ctx, hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device));
    ctx->num_cu = cu > 0 ? cu : 256;
    *out = ctx;
    return VMP_OK;
}

int32_t vmp_ctx_destroy(vmp_ctx *ctx)
{
    if (!ctx) return VMP_OK;
    (void)destroy_small_queue(ctx);
    (void)vmp_comm_destroy(ctx);
    for (int i = 0; i < 3 * VMP_EV_RING; ++i)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->xs) {
        (void)hipStreamSynchronize(ctx->xs);
        (void)hipStreamDestroy(ctx->xs);
    }
    for (int i = 0; i < 3; ++i)
        if (ctx->ms[i]) {
            (void)hipStreamSynchronize(ctx->ms[i]);
            (void)hipStreamDestroy(ctx->ms[i]);
        }
    for (int i = 0; i < VMP_NME; ++i)
        if (ctx->me[i]) (void)hipEventDestroy(ctx->me[i]);
    if (ctx->ev_xfork) (void)hipEventDestroy(ctx->ev_xfork);
    if (ctx->ev_xdone) (void)hipEventDestroy(ctx->ev_xdone);
    for (int i = 0; i < 2; ++i)
        if (ctx->ev_xbuf[i]) (void)hipEventDestroy(ctx->ev_xbuf[i]);
    delete ctx;
    return VMP_OK;
}

int32_t vmp_ctx_set_stream(vmp_ctx *ctx, void *stream)
{
    if (!ctx) return VMP_ERR_INVALID;
    if ((hipStream_t)stream != ctx->stream) VMP_FLUSH_SMALL(ctx);      // queued work belongs to the old stream
    ctx->stream = (hipStream_t)stream;
    return VMP_OK;
}

int32_t vmp_ctx_sync(vmp_ctx *ctx)
{
    if (!ctx) return VMP_ERR_INVALID;
    VMP_FLUSH_SMALL(ctx);
    {
        const int32_t rcg = vmp_pca_ensure_gram(ctx);
        if (rcg != VMP_OK) return rcg;
    }
    VMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->xs) VMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->xs));
    for (int i = 0; i < 3; ++i)
        if (ctx->ms[i]) VMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->ms[i]));
    return VMP_OK;
}

int32_t vmp_ctx_num_cu(vmp_ctx *ctx) { return ctx ? ctx->num_cu : 0; }

// ---- the launches of a call sequence as a HIP graph (a VB sweep: same kernels, same shapes every
// iteration, only the contents of the arrays move) ------------------------------------------------
struct vmp_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

int32_t vmp_graph_begin(vmp_ctx *ctx)
{
    VMP_REQUIRE(ctx, ctx, VMP_ERR_INVALID, "null context");
    VMP_REQUIRE(ctx, ctx->stream != nullptr, VMP_ERR_INVALID,
                "the legacy default stream cannot record: give the context a stream of its own");
    VMP_FLUSH_SMALL(ctx);                         // what is queued belongs in front of the recording
    VMP_HIP_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    return VMP_OK;
}

int32_t vmp_graph_end(vmp_ctx *ctx, void **graph)
{
    VMP_REQUIRE(ctx, ctx && graph, VMP_ERR_INVALID, "null argument");
    *graph = nullptr;
    const int32_t rcq = vmp_queue_flush(ctx);     // inside the recording (replayed with it)
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rcq != VMP_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rcq;
    }
    VMP_HIP_CHECK(ctx, e);
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        VMP_HIP_CHECK(ctx, e);
    }
    const int32_t rcc = vmp_queue_commit(ctx);    // device copies of the recorded small operations
    if (rcc != VMP_OK) {
        (void)hipGraphExecDestroy(x);
        (void)hipGraphDestroy(g);
        return rcc;
    }
    vmp_graph *h = new (std::nothrow) vmp_graph{g, x};
    if (!h) {
        (void)hipGraphExecDestroy(x);
        (void)hipGraphDestroy(g);
        VMP_REQUIRE(ctx, false, VMP_ERR_HIP, "out of host memory");
    }
    *graph = h;
    return VMP_OK;
}

int32_t vmp_graph_launch(vmp_ctx *ctx, void *graph)
{
    VMP_REQUIRE(ctx, ctx && graph, VMP_ERR_INVALID, "null argument");
    VMP_FLUSH_SMALL(ctx);
    VMP_HIP_CHECK(ctx, hipGraphLaunch(reinterpret_cast<vmp_graph *>(graph)->exec, ctx->stream));
    return VMP_OK;
}

int32_t vmp_graph_destroy(vmp_ctx *ctx, void *graph)
{
    (void)ctx;
    if (!graph) return VMP_OK;
    vmp_graph *h = reinterpret_cast<vmp_graph *>(graph);
    (void)hipGraphExecDestroy(h->exec);
    (void)hipGraphDestroy(h->graph);
    delete h;
    return VMP_OK;
}

const char *vmp_last_error(vmp_ctx *ctx) { return ctx ? ctx->err : g_err; }

int32_t vmp_ctx_set_timing(vmp_ctx *ctx, int32_t enabled)
{
    if (!ctx) return VMP_ERR_INVALID;
    if (enabled && !ctx->ev[0]) {
        VMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        for (int i = 0; i < 3 * VMP_EV_RING; ++i) VMP_HIP_CHECK(ctx, hipEventCreate(&ctx->ev[i]));
    }
    ctx->timing = enabled ? 1 : 0;
    ctx->ev_n = 0;
    return VMP_OK;
}

int32_t vmp_pass_times_ms(vmp_ctx *ctx, double *ms_pass, double *ms_reduce, int32_t cap,
                          int32_t *count)
{
    VMP_REQUIRE(ctx, ctx && count && cap >= 0, VMP_ERR_INVALID, "null argument");
    VMP_REQUIRE(ctx, ctx->timing && ctx->ev[0], VMP_ERR_INVALID,
                "timing not enabled (vmp_ctx_set_timing)");
    int64_t n = ctx->ev_n < VMP_EV_RING ? ctx->ev_n : VMP_EV_RING;
    if (n > cap) n = cap;
    for (int64_t i = 0; i < n; ++i) {
        hipEvent_t *e = ctx->ev + 3 * ((ctx->ev_n - n + i) % VMP_EV_RING);
        VMP_HIP_CHECK(ctx, hipEventSynchronize(e[2]));
        float a = 0.f, b = 0.f;
        VMP_HIP_CHECK(ctx, hipEventElapsedTime(&a, e[0], e[1]));
        VMP_HIP_CHECK(ctx, hipEventElapsedTime(&b, e[1], e[2]));
        if (ms_pass) ms_pass[i] = a;
        if (ms_reduce) ms_reduce[i] = b;
    }
    *count = (int32_t)n;
    ctx->ev_n = 0;
    return VMP_OK;
}

int32_t vmp_malloc(vmp_ctx *ctx, size_t bytes, void **ptr)
{
    if (!ctx || !ptr) return VMP_ERR_INVALID;
    VMP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    VMP_HIP_CHECK(ctx, hipMalloc(ptr, bytes ? bytes : 8));
    return VMP_OK;
}

int32_t vmp_free(vmp_ctx *ctx, void *ptr)
{
    if (!ctx) return VMP_ERR_INVALID;
    if (ptr) VMP_HIP_CHECK(ctx, hipFree(ptr));
    return VMP_OK;
}

int32_t vmp_memcpy_h2d(vmp_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    VMP_FLUSH_SMALL(ctx);
    if (!ctx) return VMP_ERR_INVALID;
    VMP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    VMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VMP_OK;
}

int32_t vmp_memcpy_d2h(vmp_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    VMP_FLUSH_SMALL(ctx);
    if (!ctx) return VMP_ERR_INVALID;
    VMP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VMP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VMP_OK;
}

int32_t vmp_memset_zero(vmp_ctx *ctx, void *dst, size_t bytes)
{
    VMP_FLUSH_SMALL(ctx);
    if (!ctx) return VMP_ERR_INVALID;
    VMP_HIP_CHECK(ctx, hipMemsetAsync(dst, 0, bytes, ctx->stream));
    return VMP_OK;
}

// Several small device-to-device copies as ONE launch (the copy-back of a recorded sweep: a dozen
// state arrays of a few hundred bytes each were a dozen dependent copy nodes of 4 us).
struct CopyMany {
    const double *src[24];
    double *dst[24];
    int64_t count[24];
};

__global__ void __launch_bounds__(256) copy_many_kernel(CopyMany a)
{
    const int c = blockIdx.y;
    const double *s = a.src[c];
    double *d = a.dst[c];
    const int64_t n = a.count[c];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        d[i] = s[i];
}

int32_t vmp_copy_many(vmp_ctx *ctx, int32_t n, const double *const *src, double *const *dst,
                      const int64_t *count)
{
    VMP_REQUIRE(ctx, ctx != nullptr, VMP_ERR_INVALID, "null context");
    VMP_REQUIRE(ctx, n >= 0 && (n == 0 || (src && dst && count)), VMP_ERR_INVALID, "null argument");
    VMP_FLUSH_SMALL(ctx);
    for (int32_t b = 0; b < n; b += 24) {
        CopyMany a;
        memset(&a, 0, sizeof(a));
        const int m = (n - b) < 24 ? (n - b) : 24;
        int64_t most = 0;
        for (int i = 0; i < m; ++i) {
            a.src[i] = src[b + i];
            a.dst[i] = dst[b + i];
            a.count[i] = count[b + i];
            VMP_REQUIRE(ctx, a.count[i] >= 0 && (a.count[i] == 0 || (a.src[i] && a.dst[i])),
                        VMP_ERR_INVALID, "vmp_copy_many: bad entry %d", b + i);
            if (a.count[i] > most) most = a.count[i];
        }
        int64_t gx = (most + 1023) / 1024;
        if (gx < 1) gx = 1;
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)gx, (unsigned)m), dim3(256), 0,
                           ctx->stream, a);
        VMP_HIP_CHECK(ctx, hipGetLastError());
    }
    return VMP_OK;
}

// The outputs of a recorded sweep in one vector: lower-bound terms (device scalars) and validity
// flags (any element non-zero) -- one launch instead of a reduction, a conversion and a copy node
// per entry (graph_iter.py: eight dependent nodes of 4-5 us at the end of a PCA sweep).
struct PackArgs {
    const void *src[48];
    int64_t count[48];
    int32_t kind[48];          // 0: copy one double; 1: any(int32 != 0); 2: any(double != 0)
};

__global__ void __launch_bounds__(64) pack_outputs_kernel(PackArgs a, double *__restrict__ out,
                                                           int base)
{
    const int e = blockIdx.x;
    const int kind = a.kind[e];
    if (kind == 0) {
        if (threadIdx.x == 0) out[base + e] = *reinterpret_cast<const double *>(a.src[e]);
        return;
    }
    int bad = 0;
    const int64_t n = a.count[e];
    if (kind == 1) {
        const int32_t *f = reinterpret_cast<const int32_t *>(a.src[e]);
        for (int64_t i = threadIdx.x; i < n; i += 64) bad |= (f[i] != 0);
    } else {
        const double *f = reinterpret_cast<const double *>(a.src[e]);
        for (int64_t i = threadIdx.x; i < n; i += 64) bad |= (f[i] != 0.0);   // (NaN counts too)
    }
    const unsigned long long m = __ballot(bad);
    if (threadIdx.x == 0) out[base + e] = m ? 1.0 : 0.0;
}

int32_t vmp_pack_outputs(vmp_ctx *ctx, int32_t n, const void *const *src, const int64_t *count,
                         const int32_t *kind, double *out)
{
    VMP_REQUIRE(ctx, ctx != nullptr, VMP_ERR_INVALID, "null context");
    VMP_REQUIRE(ctx, n >= 0 && (n == 0 || (src && count && kind && out)), VMP_ERR_INVALID,
                "null argument");
    VMP_FLUSH_SMALL(ctx);
    for (int32_t b = 0; b < n; b += 48) {
        PackArgs a;
        memset(&a, 0, sizeof(a));
        const int m = (n - b) < 48 ? (n - b) : 48;
        for (int i = 0; i < m; ++i) {
            a.src[i] = src[b + i];
            a.count[i] = count[b + i];
            a.kind[i] = kind[b + i];
            VMP_REQUIRE(ctx, a.src[i] && a.kind[i] >= 0 && a.kind[i] <= 2 && a.count[i] >= 0,
                        VMP_ERR_INVALID, "vmp_pack_outputs: bad entry %d", b + i);
        }
        hipLaunchKernelGGL(pack_outputs_kernel, dim3((unsigned)m), dim3(64), 0, ctx->stream, a, out,
                           (int)b);
        VMP_HIP_CHECK(ctx, hipGetLastError());
    }
    return VMP_OK;
}

}  // extern "C"
