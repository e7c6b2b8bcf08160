// xpass_place.hip -- does the placement of the arrays decide the speed of the PCA plate pass?
// bench.py measured 2.21 and 2.48 ms for the same kernel on the same box in two processes
// (profiles/r03/defer_bound_ab.txt).  Same process, same data: the tile-major pass
// (vmp_pca_xpass_tiled, default variant) with X at different byte offsets inside one allocation,
// and with the arrays in freshly made allocations separated by spacers of different sizes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/xpass_place.hip -Iinclude -Lbayespy_amd/csrc -lvmp_hip \
//            -Wl,-rpath,'$ORIGIN/../bayespy_amd/csrc' -o tools/xpass_place.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "vmp_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define VK(x) do { int32_t r = (x); if (r != VMP_OK) { printf("vmp error %d (%s) at %d\n", r, vmp_last_error(ctx), __LINE__); exit(1);} } while (0)

__global__ void fill_kernel(double *p, size_t n, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
    }
}

static vmp_ctx *ctx = nullptr;

static double time_pass(const double *Yt, int64_t N, int D, int K, double *X, int64_t ld,
                        double *state, void *ws, int reps, double *best, int x_tiled = 0)
{
    for (int i = 0; i < reps + 1; ++i)
        VK(vmp_pca_xpass_tiled(ctx, Yt, N, D, K, X, ld, x_tiled, state, ws));
    VK(vmp_pca_xjoin(ctx));
    VK(vmp_ctx_sync(ctx));
    double ms[64], red[64];
    int32_t cnt = 0;
    VK(vmp_pass_times_ms(ctx, ms, red, 64, &cnt));
    double s = 0.0;
    *best = 1e30;
    for (int i = cnt - reps; i < cnt; ++i) {
        s += ms[i] / reps;
        if (ms[i] < *best) *best = ms[i];
    }
    return s;
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 10000000;
    const int D = argc > 2 ? atoi(argv[2]) : 128;
    const int K = argc > 3 ? atoi(argv[3]) : 32;
    hipStream_t stream;
    CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    VK(vmp_ctx_create(0, stream, &ctx));
    vmp_pca_layout L;
    VK(vmp_pca_get_layout(D, K, &L));
    const int64_t ld = (N + 31) / 32 * 32;
    int64_t yt_n = 0, xt_n = 0;
    VK(vmp_pca_tiled_doubles(D, K, N, &yt_n, &xt_n));
    size_t wsb = 0;
    VK(vmp_pca_workspace_bytes(ctx, D, K, &wsb));
    double *Y, *Yt, *state;
    void *ws;
    CK(hipMalloc(&Y, (size_t)D * ld * 8));
    CK(hipMalloc(&Yt, (size_t)yt_n * 8));
    CK(hipMalloc(&state, (size_t)L.total * 8));
    CK(hipMalloc(&ws, wsb));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, stream, Y, (size_t)D * ld, 1u);
    VK(vmp_pca_init_state(ctx, D, K, 1e-2, 1e-2, 1e-2, 1e-2, state));
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, stream, state + L.off_A,
                       (size_t)(L.KP * L.DP), 7u);
    VK(vmp_pca_tile_y(ctx, Y, ld, N, D, K, Yt));
    CK(hipStreamSynchronize(stream));
    CK(hipFree(Y));
    VK(vmp_ctx_set_timing(ctx, 1));
    const double bytes = 8.0 * (double)N * (D + K);
    const size_t xbytes = (size_t)L.KP * ld * 8;
    printf("PCA plate pass N=%lld D=%d K=%d (tile-major Y), %.3f GB per launch; Yt at %p\n",
           (long long)N, D, K, bytes / 1e9, (void *)Yt);

    // ---- (a) X at byte offsets inside ONE allocation -----------------------------------------
    const size_t slack = (size_t)80 << 20;
    char *big;
    CK(hipMalloc(&big, xbytes + slack));
    const size_t offs[] = {0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 4 << 20,
                           8 << 20, 16 << 20, (16 << 20) + 65536, 32 << 20, 64 << 20};
    printf("(a) X inside one allocation at %p + offset\n", (void *)big);
    for (int r = 0; r < 2; ++r)
        for (size_t o : offs) {
            double best;
            const double avg = time_pass(Yt, N, D, K, (double *)(big + o), ld, state, ws, 4, &best);
            printf("  round %d  offset %10zu B   avg %.4f ms = %5.0f GB/s   best %.4f\n", r, o, avg,
                   bytes / avg / 1e6, best);
        }
    CK(hipFree(big));

    // ---- (b) fresh allocations behind spacers of different sizes ------------------------------
    printf("(b) X in a fresh allocation made after a spacer\n");
    const size_t spacers[] = {0, 1 << 20, 2 << 20, 37 << 20, 100 << 20, (size_t)1 << 30,
                              (size_t)3 << 30, (size_t)5 << 30, 0, 64 << 20};
    for (size_t sp : spacers) {
        char *spacer = nullptr;
        if (sp) CK(hipMalloc(&spacer, sp));
        double *X;
        CK(hipMalloc(&X, xbytes));
        double best;
        const double avg = time_pass(Yt, N, D, K, X, ld, state, ws, 4, &best);
        printf("  spacer %11zu B   X at %p   avg %.4f ms = %5.0f GB/s   best %.4f\n", sp, (void *)X,
               avg, bytes / avg / 1e6, best);
        CK(hipFree(X));
        if (spacer) CK(hipFree(spacer));
    }

    // ---- (c) Yt re-made as well ------------------------------------------------------------------
    printf("(c) Yt copied into fresh allocations behind spacers, X fresh\n");
    for (size_t sp : {(size_t)0, (size_t)3 << 20, (size_t)1 << 30, (size_t)7 << 30}) {
        char *spacer = nullptr;
        if (sp) CK(hipMalloc(&spacer, sp));
        double *Yt2, *X;
        CK(hipMalloc(&Yt2, (size_t)yt_n * 8));
        CK(hipMemcpyAsync(Yt2, Yt, (size_t)yt_n * 8, hipMemcpyDeviceToDevice, stream));
        CK(hipMalloc(&X, xbytes));
        double best;
        const double avg = time_pass(Yt2, N, D, K, X, ld, state, ws, 4, &best);
        printf("  spacer %11zu B   Yt at %p  X at %p   avg %.4f ms = %5.0f GB/s   best %.4f\n", sp,
               (void *)Yt2, (void *)X, avg, bytes / avg / 1e6, best);
        CK(hipFree(X));
        CK(hipFree(Yt2));
        if (spacer) CK(hipFree(spacer));
    }
    // ---- (d) physically contiguous allocations (hipDeviceMallocContiguous) ---------------------
    printf("(d) hipExtMallocWithFlags(hipDeviceMallocContiguous): X only, then Yt and X\n");
    for (int rep = 0; rep < 6; ++rep) {
        const size_t sp = rep == 0 ? 0 : ((size_t)(rep * 37 + 2) << 20);
        char *spacer = nullptr;
        if (sp) CK(hipMalloc(&spacer, sp));
        double *X = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **)&X, xbytes, hipDeviceMallocContiguous);
        if (e != hipSuccess) {
            printf("  contiguous X: %s\n", hipGetErrorString(e));
            (void)hipGetLastError();
            if (spacer) CK(hipFree(spacer));
            break;
        }
        double best;
        const double avg = time_pass(Yt, N, D, K, X, ld, state, ws, 4, &best);
        printf("  spacer %11zu B   contiguous X at %p   avg %.4f ms = %5.0f GB/s   best %.4f\n", sp,
               (void *)X, avg, bytes / avg / 1e6, best);
        CK(hipFree(X));
        if (spacer) CK(hipFree(spacer));
    }
    for (int rep = 0; rep < 4; ++rep) {
        const size_t sp = rep == 0 ? 0 : ((size_t)(rep * 53 + 1) << 20);
        char *spacer = nullptr;
        if (sp) CK(hipMalloc(&spacer, sp));
        double *Yt2 = nullptr, *X = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **)&Yt2, (size_t)yt_n * 8, hipDeviceMallocContiguous);
        if (e == hipSuccess) e = hipExtMallocWithFlags((void **)&X, xbytes, hipDeviceMallocContiguous);
        if (e != hipSuccess) {
            printf("  contiguous Yt / X: %s\n", hipGetErrorString(e));
            (void)hipGetLastError();
            break;
        }
        CK(hipMemcpyAsync(Yt2, Yt, (size_t)yt_n * 8, hipMemcpyDeviceToDevice, stream));
        double best;
        const double avg = time_pass(Yt2, N, D, K, X, ld, state, ws, 4, &best);
        printf("  spacer %11zu B   contiguous Yt at %p  X at %p   avg %.4f ms = %5.0f GB/s   best %.4f\n",
               sp, (void *)Yt2, (void *)X, avg, bytes / avg / 1e6, best);
        CK(hipFree(X));
        CK(hipFree(Yt2));
        if (spacer) CK(hipFree(spacer));
    }
    // ---- (f) contiguous allocations (the regular, slow case): row stride of X, tile-major X ------
    printf("(f) contiguous Yt and X: row stride ld of the row-major X, and tile-major X\n");
    {
        double *Yt2 = nullptr, *X = nullptr;
        const size_t xb = (size_t)L.KP * (ld + 70000) * 8 > (size_t)xt_n * 8
                              ? (size_t)L.KP * (ld + 70000) * 8 : (size_t)xt_n * 8;
        hipError_t e = hipExtMallocWithFlags((void **)&Yt2, (size_t)yt_n * 8, hipDeviceMallocContiguous);
        if (e == hipSuccess) e = hipExtMallocWithFlags((void **)&X, xb, hipDeviceMallocContiguous);
        if (e != hipSuccess) {
            printf("  contiguous Yt / X: %s\n", hipGetErrorString(e));
            (void)hipGetLastError();
        } else {
            CK(hipMemcpyAsync(Yt2, Yt, (size_t)yt_n * 8, hipMemcpyDeviceToDevice, stream));
            const int64_t pads[] = {0, 32, 64, 96, 128, 160, 256, 288, 512, 544, 1024, 1056, 2048, 2080,
                                    4096, 4128, 8192, 8224, 16384, 16416, 32768, 32800, 65536, 65568};
            for (int64_t pad : pads) {
                double best;
                const double avg = time_pass(Yt2, N, D, K, X, ld + pad, state, ws, 4, &best);
                printf("  ld = N + %6lld (row stride %% 1 MiB = %8lld B)   avg %.4f ms = %5.0f GB/s   best %.4f\n",
                       (long long)pad, (long long)(((ld + pad) * 8) % (1 << 20)), avg,
                       bytes / avg / 1e6, best);
            }
            double best;
            const double avg = time_pass(Yt2, N, D, K, X, ld, state, ws, 4, &best, 1);
            printf("  tile-major X                                   avg %.4f ms = %5.0f GB/s   best %.4f\n",
                   avg, bytes / avg / 1e6, best);
            CK(hipFree(X));
            CK(hipFree(Yt2));
        }
    }
    // tile-major X in default allocations
    for (int rep = 0; rep < 6; ++rep) {
        char *spacer = nullptr;
        CK(hipMalloc(&spacer, (size_t)(rep * 13 + 1) << 20));
        double *Yt2, *X;
        CK(hipMalloc(&Yt2, (size_t)yt_n * 8));
        CK(hipMalloc(&X, (size_t)xt_n * 8));
        CK(hipMemcpyAsync(Yt2, Yt, (size_t)yt_n * 8, hipMemcpyDeviceToDevice, stream));
        double best;
        const double avg = time_pass(Yt2, N, D, K, X, ld, state, ws, 4, &best, 1);
        printf("  default allocations #%d, tile-major X   avg %.4f ms = %5.0f GB/s   best %.4f\n", rep, avg,
               bytes / avg / 1e6, best);
        CK(hipFree(X));
        CK(hipFree(Yt2));
        CK(hipFree(spacer));
    }
    // ---- (g) landscape: X after spacers of 0 ... 96 GB (Yt fixed), then Yt AND X after them -------
    if (argc > 4) {
        printf("(g) X after a spacer of S GB (kept allocated while timed), Yt fixed at %p\n", (void *)Yt);
        for (int sg = 0; sg <= 96; sg += 4) {
            char *spacer = nullptr;
            if (sg) CK(hipMalloc(&spacer, (size_t)sg << 30));
            double *X;
            CK(hipMalloc(&X, xbytes));
            double best;
            const double avg = time_pass(Yt, N, D, K, X, ld, state, ws, 3, &best);
            printf("  S = %3d GB   X at %p   avg %.4f ms   best %.4f\n", sg, (void *)X, avg, best);
            CK(hipFree(X));
            if (spacer) CK(hipFree(spacer));
        }
        printf("(h) Yt and X both after a spacer of S GB\n");
        for (int sg = 0; sg <= 96; sg += 8) {
            char *spacer = nullptr;
            if (sg) CK(hipMalloc(&spacer, (size_t)sg << 30));
            double *Yt2, *X;
            CK(hipMalloc(&Yt2, (size_t)yt_n * 8));
            CK(hipMalloc(&X, xbytes));
            CK(hipMemcpyAsync(Yt2, Yt, (size_t)yt_n * 8, hipMemcpyDeviceToDevice, stream));
            double best;
            const double avg = time_pass(Yt2, N, D, K, X, ld, state, ws, 3, &best);
            printf("  S = %3d GB   Yt at %p  X at %p   avg %.4f ms   best %.4f\n", sg, (void *)Yt2,
                   (void *)X, avg, best);
            CK(hipFree(X));
            CK(hipFree(Yt2));
            if (spacer) CK(hipFree(spacer));
        }
        // X candidates allocated one after the other and ALL kept (as the plan's trial does)
        printf("(i) 24 allocations of X kept alive, in allocation order\n");
        std::vector<double *> keep;
        for (int c = 0; c < 24; ++c) {
            double *X;
            CK(hipMalloc(&X, xbytes));
            keep.push_back(X);
            double best;
            const double avg = time_pass(Yt, N, D, K, X, ld, state, ws, 3, &best);
            printf("  #%2d  X at %p   avg %.4f ms   best %.4f\n", c, (void *)X, avg, best);
        }
        for (double *x : keep) CK(hipFree(x));
        return 0;
    }
    // ---- (e) ten fresh default allocations of both arrays: the spread -----------------------------
    printf("(e) ten fresh default allocations of Yt and X\n");
    for (int rep = 0; rep < 10; ++rep) {
        char *spacer = nullptr;
        CK(hipMalloc(&spacer, (size_t)(rep * 11 + 1) << 20));
        double *Yt2, *X;
        CK(hipMalloc(&Yt2, (size_t)yt_n * 8));
        CK(hipMalloc(&X, xbytes));
        CK(hipMemcpyAsync(Yt2, Yt, (size_t)yt_n * 8, hipMemcpyDeviceToDevice, stream));
        double best;
        const double avg = time_pass(Yt2, N, D, K, X, ld, state, ws, 4, &best);
        printf("  #%d  avg %.4f ms = %5.0f GB/s   best %.4f\n", rep, avg, bytes / avg / 1e6, best);
        CK(hipFree(X));
        CK(hipFree(Yt2));
        CK(hipFree(spacer));
    }
    return 0;
}
