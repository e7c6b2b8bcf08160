// xpass_lab.hip -- A/B harness for the PCA plate pass (vmp_pca_xpass / vmp_pca_xpass_tiled) through
// the C ABI of libvmp_hip.so: same data, same A, every layout / occupancy / cache-policy variant
// interleaved in ONE process (cdna_hip_programming.md 5.4 rule 24), results compared bit for bit.
// Build: hipcc --offload-arch=gfx950 -O3 tools/xpass_lab.hip -Iinclude -Lbayespy_amd/csrc -lvmp_hip \
//            -Wl,-rpath,'$ORIGIN/../bayespy_amd/csrc' -o tools/xpass_lab.bin
// Run:   tools/xpass_lab.bin [N=10000000] [D=128] [K=32] [rounds=5]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;   // uniform [-1, 1)
    }
}

struct variant { const char *name; int tiled_y, tiled_x, nt, wgs, overlap, mf4; };

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 10000000;
    const int D = argc > 2 ? atoi(argv[2]) : 128;
    const int K = argc > 3 ? atoi(argv[3]) : 32;
    const int rounds = argc > 4 ? atoi(argv[4]) : 5;
    hipStream_t stream;
    CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    vmp_ctx *ctx = nullptr;
    VK(vmp_ctx_create(0, stream, &ctx));
    vmp_pca_layout L;
    VK(vmp_pca_get_layout(D, K, &L));
    const int64_t ld = (N + 31) / 32 * 32;
    int64_t yt_n = 0, xt_n = 0;
    VK(vmp_pca_tiled_doubles(D, K, N, &yt_n, &xt_n));
    size_t wsb = 0;
    VK(vmp_pca_workspace_bytes(ctx, D, K, &wsb));
    double *Y, *Yt, *X, *Xt, *Xu, *state;
    void *ws;
    CK(hipMalloc(&Y, (size_t)D * ld * 8));
    CK(hipMalloc(&Yt, (size_t)yt_n * 8));
    CK(hipMalloc(&X, (size_t)L.KP * ld * 8));
    CK(hipMalloc(&Xt, (size_t)xt_n * 8));
    CK(hipMalloc(&Xu, (size_t)L.KP * ld * 8));
    CK(hipMalloc(&state, (size_t)L.total * 8));
    CK(hipMalloc(&ws, wsb));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, stream, Y, (size_t)D * ld, 1u);
    VK(vmp_pca_init_state(ctx, D, K, 1e-2, 1e-2, 1e-2, 1e-2, state));
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, stream, state + L.off_A,
                       (size_t)(L.KP * L.DP), 7u);
    VK(vmp_pca_tile_y(ctx, Y, ld, N, D, K, Yt));
    CK(hipStreamSynchronize(stream));
    VK(vmp_ctx_set_timing(ctx, 1));

    const variant vs[] = {
        {"tile-major Y, 16x16x4 MFMA, 4 WG/CU (round 2 default)", 1, 0, 3, 4, 1, 0},
        {"row-major Y, 16x16x4 MFMA, 1 WG/CU", 0, 0, 3, 1, 1, 0},
        {"row-major Y, 16x16x4 MFMA, 2 WG/CU", 0, 0, 3, 2, 1, 0},
        {"tile-major Y, 4x4x4 MFMA, 1 WG/CU", 1, 0, 3, 1, 1, 1},
        {"tile-major Y, 4x4x4 MFMA, 2 WG/CU", 1, 0, 3, 2, 1, 1},
        {"tile-major Y, 4x4x4 MFMA, 3 WG/CU", 1, 0, 3, 3, 1, 1},
        {"row-major Y, 4x4x4 MFMA, 1 WG/CU", 0, 0, 3, 1, 1, 1},
        {"row-major Y, 4x4x4 MFMA, 2 WG/CU", 0, 0, 3, 2, 1, 1},
        {"row-major Y, 4x4x4 MFMA, 3 WG/CU", 0, 0, 3, 3, 1, 1},
    };
    const int nv = sizeof(vs) / sizeof(vs[0]);
    std::vector<double> best(nv, 1e30), sum(nv, 0.0);
    const double bytes = 8.0 * (double)N * (D + K);
    for (int r = 0; r < rounds + 1; ++r) {
        for (int v = 0; v < nv; ++v) {
            vmp_tune_set("xpass_nt", vs[v].nt);
            vmp_tune_set("xpass_wgs_per_cu", vs[v].wgs);
            vmp_tune_set("plate_stream", vs[v].overlap);
            vmp_tune_set("xpass_mfma4", vs[v].mf4);
            const int reps = 4;
            for (int i = 0; i < reps; ++i) {
                if (vs[v].tiled_y)
                    VK(vmp_pca_xpass_tiled(ctx, Yt, N, D, K, vs[v].tiled_x ? Xt : X, ld,
                                           vs[v].tiled_x, state, ws));
                else
                    VK(vmp_pca_xpass(ctx, Y, ld, N, D, K, X, ld, state, ws));
            }
            VK(vmp_pca_xjoin(ctx));
            VK(vmp_ctx_sync(ctx));
            double ms[64], red[64];
            int32_t cnt = 0;
            VK(vmp_pass_times_ms(ctx, ms, red, 64, &cnt));
            if (r == 0) continue;   // warm-up round
            for (int i = 0; i < cnt; ++i) {
                if (ms[i] < best[v]) best[v] = ms[i];
                sum[v] += ms[i] / (cnt * rounds);
            }
        }
    }
    printf("PCA plate pass N=%lld D=%d K=%d, algorithmic bytes %.3f GB per launch\n", (long long)N,
           D, K, bytes / 1e9);
    for (int v = 0; v < nv; ++v)
        printf("%-52s avg %.4f ms = %.0f GB/s (%.3f of 8 TB/s)   best %.4f ms = %.0f GB/s\n",
               vs[v].name, sum[v], bytes / sum[v] / 1e6, bytes / sum[v] / 1e6 / 8000.0, best[v],
               bytes / best[v] / 1e6);

    // ---- bit-for-bit agreement of the layouts ------------------------------------------------
    vmp_tune_set("plate_stream", 0);
    vmp_tune_set("xpass_mfma4", 0);
    CK(hipMemsetAsync(X, 0, (size_t)L.KP * ld * 8, stream));
    VK(vmp_pca_xpass(ctx, Y, ld, N, D, K, X, ld, state, ws));
    CK(hipMemsetAsync(Xu, 0, (size_t)L.KP * ld * 8, stream));
    VK(vmp_pca_xpass_tiled(ctx, Yt, N, D, K, Xu, ld, 0, state, ws));
    const size_t nchk = (size_t)K * ld;
    std::vector<double> a(nchk), b(nchk);
    CK(hipMemcpyAsync(a.data(), X, nchk * 8, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(b.data(), Xu, nchk * 8, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    size_t bad = 0;
    for (int k = 0; k < K; ++k)
        for (int64_t n = 0; n < N; ++n)
            if (memcmp(&a[(size_t)k * ld + n], &b[(size_t)k * ld + n], 8) != 0) ++bad;
    printf("row-major vs tile-major Y: %zu differing elements of %zu\n", bad, (size_t)K * N);
    VK(vmp_pca_xpass_tiled(ctx, Yt, N, D, K, Xt, ld, 1, state, ws));
    CK(hipMemsetAsync(Xu, 0, (size_t)L.KP * ld * 8, stream));
    VK(vmp_pca_tile_x(ctx, 0, Xu, ld, N, D, K, Xt));
    CK(hipMemcpyAsync(b.data(), Xu, nchk * 8, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    size_t bad2 = 0;
    for (int k = 0; k < K; ++k)
        for (int64_t n = 0; n < N; ++n)
            if (memcmp(&a[(size_t)k * ld + n], &b[(size_t)k * ld + n], 8) != 0) ++bad2;
    printf("row-major vs tile-major X (un-tiled): %zu differing elements of %zu\n", bad2,
           (size_t)K * N);
    // 4x4x4 against 16x16x4: the same sums in a different internal order (round-off only)
    vmp_tune_set("xpass_mfma4", 1);
    CK(hipMemsetAsync(Xu, 0, (size_t)L.KP * ld * 8, stream));
    VK(vmp_pca_xpass(ctx, Y, ld, N, D, K, Xu, ld, state, ws));
    CK(hipMemcpyAsync(b.data(), Xu, nchk * 8, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    double maxrel = 0.0, maxabs = 0.0;
    size_t bad3 = 0;
    for (int k = 0; k < K; ++k)
        for (int64_t n = 0; n < N; ++n) {
            const double x = a[(size_t)k * ld + n], y = b[(size_t)k * ld + n];
            if (memcmp(&x, &y, 8) != 0) ++bad3;
            const double d = x > y ? x - y : y - x;
            if (d > maxabs) maxabs = d;
        }
    printf("16x16x4 vs 4x4x4 MFMA: %zu differing elements of %zu, max |difference| %.3g\n", bad3,
           (size_t)K * N, maxabs);
    (void)maxrel;
    return (bad || bad2) ? 1 : 0;
}
