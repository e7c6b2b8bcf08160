// microbench.hip -- the two roofline denominators measured on the box itself:
//   (1) fp64 MFMA issue loop (v_mfma_f64_16x16x4_f64)  -> TFLOP/s
//   (2) HBM stream read / copy                          -> GB/s
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ long long g_cycles[8];

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(double *out, int iters, double scale = 1.0)
{
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0, 0, 0, 0};
    double a = scale * threadIdx.x * 1e-3, b = scale * (1.0 + threadIdx.x * 1e-4);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) g_cycles[0] = t1 - t0;
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) fma_loop(double *out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9;
    double x0 = 1, x1 = 2, x2 = 3, x3 = 4, x4 = 5, x5 = 6, x6 = 7, x7 = 8;
    for (int it = 0; it < iters; ++it) {
        x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
        x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void __launch_bounds__(256) stream_read(const v2f64 *in, double *out, size_t n)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v2f64 v = in[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}

__global__ void __launch_bounds__(256) stream_copy(const v2f64 *in, v2f64 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = in[i];
}

// copy with U independent 16-byte loads in flight per thread; NT = nontemporal accesses
template <int U, bool NT>
__global__ void __launch_bounds__(256) stream_copy_u(const v2f64 *in, v2f64 *out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v2f64 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride);
            else out[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// the traffic mix of the PCA plate pass: read 4 units, write 1 (Y is D = 128 rows, X is K = 32)
template <bool NT>
__global__ void __launch_bounds__(256) stream_r4w1(const v2f64 *in, v2f64 *out, size_t nout)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += stride) {
        v2f64 a, b, c, d;
        if (NT) {
            a = __builtin_nontemporal_load(in + i);
            b = __builtin_nontemporal_load(in + nout + i);
            c = __builtin_nontemporal_load(in + 2 * nout + i);
            d = __builtin_nontemporal_load(in + 3 * nout + i);
            __builtin_nontemporal_store((a + b) + (c + d), out + i);
        } else {
            a = in[i]; b = in[nout + i]; c = in[2 * nout + i]; d = in[3 * nout + i];
            out[i] = (a + b) + (c + d);
        }
    }
}

// the same 4:1 mix with the READS as LDS-DMA (global_load_lds_dwordx4: 1 KB per wavefront
// instruction, HBM -> LDS without passing the registers; MI355X_MICROARCH.md quotes 6.4-6.8 TB/s
// for such streams).  A wavefront keeps DEPTH output blocks (4 KB of reads each) in flight in a
// private LDS ring; counted vmcnt waits (5 memory operations per block: 4 DMA reads + 1 store).
template <int DEPTH, bool NTL>
__global__ void __launch_bounds__(512) stream_r4w1_dma(const v2f64 *in, v2f64 *out, size_t nout)
{
    extern __shared__ __attribute__((aligned(16))) v2f64 ring[];      // [4 waves][DEPTH][4][64]
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    v2f64 *my = ring + (size_t)w * DEPTH * 256;
    const int nw = blockDim.x >> 6;
    const size_t nblk = nout / 64, stride = (size_t)gridDim.x * nw, first = (size_t)blockIdx.x * nw + w;
    auto issue = [&](size_t blk, int slot) {
        const size_t bb = blk < nblk ? blk : nblk - 1;       // beyond the end: a redundant re-load
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(in + j * nout + bb * 64 + l),
                (__attribute__((address_space(3))) void *)(my + (slot * 4 + j) * 64), 16, 0, NTL ? 2 : 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(first + d * stride, d);
    const uint32_t rd = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) v2f64 *)(my + l);
    for (size_t blk0 = first; blk0 < nblk; blk0 += DEPTH * stride) {
#pragma unroll
        for (int sl = 0; sl < DEPTH; ++sl) {
            const size_t blk = blk0 + sl * stride;
            // (DEPTH - 1) newer blocks are in flight behind this one: 5 operations each ...
            // (the first DEPTH - 1 waits of the kernel see fewer stores: they only wait longer)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH > 1 ? 5 * (DEPTH - 1) - (DEPTH - 1) : 0) : "memory");
            v2f64 a, b, c, d;
            asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"
                         "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                         : "v"(rd), "n"(sl * 4096), "n"(sl * 4096 + 1024), "n"(sl * 4096 + 2048),
                           "n"(sl * 4096 + 3072)
                         : "memory");
            if (blk < nblk) __builtin_nontemporal_store((a + b) + (c + d), out + blk * 64 + l);
            issue(blk + DEPTH * stride, sl);
        }
    }
}

// The access pattern of the PCA plate pass on tile-major data: a wavefront owns whole 32 KB tiles
// (32 load instructions of 1 KB, 8 in flight) and writes 8 KB per tile.  STAG: the wavefronts start
// at different 8 KB chunks of their tiles (do lock-stepped wavefronts 32 KB apart collide on the
// memory channels?).  SPLIT: the four wavefronts of a workgroup share ONE tile, 8 KB each.
template <int STAG, int SPLIT>
__global__ void __launch_bounds__(256) stream_tiles(const v2f64 *in, v2f64 *out, size_t ntiles)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t stride = SPLIT ? (size_t)gridDim.x : (size_t)gridDim.x * 4;
    for (size_t t = SPLIT ? (size_t)blockIdx.x : (size_t)blockIdx.x * 4 + w; t < ntiles; t += stride) {
        const v2f64 *src = in + t * 2048;                // 32 KB = 2048 16-byte units
        v2f64 acc[2] = {v2f64{0, 0}, v2f64{0, 0}};
        const int rot = STAG ? (int)((blockIdx.x * 4 + w) & 3) : 0;
        const int c0 = SPLIT ? w : 0, c1 = SPLIT ? w + 1 : 4;
        for (int c = c0; c < c1; ++c) {
            const int cc = (c + rot) & 3;
            v2f64 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(src + (cc * 8 + i) * 64 + l);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 1] += v[i];
            if (SPLIT) {
                __builtin_nontemporal_store(acc[0], out + t * 512 + c * 128 + l);
                __builtin_nontemporal_store(acc[1], out + t * 512 + c * 128 + 64 + l);
            }
        }
        if (!SPLIT) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_nontemporal_store(acc[i & 1] * (double)(i + 1), out + t * 512 + i * 64 + l);
        }
    }
}

template <typename F>
float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const bool stream_only = getenv("MB_STREAM_ONLY") != nullptr;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, CUs %d, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    double *out;
    CK(hipMalloc(&out, 256 * 4096 * sizeof(double)));
    const int iters = 20000;
    auto report = [&](const char *name, int nacc, int wgs, float ms, double scale) {
        int grid = p.multiProcessorCount * wgs;
        long long cyc = 0;
        CK(hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cycles), sizeof(cyc)));
        double flops = (double)grid * 4 /*waves*/ * iters * nacc * 2.0 * 16 * 16 * 4;
        printf("%s acc=%d WG/CU=%d data=%s: %.2f TFLOP/s, %.1f shader-cycles per mfma per SIMD, eff clock %.0f MHz\n",
               name, nacc, wgs, scale == 0.0 ? "zero" : "nonzero", flops / ms / 1e9,
               (double)cyc / ((double)iters * nacc * wgs), (double)cyc / (ms * 1e3));
    };
    if (!stream_only)
    for (double scale : {1.0, 0.0}) {
        for (int wgs = 1; wgs <= 4; wgs *= 2) {
            int grid = p.multiProcessorCount * wgs;
            float ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, out, iters, scale); }, 3);
            report("mfma_f64_16x16x4", 1, wgs, ms, scale);
            ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters, scale); }, 3);
            report("mfma_f64_16x16x4", 4, wgs, ms, scale);
            ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, iters, scale); }, 3);
            report("mfma_f64_16x16x4", 8, wgs, ms, scale);
        }
    }
    if (!stream_only) {
        int grid = p.multiProcessorCount * 8;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, dim3(grid), dim3(256), 0, 0, out, iters); }, 3);
        double flops = (double)grid * 256 * iters * 8 * 2.0;
        printf("v_fma_f64 VALU: %.2f TFLOP/s\n", flops / ms / 1e9);
    }
    size_t bytes = (size_t)4 << 30;
    v2f64 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    size_t n = bytes / sizeof(v2f64);
    for (int g = 1024; g <= 8192; g *= 2) {
        float ms = time_ms([&] { hipLaunchKernelGGL(stream_read, dim3(g), dim3(256), 0, 0, a, out, n); }, 5);
        float mc = time_ms([&] { hipLaunchKernelGGL(stream_copy, dim3(g), dim3(256), 0, 0, a, b, n); }, 5);
        printf("grid %5d: stream read %.0f GB/s, copy (r+w) %.0f GB/s\n", g, bytes / ms / 1e6, 2.0 * bytes / mc / 1e6);
    }
    for (int g = 1024; g <= 8192; g *= 2) {
        float m1 = time_ms([&] { hipLaunchKernelGGL((stream_copy_u<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); }, 5);
        float m2 = time_ms([&] { hipLaunchKernelGGL((stream_copy_u<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); }, 5);
        float m3 = time_ms([&] { hipLaunchKernelGGL((stream_copy_u<8, true>), dim3(g), dim3(256), 0, 0, a, b, n); }, 5);
        printf("grid %5d: copy (r+w) 4 loads in flight %.0f GB/s, nontemporal %.0f GB/s, 8 in flight nontemporal %.0f GB/s\n",
               g, 2.0 * bytes / m1 / 1e6, 2.0 * bytes / m2 / 1e6, 2.0 * bytes / m3 / 1e6);
    }
    {
        // 4:1 read:write stream (the plate pass of PCA at D=128, K=32): 4 GB read, 1 GB written
        const size_t nout = n / 4;
        for (int g = 256; g <= 8192; g *= 2) {
            float m1 = time_ms([&] { hipLaunchKernelGGL(stream_r4w1<false>, dim3(g), dim3(256), 0, 0, a, b, nout); }, 5);
            float m2 = time_ms([&] { hipLaunchKernelGGL(stream_r4w1<true>, dim3(g), dim3(256), 0, 0, a, b, nout); }, 5);
            printf("grid %5d: 4:1 read:write stream %.0f GB/s, nontemporal %.0f GB/s\n", g,
                   1.25 * bytes / m1 / 1e6, 1.25 * bytes / m2 / 1e6);
        }
    }
    {
        // the tile pattern of the plate pass (round 3)
        const size_t ntiles = bytes / 32768;
        for (int g : {256, 512, 768, 1024}) {
            float m0 = time_ms([&] { hipLaunchKernelGGL((stream_tiles<0, 0>), dim3(g), dim3(256), 0, 0, a, b, ntiles); }, 5);
            float m1 = time_ms([&] { hipLaunchKernelGGL((stream_tiles<1, 0>), dim3(g), dim3(256), 0, 0, a, b, ntiles); }, 5);
            float m2 = time_ms([&] { hipLaunchKernelGGL((stream_tiles<0, 1>), dim3(g), dim3(256), 0, 0, a, b, ntiles); }, 5);
            printf("grid %5d: 4:1 tile pattern (32 KB per wavefront): %.0f GB/s, staggered chunks %.0f GB/s, tile split over the 4 wavefronts %.0f GB/s\n",
                   g, 1.25 * bytes / m0 / 1e6, 1.25 * bytes / m1 / 1e6, 1.25 * bytes / m2 / 1e6);
        }
        // the same mix with LDS-DMA reads (round 3): workgroups per CU x wavefronts per workgroup x
        // blocks in flight per wavefront
        const size_t nout = n / 4;
        for (int g : {128, 256, 512}) {
            for (int nw : {2, 4, 8}) {
                const int nt = 64 * nw;
                float m1 = time_ms([&] { hipLaunchKernelGGL((stream_r4w1_dma<1, true>), dim3(g), dim3(nt), nw * 1 * 4096, 0, a, b, nout); }, 5);
                float m2 = time_ms([&] { hipLaunchKernelGGL((stream_r4w1_dma<2, true>), dim3(g), dim3(nt), nw * 2 * 4096, 0, a, b, nout); }, 5);
                float m3 = time_ms([&] { hipLaunchKernelGGL((stream_r4w1_dma<3, true>), dim3(g), dim3(nt), nw * 3 * 4096, 0, a, b, nout); }, 5);
                float m4 = time_ms([&] { hipLaunchKernelGGL((stream_r4w1_dma<4, true>), dim3(g), dim3(nt), nw * 4 * 4096, 0, a, b, nout); }, 5);
                float m2p = time_ms([&] { hipLaunchKernelGGL((stream_r4w1_dma<2, false>), dim3(g), dim3(nt), nw * 2 * 4096, 0, a, b, nout); }, 5);
                printf("grid %4d x %d waves: 4:1 read:write, LDS-DMA reads nt: depth 1 %.0f, 2 %.0f, 3 %.0f, 4 %.0f GB/s; depth 2 plain %.0f GB/s\n",
                       g, nw, 1.25 * bytes / m1 / 1e6, 1.25 * bytes / m2 / 1e6, 1.25 * bytes / m3 / 1e6,
                       1.25 * bytes / m4 / 1e6, 1.25 * bytes / m2p / 1e6);
            }
        }
    }
    return 0;
}
