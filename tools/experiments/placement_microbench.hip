#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned long long stamps[8];
template <int LDSB, int NR>
__global__ void __launch_bounds__(256, 3) hog(unsigned long long ticks, double *sink)
{
    __shared__ double buf[LDSB / 8];
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t0; stamps[1] = 0; }
    buf[threadIdx.x] = (double)threadIdx.x;
    __syncthreads();
    double r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = buf[(threadIdx.x + i) & 255];
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = r[i] * 1.0000001 + r[(i + 1) % NR];
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NR; ++i) acc += r[i];
    if (acc == 12345.678) sink[0] = acc;
    if (threadIdx.x == 0) atomicMax(&stamps[1], (unsigned long long)wall_clock64());
}
template <int LDSB, int NR>
__global__ void __launch_bounds__(256) probe(double *sink)
{
    __shared__ double buf[LDSB / 8];
    if (threadIdx.x == 0) stamps[2] = wall_clock64();
    buf[threadIdx.x] = 1.0;
    __syncthreads();
    double r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = buf[(threadIdx.x + i) & 255];
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = r[i] * 1.0000001 + r[(i + 1) % NR];
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NR; ++i) acc += r[i];
    if (acc == 2.0) sink[1] = 1.0;
    if (threadIdx.x == 0) stamps[3] = wall_clock64();
}
template <int HL, int HR, int PL, int PR>
void run(int wgs_per_cu, int ncu, hipStream_t a, hipStream_t b, double *sink)
{
    hipLaunchKernelGGL((hog<HL, HR>), dim3(ncu * wgs_per_cu), dim3(256), 0, a, 1000ull, sink);
    hipLaunchKernelGGL((probe<PL, PR>), dim3(1), dim3(256), 0, b, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((hog<HL, HR>), dim3(ncu * wgs_per_cu), dim3(256), 0, a, 300000ull, sink);
    for (volatile int i = 0; i < 100000; ++i) {}
    hipLaunchKernelGGL((probe<PL, PR>), dim3(1), dim3(256), 0, b, sink);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(stamps), sizeof(h));
    hipFuncAttributes fa, fb;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(hog<HL, HR>));
    hipFuncGetAttributes(&fb, reinterpret_cast<const void *>(probe<PL, PR>));
    printf("hog %2d KB %3d regs x%d/CU | probe %2d KB %3d regs: probe started %.1f us after the hog's start (hog ran %.0f us)\n",
           HL / 1024, fa.numRegs, wgs_per_cu, PL / 1024, fb.numRegs, ((long long)h[2] - (long long)h[0]) / 100.0, (h[1] - h[0]) / 100.0);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    double *sink; hipMalloc(&sink, 64);
    const int n = p.multiProcessorCount;
    run<32768, 40, 26624, 41>(3, n, a, b, sink);
    run<32768, 40, 26624, 42>(3, n, a, b, sink);
    run<32768, 40, 26624, 43>(3, n, a, b, sink);
    run<32768, 40, 8192, 43>(3, n, a, b, sink);
    run<32768, 40, 8192, 44>(3, n, a, b, sink);
    return 0;
}
