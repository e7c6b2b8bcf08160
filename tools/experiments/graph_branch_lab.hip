// Do independent branches of a HIP graph run side by side on gfx950 / ROCm 7.2?  Chains of tiny
// dependent kernels (each reads what its predecessor wrote), captured (a) on one stream,
// (b) forked over NB streams with event fork / join; replay time per graph launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(double *p, int n)
{
    const int i = threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
static float replay(hipGraphExec_t ex, hipStream_t s, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ex, s);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ex, s);
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}
int main()
{
    const int TOTAL = 128;
    double *buf; CK(hipMalloc(&buf, 64 * 4096 * sizeof(double))); CK(hipMemset(buf, 0, 64 * 4096 * sizeof(double)));
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    for (int NB : {1, 2, 4, 8, 16}) {
        std::vector<hipStream_t> st(NB);
        std::vector<hipEvent_t> ev(NB);
        for (int b = 0; b < NB; ++b) { CK(hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming)); }
        hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s0, buf, 16);          // a common head
        CK(hipEventRecord(fork, s0));
        for (int b = 0; b < NB; ++b) {
            hipStream_t s = (b == 0) ? s0 : st[b];
            if (b) CK(hipStreamWaitEvent(s, fork, 0));
            for (int k = 0; k < TOTAL / NB; ++k)
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, buf + (b + 1) * 4096, 16);
            if (b) { CK(hipEventRecord(ev[b], s)); CK(hipStreamWaitEvent(s0, ev[b], 0)); }
        }
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s0, buf, 16);          // a common tail
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        const float us = replay(ex, s0, 50);
        printf("branches %2d x %3d kernels: %8.1f us per replay (%.2f us per kernel of the longest chain)\n",
               NB, TOTAL / NB, us, us / (TOTAL / NB + 2));
        hipGraphExecDestroy(ex); hipGraphDestroy(g);
    }
    // fine-grained: a chain where every second kernel is independent (diamond pattern)
    return 0;
}
