#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *p, int v) { atomicAdd(p, v); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv)
{
    int variant = argc > 1 ? atoi(argv[1]) : 0;
    int *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    hipStream_t s0, s1, s2;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t f1, j1, f2, j2;
    CK(hipEventCreateWithFlags(&f1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&f2, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k, 1, 1, 0, s0, d, 1);
    CK(hipEventRecord(f1, s0)); CK(hipStreamWaitEvent(s1, f1, 0));
    hipLaunchKernelGGL(k, 1, 1, 0, s1, d, 10);
    if (variant == 0) {         // nested fork from the forked stream
        CK(hipEventRecord(f2, s1)); CK(hipStreamWaitEvent(s2, f2, 0));
        hipLaunchKernelGGL(k, 1, 1, 0, s2, d, 100);
        CK(hipEventRecord(j2, s2));
        hipLaunchKernelGGL(k, 1, 1, 0, s1, d, 1000);
        CK(hipStreamWaitEvent(s1, j2, 0));
    }
    hipLaunchKernelGGL(k, 1, 1, 0, s1, d, 10000);
    CK(hipEventRecord(j1, s1));
    hipLaunchKernelGGL(k, 1, 1, 0, s0, d, 100000);
    CK(hipStreamWaitEvent(s0, j1, 0));
    hipGraph_t g; CK(hipStreamEndCapture(s0, &g));
    printf("captured\n"); fflush(stdout);
    hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    printf("instantiated\n"); fflush(stdout);
    CK(hipGraphLaunch(ex, s0)); CK(hipStreamSynchronize(s0));
    int h; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("variant %d sum %d\n", variant, h);
    return 0;
}
