// Dependent-chain latency of the fp64 operations the sequential detector loops are made of (one wave, nothing to
// overlap with): shader clocks per operation.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/f64_latency tools/ubench/f64_latency.hip
#include <cstdio>
#include <hip/hip_runtime.h>

#define CHAIN(NAME, BODY)                                                                    \
    __global__ void NAME(double *out, long long *clk, double a, double b)                    \
    {                                                                                        \
        double x = a + threadIdx.x * 1e-9, y = b;                                            \
        long long t0 = clock64();                                                            \
        _Pragma("unroll 1") for (int i = 0; i < 256; i++) { BODY BODY BODY BODY BODY BODY BODY BODY } \
        long long t1 = clock64();                                                            \
        out[threadIdx.x] = x + y;                                                            \
        if (threadIdx.x == 0) clk[0] = t1 - t0;                                              \
    }

CHAIN(k_fma, x = __builtin_fma(x, 0.999999, y);)
CHAIN(k_add, x = x + y;)
CHAIN(k_rndne, x = rint(x) + y;)          /* rndne + add */
CHAIN(k_fract, x = __builtin_amdgcn_fract(x) + y;)
CHAIN(k_max, x = fmax(x, y) + y;)
CHAIN(k_cnd, x = (x > y ? x : y) + y;)
__device__ __forceinline__ double rl(double x)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 3);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double viaf32(double x) { float f = (float)x; f = __builtin_fmaf(f, 0.5f, 1.0f); return (double)f; }
CHAIN(k_readlane, x = rl(x) + y;)
CHAIN(k_f32fma, x = viaf32(x);)

int main()
{
    double *out; long long *clk;
    hipMalloc(&out, 64 * 8); hipMalloc(&clk, 8);
    struct { const char *n; void (*k)(double *, long long *, double, double); int ops; } t[] = {
        { "fma", k_fma, 1 }, { "add", k_add, 1 }, { "rndne+add", k_rndne, 2 }, { "fract+add", k_fract, 2 }, { "max+add", k_max, 2 },
        { "cmp+cndmask+add", k_cnd, 3 }, { "readlane x2 + add", k_readlane, 2 }, { "cvt f64->f32, fma f32, cvt back", k_f32fma, 3 } };
    for (auto &e : t) {
        long long c = 0;
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, out, clk, 1.0, 1e-3); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost); }
        printf("%-34s %7.1f clocks per chain step (%d dependent instructions)\n", e.n, (double)c / 2048.0, e.ops);
    }
    return 0;
}
