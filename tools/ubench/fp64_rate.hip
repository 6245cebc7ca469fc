// Micro-benchmark: issue rate of fp64 VALU instructions on gfx950 (per-CU and whole-chip), 1..8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip && ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b)
{
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
                else if (OP == 1) x[i] = x[i] + b;
                else if (OP == 2) x[i] = x[i] * a;
                else if (OP == 3) { float f = (float)x[i]; f = __builtin_fmaf(f, (float)a, (float)b); x[i] = f; }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP> void run(const char *name, int blocks_per_cu)
{
    double *d;
    hipMalloc(&d, 8);
    int iters = 2000;
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 256 * iters * 128.0;      // lane-instructions
    double wave_insts_per_simd = (double)blocks_per_cu * 4 /*waves per block*/ * iters * 128.0 / 4 /*simds*/;
    printf("%-10s waves/SIMD %d: %.3f ms  %.2f Tinst/s (lane)  => %.2f cycles per wave-instr per SIMD @2.4GHz\n", name,
           blocks_per_cu, ms, insts / ms / 1e9, ms * 1e-3 * 2.4e9 / wave_insts_per_simd);
    hipFree(d);
}

int main()
{
    for (int b : {1, 2, 4, 8}) {
        run<0>("fma_f64", b);
        run<1>("add_f64", b);
        run<2>("mul_f64", b);
    }
    run<3>("cvt+fma32", 2);
    return 0;
}
