// Dependent fp64 chains on one wavefront: cycles per step (tools/ubench; hipcc --offload-arch=gfx950 -O3 chain_latency.hip -o chain_latency)
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(body) ".rept 8\n\t" body ".endr\n\t"

template <int MODE> __global__ void k(double *out, long long *cyc, double a, double c, int iters)
{
    double g = out[threadIdx.x];
    unsigned long long m = ~0ull << 1, sv;
    long long t0 = wall_clock64();
    long long s0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0)
            asm volatile(REP8("v_mul_f64 %[g], %[g], %[a]\n\tv_add_f64 %[g], %[g], %[c]\n\t") : [g] "+v"(g) : [a] "v"(a), [c] "v"(c));
        else if (MODE == 1)
            asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m]\n\t"
                         REP8("v_mul_f64 %[g], %[g], %[a]\n\tv_add_f64 %[g], %[g], %[c]\n\ts_lshl_b64 exec, exec, 1\n\t")
                         "s_mov_b64 exec, %[sv]" : [g] "+v"(g), [sv] "=&s"(sv) : [a] "v"(a), [c] "v"(c), [m] "s"(m) : "scc");
        else if (MODE == 2)
            asm volatile(REP8("v_add_f64 %[g], %[g], %[c]\n\t") : [g] "+v"(g) : [c] "v"(c));
        else if (MODE == 3)
            asm volatile(REP8("v_fma_f64 %[g], %[g], %[a], %[c]\n\t") : [g] "+v"(g) : [a] "v"(a), [c] "v"(c));
        else if (MODE == 4)      // exec from a precomputed SGPR pair per step instead of a shift of exec itself
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         REP8("s_lshl_b64 %[m], %[m], 1\n\tv_mul_f64 %[g], %[g], %[a]\n\tv_add_f64 %[g], %[g], %[c]\n\ts_mov_b64 exec, %[m]\n\t")
                         "s_mov_b64 exec, %[sv]" : [g] "+v"(g), [sv] "=&s"(sv), [m] "+s"(m) : [a] "v"(a), [c] "v"(c) : "scc");
        if (MODE == 4) m = ~0ull << 1;
        else if (MODE == 6) {    // two independent chains interleaved
            double h = g + 1.0;
            asm volatile(REP8("v_mul_f64 %[g], %[g], %[a]\n\tv_mul_f64 %[h], %[h], %[a]\n\tv_add_f64 %[g], %[g], %[c]\n\tv_add_f64 %[h], %[h], %[c]\n\t")
                         : [g] "+v"(g), [h] "+v"(h) : [a] "v"(a), [c] "v"(c));
            g += h;
        }
    }
    long long s1 = __builtin_readcyclecounter();
    long long t1 = wall_clock64();
    out[threadIdx.x] = g + c;
    if (threadIdx.x == 0) { cyc[0] = s1 - s0; cyc[1] = t1 - t0; }
}

template <int MODE> void run(const char *name, int per)
{
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
    hipMemset(out, 0, 64 * 8);
    hipMemset(cyc, 0xff, 16);
    const int iters = 100000;
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 0.99999, 1e-3, iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) printf("  (%s)\n", hipGetErrorString(e));
    long long h[2];
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-44s  %.2f shader clocks per step (%.2f at 100 MHz wall clock)\n", name, (double)h[0] / ((double)iters * per), (double)h[1] / ((double)iters * per));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("mul + add, dependent", 8);
    run<1>("mul + add + s_lshl exec", 8);
    run<2>("add, dependent", 8);
    run<3>("fma, dependent", 8);
    run<4>("mul + add, exec from an SGPR pair", 8);
    run<6>("two chains of mul + add interleaved (per pair)", 8);
    return 0;
}
