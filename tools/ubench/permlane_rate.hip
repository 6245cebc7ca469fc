// Micro-benchmark: issue cost of the gfx950 cross-lane register swaps (v_permlane32_swap / v_permlane16_swap) and of a DPP move, against
// v_add_f64, at 1..4 waves per SIMD -- what an in-wavefront FFT exchange would pay per dword moved (profiles/r05_notes.md).
// hipcc --offload-arch=gfx950 -O3 -o permlane_rate permlane_rate.hip && ./permlane_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, double a)
{
    unsigned x[32];
    double d[16];
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = threadIdx.x * 2654435761u + i;
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) { auto r = __builtin_amdgcn_permlane32_swap(x[i], x[i + 16], false, false); x[i] = r[0]; x[i + 16] = r[1]; }
                else if (OP == 1) { auto r = __builtin_amdgcn_permlane16_swap(x[i], x[i + 16], false, false); x[i] = r[0]; x[i + 16] = r[1]; }
                else if (OP == 2) x[i] = __builtin_amdgcn_mov_dpp(x[i + 16], 0xB1, 0xf, 0xf, false) + 1u;     // quad_perm [1,0,3,2]
                else if (OP == 3) d[i] = d[i] + a;
                else if (OP == 4) {     // the mix an in-wave exchange stage would run: one swap per two fp64 additions
                    auto r = __builtin_amdgcn_permlane32_swap(x[i], x[i + 16], false, false); x[i] = r[0]; x[i + 16] = r[1];
                    d[i] = d[i] + a; d[(i + 5) & 15] = d[(i + 5) & 15] * a;
                }
            }
        }
    }
    unsigned s = 0;
    double sd = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 16; i++) sd += d[i];
    if (s == 12345u && sd == 1.5) out[0] = s;
}

template <int OP> void run(const char *name, int blocks_per_cu, double per_iter)
{
    unsigned *d;
    hipMalloc(&d, 8);
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 1.0000001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts_per_simd = (double)blocks_per_cu * iters * per_iter;      // one wave of each block per SIMD
    printf("%-26s waves/SIMD %d: %.3f ms => %.2f cycles per wave-instruction per SIMD @2.4 GHz\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / wave_insts_per_simd);
    hipFree(d);
}

int main()
{
    for (int b : {1, 2, 4}) {
        run<0>("v_permlane32_swap", b, 128.0);
        run<1>("v_permlane16_swap", b, 128.0);
        run<2>("v_mov_dpp + v_add_u32", b, 256.0);
        run<3>("v_add_f64", b, 128.0);
        run<4>("swap + add_f64 + mul_f64", b, 384.0);
    }
    return 0;
}
