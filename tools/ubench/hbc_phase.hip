// Per-phase shader-clock stamps of one workgroup of the half-band cascade kernel (fp32, 8 stages).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQH_HBC_PROBE -I quisk_amd/csrc -o tools/ubench/hbc_phase tools/ubench/hbc_phase.hip
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "qh_hbcascade.hpp"
using namespace qh;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

int main(int argc, char **argv)
{
    constexpr int NS = 8;
    using G = HbGeom<NS>;
    const int segsteps = argc > 1 ? atoi(argv[1]) : 128;
    const long long n = 1LL << 26;
    float2 *x, *y, *h;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, (n >> NS) * 8)); CK(hipMalloc(&h, G::WARM * 8));
    CK(hipMemset(x, 0, n * 8)); CK(hipMemset(h, 0, G::WARM * 8));
    if (argc > 2) {                                   // random input instead of zeros
        std::vector<float> r((size_t)n * 2);
        unsigned s = 12345u;
        for (auto &v : r) { s = s * 1664525u + 1013904223u; v = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
        CK(hipMemcpy(x, r.data(), n * 8, hipMemcpyHostToDevice));
    }
    const int seg = segsteps * G::STEP, nseg = (int)(n / seg);
    const size_t lds = (size_t)G::ring_pairs() * sizeof(HbPair<float>);
    auto k = hb45_cascade_kernel<float, NS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(nseg, 1), dim3(NT), lds, 0, x, n, h, (int)n, y, n >> NS, seg, (cplx<float> *)nullptr, G::WARM);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("segsteps %d blocks %d lds %zu: %.3f ms\n", segsteps, nseg, lds, ms);
    }
    std::vector<long long> p(64 * 16);
    CK(hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(g_hbc_probe), p.size() * 8));
    printf("step:  fillwait  fetch+S0  S1 S2 S3 S4 S5 S6 S7  carry  | total (shader clocks)\n");
    for (int r = 0; r < 8; r++) {
        long long *q = &p[(size_t)r * 16];
        printf("%2d: %6lld |", r, q[1] - q[0]);
        for (int s = 0; s < NS; s++) printf(" %5lld", q[2 + s] - q[1 + s]);
        printf(" | %5lld | next-step gap %lld\n", q[12] - q[9], p[(size_t)(r + 1) * 16] - q[12]);
    }
    return 0;
}
