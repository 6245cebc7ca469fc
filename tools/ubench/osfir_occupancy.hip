// Micro-benchmark: does occupancy pay?  The same 256-tap complex FIR over 256 channels x 2^20 samples through
// qh::osfir_kernel with 4096-point tiles (68 KiB LDS, 2 workgroups/CU) and 2048-point tiles (34 KiB, 4/CU).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quisk_amd/csrc -o osfir_occupancy tools/ubench/osfir_occupancy.hip quisk_amd/csrc/qh_design.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "qh_osfir.hpp"
#include "qh_design.hpp"
using namespace qh;

template <int N> double run(int nch, int n, int ntaps)
{
    const int P = ntaps - 1, L = N - P;
    double2 *in, *out, *mask, *tw;
    hipMalloc(&in, (size_t)nch * n * 16); hipMalloc(&out, (size_t)nch * n * 16);
    hipMemset(in, 0, (size_t)nch * n * 16);
    std::vector<cd> h(ntaps, cd(1.0 / ntaps, 0)), m = make_mask(h, N), t = fft_twiddle_table(N);
    hipMalloc(&mask, m.size() * 16); hipMemcpy(mask, m.data(), m.size() * 16, hipMemcpyHostToDevice);
    hipMalloc(&tw, t.size() * 16); hipMemcpy(tw, t.data(), t.size() * 16, hipMemcpyHostToDevice);
    OsfirArgs<double> a{};
    a.in = in; a.in_stride = n; a.out = out; a.out_stride = n; a.mask = mask; a.mask_stride = 0; a.tw_fwd = tw; a.tw_inv = tw;
    a.n_in = n; a.n_out = n; a.P = P; a.Lout = L; a.ntiles = (n + L - 1) / L;
    const int lds = lds_elems<N>() * 16;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&osfir_kernel<double, N, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 g(a.ntiles, nch);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((osfir_kernel<double, N, 1, false>), g, dim3(NT), lds, 0, a);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((osfir_kernel<double, N, 1, false>), g, dim3(NT), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("NFFT %d: %d tiles/ch, %.3f ms, %.1f Gsamp/s, %.1f Gpoint/s of FFT work\n", N, a.ntiles, ms, (double)nch * n / ms / 1e6,
           (double)nch * a.ntiles * N / ms / 1e6);
    hipFree(in); hipFree(out); hipFree(mask); hipFree(tw);
    return ms;
}

int main()
{
    run<4096>(256, 1 << 20, 256);
    run<2048>(256, 1 << 20, 256);
    run<4096>(256, 1 << 20, 256);
    run<2048>(256, 1 << 20, 256);
    return 0;
}
