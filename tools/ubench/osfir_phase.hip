// Shader-clock stamps of one workgroup of the C2 front kernel (osfir_kernel<double,4096,4,OUTMIX,POLY>) per phase,
// at full-chip occupancy.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQH_OSFIR_PROBE -I quisk_amd/csrc -o tools/ubench/osfir_phase tools/ubench/osfir_phase.hip quisk_amd/csrc/qh_design.cpp
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "qh_osfir.hpp"
#include "qh_design.hpp"
using namespace qh;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

template <int D, bool MIX, bool OUTMIX = false, bool POLY = false> int run(const char *name)
{
    const int nch = 256, ntiles = 128, P = (MIX || OUTMIX) ? 560 : 2047, Lout = (4096 - (P + D - 1) / D * D) / D;
    const long long n_in = (long long)ntiles * Lout * D + 8192, n_out = (long long)ntiles * Lout;
    double2 *in, *out, *mask, *twf, *twi; unsigned long long *ph, *dph; double2 *st;
    CK(hipMalloc(&in, nch * n_in * 16)); CK(hipMalloc(&out, nch * n_out * 16)); CK(hipMalloc(&mask, (size_t)nch * 4096 * 16));
    CK(hipMemset(in, 0, nch * n_in * 16)); CK(hipMemset(mask, 0, (size_t)nch * 4096 * 16));
    std::vector<cd> t1 = fft_twiddle_table(4096), t2 = fft_twiddle_table(4096 / D);
    CK(hipMalloc(&twf, t1.size() * 16)); CK(hipMalloc(&twi, t2.size() * 16));
    CK(hipMemcpy(twf, t1.data(), t1.size() * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(twi, t2.data(), t2.size() * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&ph, nch * 8)); CK(hipMalloc(&dph, nch * 8)); CK(hipMalloc(&st, nch * 16));
    CK(hipMemset(ph, 0, nch * 8)); CK(hipMemset(dph, 1, nch * 8)); CK(hipMemset(st, 0, nch * 16));
    double2 *trot, *lrot;
    CK(hipMalloc(&trot, (size_t)nch * ntiles * 16)); CK(hipMalloc(&lrot, (size_t)nch * 256 * 16));
    CK(hipMemset(trot, 0, (size_t)nch * ntiles * 16)); CK(hipMemset(lrot, 0, (size_t)nch * 256 * 16));
    OsfirArgs<double> a{};
    a.in = in + 4096; a.in_stride = n_in; a.out = out; a.out_stride = n_out; a.mask = mask; a.mask_stride = 4096 /* per-channel masks, as in the engine */; a.tw_fwd = twf; a.tw_inv = twi;
    a.nco_phase = ph; a.nco_dphase = dph; a.nco_step = st; a.n_in = (int)(n_in - 8192); a.n_out = (int)n_out; a.P = (P + D - 1) / D * D;
    a.Lout = Lout; a.ntiles = ntiles; a.tile_rot = trot; a.lane_rot = lrot;
    constexpr int lds = osfir_lds_bytes<double, 4096, D>();
    auto k = osfir_kernel<double, 4096, D, MIX, false, false, OUTMIX, false, POLY>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(ntiles * nch), dim3(NT), lds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    long long p[16];
    CK(hipMemcpyFromSymbol(p, HIP_SYMBOL(g_osfir_probe), sizeof(p)));
    printf("%s: %.3f ms for %d tiles;  one workgroup: load+NCO %lld | fwd FFT %lld | mask+fold %lld | inv FFT %lld | store %lld | total %lld clocks\n",
           name, ms, ntiles * nch, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5] - p[4], p[5] - p[0]);
    return 0;
}

int main() { return run<4, false, true, true>("front D=4 OUTMIX polyphase") || run<4, false, true, false>("front D=4 OUTMIX") || run<1, false>("band D=1"); }
