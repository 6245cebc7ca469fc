// qh_kernels.hpp -- small bookkeeping kernels around qh::osfir_kernel.
#pragma once
#include "qh_osfir.hpp"

namespace qh {

// Carry a stage's input history to the next call (the reference keeps it in the resampler ring,
// wdsp/resample.c:133-134, and in fircore's fftin/fftout delay line, wdsp/firmin.c:412-429):
//   new_hist[j] <- sample at stream index  n_in - H + j   (j = 0..H-1; index -1 is the newest old sample)
// taken from `in` (mixed with the NCO when MIX, so that history is stored already rotated) or, for
// negative indices, from old_hist.  Ping-pong buffers, so reads never race the writes.
template <typename T, bool MIX, bool PACKED = false>
__global__ __launch_bounds__(NT) void hist_update_kernel(const cplx<T> *in, long long in_stride, int n_in,
                                                         const cplx<T> *old_hist, cplx<T> *new_hist, int H,
                                                         const unsigned long long *nco_phase,
                                                         const unsigned long long *nco_dphase,
                                                         const int *chan_list = nullptr,
                                                         const unsigned char *pk_src = nullptr, PackedFmt pk = PackedFmt{})
{
    using C = cplx<T>;
    const int ch = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j >= H) return;
    const long long g = (long long)n_in - H + j;
    C v;
    if (g >= 0) {
        if constexpr (PACKED) v = decode_packed<T>(pk_src, pk, ch, g);
        else v = in[(long long)ch * in_stride + g];
        if constexpr (MIX) {
            unsigned long long ph = nco_phase[ch] + nco_dphase[ch] * (unsigned long long)g;
            C rot;
            sincos_turns<T>(ph, rot.x, rot.y);
            v = cmul(v, rot);
        }
    } else {
        v = old_hist[(long long)ch * H + (H + g)];
    }
    new_hist[(long long)ch * H + j] = v;
}

// phase[ch] += dphase[ch] * n   (wdsp/shift.c:77-79 accumulates the same quantity in radians)
[[maybe_unused]] static __global__ void nco_advance_kernel(unsigned long long *phase, const unsigned long long *dphase, int nch, long long n)
{
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < nch) phase[ch] += dphase[ch] * (unsigned long long)n;
}

// xshift with run = 0 passes the samples through and leaves the phase where it is (wdsp/shift.c:60-85): park the phase
// of channel `ch` while the shift is off (phase 0, step 0 = no rotation) and bring it back when it is switched on.
[[maybe_unused]] static __global__ void nco_park_kernel(unsigned long long *phase, unsigned long long *parked, int ch, int run)
{
    if (run) phase[ch] = parked[ch];
    else { parked[ch] = phase[ch]; phase[ch] = 0; }
}

// Elementwise stage used when a chain has no FIR stage to fuse into:
//   out = epi * (in * nco)      (xshift, wdsp/shift.c:60-85; xwcpagc mode 0 + xpanel)
template <typename T, bool MIX>
__global__ __launch_bounds__(NT) void pointwise_kernel(const cplx<T> *in, long long in_stride, cplx<T> *out,
                                                       long long out_stride, int n,
                                                       const unsigned long long *nco_phase,
                                                       const unsigned long long *nco_dphase,
                                                       const EpiParam *epi, const int *chan_list = nullptr)
{
    using C = cplx<T>;
    const int ch = chan_list ? chan_list[blockIdx.y] : (int)blockIdx.y;
    EpiParam ep;
    if (epi) ep = epi[ch]; else { ep.a = 1; ep.b = 0; ep.c = 0; ep.d = 1; }
    for (long long g = (long long)blockIdx.x * NT + threadIdx.x; g < n; g += (long long)gridDim.x * NT) {
        C v = in[(long long)ch * in_stride + g];
        if constexpr (MIX) {
            unsigned long long ph = nco_phase[ch] + nco_dphase[ch] * (unsigned long long)g;
            C rot;
            sincos_turns<T>(ph, rot.x, rot.y);
            v = cmul(v, rot);
        }
        C o;
        o.x = (T)ep.a * v.x + (T)ep.b * v.y;
        o.y = (T)ep.c * v.x + (T)ep.d * v.y;
        out[(long long)ch * out_stride + g] = o;
    }
}

}  // namespace qh
