// qh_egress.hpp -- audio egress: the narrowing Quisk's sound back ends apply to the complex-double samples a receive chain
// hands them, fused into the store of the chain's last kernel.
//
//   Int16   (short)(int)(volume * x / 65536)      sound_alsa.c:344-348, sound_pulseaudio.c:694-695, sound_wasapi.c:558-559
//   Int24   low three bytes, little endian, of (int)(volume * x / 256)                          sound_alsa.c:360-375
//   Int32   (int)(volume * x)                     sound_alsa.c:386-390, sound_wasapi.c:566-567
//   Float32 (float)(volume * x / CLIP32)          sound_pulseaudio.c:684-685, sound_portaudio.c:120-123, sound_wasapi.c:574-575
//
// x = creal / cimag of a sample at Quisk's +-2^31 scale; chains that run at WDSP's +-1.0 scale set prescale = CLIP32, the
// factor wdspFexchange0 multiplies by before the samples reach the sound code (quisk_wdsp.c:67).  (int) of a double
// truncates toward zero; values beyond the int range are undefined in C and saturate here.  A frame has num_channels
// slots, the real part goes to slot channel_I and the imaginary part to slot channel_Q (the other slots are not written).
// Every product is rounded on its own, in the reference's order (no fused multiply-add across them).
#pragma once
#include <hip/hip_runtime.h>

namespace qh {

enum { EG_NONE = 0, EG_I16 = 1, EG_I24 = 2, EG_I32 = 3, EG_F32 = 4 };

struct EgressFmt {
    int kind, nchan, ch_i, ch_q;
    double volume, prescale;
    unsigned char *out;             // [nch][stride] bytes
    long long stride;
};

__device__ __forceinline__ int egress_bytes(int kind) { return kind == EG_I16 ? 2 : kind == EG_I24 ? 3 : 4; }

__device__ __forceinline__ double egress_scale(double x, const EgressFmt &f)
{
#pragma clang fp contract(off)
    double t = x;
    if (f.prescale != 1.0) t = t * f.prescale;
    t = f.volume * t;
    return t;
}

// frame m of channel ch
__device__ __forceinline__ void egress_store(const EgressFmt &f, int ch, long long m, double re, double im)
{
#pragma clang fp contract(off)
    const double a = egress_scale(re, f), b = egress_scale(im, f);
    unsigned char *row = f.out + (long long)ch * f.stride;
    switch (f.kind) {
    case EG_I16: {
        short *p = reinterpret_cast<short *>(row) + m * f.nchan;
        p[f.ch_i] = (short)(int)(a / 65536);
        p[f.ch_q] = (short)(int)(b / 65536);
        break;
    }
    case EG_I24: {
        const int ii = (int)(a / 256), qq = (int)(b / 256);
        unsigned char *p = row + m * f.nchan * 3;
        p[f.ch_i * 3] = (unsigned char)ii; p[f.ch_i * 3 + 1] = (unsigned char)(ii >> 8); p[f.ch_i * 3 + 2] = (unsigned char)(ii >> 16);
        p[f.ch_q * 3] = (unsigned char)qq; p[f.ch_q * 3 + 1] = (unsigned char)(qq >> 8); p[f.ch_q * 3 + 2] = (unsigned char)(qq >> 16);
        break;
    }
    case EG_I32: {
        int *p = reinterpret_cast<int *>(row) + m * f.nchan;
        p[f.ch_i] = (int)a;
        p[f.ch_q] = (int)b;
        break;
    }
    default: {
        float *p = reinterpret_cast<float *>(row) + m * f.nchan;
        p[f.ch_i] = (float)(a / 2147483647.0);
        p[f.ch_q] = (float)(b / 2147483647.0);
        break;
    }
    }
}

// stand-alone: complex double [nch][src_stride] -> frames
[[maybe_unused]] static __global__ __launch_bounds__(256) void egress_pack_kernel(const double2 *src, long long src_stride, int n, EgressFmt f)
{
    const int ch = blockIdx.y;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n; g += (long long)gridDim.x * 256) {
        const double2 z = src[(long long)ch * src_stride + g];
        egress_store(f, ch, g, z.x, z.y);
    }
}

}  // namespace qh
