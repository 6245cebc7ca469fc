// qh_wave.hpp -- wavefront helpers shared by the sequential / scan kernels (qh_demod.hpp, qh_qdemod.hpp).
#pragma once
#include "qh_fft.hpp"

namespace qh {

static constexpr double kTwoPiRef = 6.2831853071795864;     // wdsp/comm.h:147
static constexpr double kPiRef = 3.1415926535897932;        // wdsp/comm.h:146

__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    return v;
}

// x^(lane+1) for lane = 0..63 (wave-wide), by repeated squaring per lane
__device__ __forceinline__ double lane_pow(double x, int e)
{
    double r = 1.0, b = x;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        if (e & (1 << k)) r *= b;
        b *= b;
    }
    return r;
}

// inclusive scan of v_i = m*v_{i-1} + u_i over the 64 lanes with zero carry-in: returns sum_j m^(i-j) u_j
// Value of lane i for every lane, i wave-uniform: two v_readlane_b32 instead of the LDS-crossbar ds_bpermute that
// __shfl compiles to (its ~100 cycles would sit in the critical path of every step of the sequential kernels).
__device__ __forceinline__ double lane_bcast(double v, int i)
{
    const int u = __builtin_amdgcn_readfirstlane(i);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), u), hi = __builtin_amdgcn_readlane(__double2hiint(v), u);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double scan_pole(double u, double m, int lane)
{
    double md = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double up = __shfl_up(u, d, 64);
        if (lane >= d) u = __builtin_fma(md, up, u);
        md *= md;
    }
    return u;
}

// buf: [nch][stride] complex, n samples per channel, in place.  One wave per listed channel.
}  // namespace qh
