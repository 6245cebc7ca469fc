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

// Lane i takes lane i - 1's value (lane 0: 0): one DPP move per half, no LDS crossbar.
__device__ __forceinline__ double wave_shr1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true);       // wave_shr:1, bound_ctrl: lane 0 reads 0
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int CTRL> __device__ __forceinline__ double row_shr_add(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);        // bound_ctrl: lanes shifted in from outside the row read 0
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, the same value in every lane: inclusive scan inside each row of 16 by DPP shifts, then the
// four row totals by v_readlane.
__device__ __forceinline__ double wave_sum_d(double v)
{
    v = row_shr_add<0x111>(v);      // row_shr:1
    v = row_shr_add<0x112>(v);      // row_shr:2
    v = row_shr_add<0x114>(v);      // row_shr:4
    v = row_shr_add<0x118>(v);      // row_shr:8
    return (lane_bcast(v, 15) + lane_bcast(v, 31)) + (lane_bcast(v, 47) + lane_bcast(v, 63));
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

// WDSP's mlog10 (wdsp/meterlog10.c:29-32,547-554): log10(2) * (exponent + log2 of the mantissa truncated to 11 bits)
__device__ __forceinline__ double mlog10_dev(double val)
{
    const unsigned long long N = (unsigned long long)__double_as_longlong(val);
    const int e = (int)((N >> 52) & 2047) - 1023;
    const int m = (int)((N >> (52 - 11)) & 2047);
    return 0.301029995663981 * ((double)e + log2(1.0 + (double)m / 2048.0));
}

}  // namespace qh
