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

// x^e, e >= 0 any per-lane value
__device__ __forceinline__ double ipow_d(double x, int e)
{
    double r = 1.0;
    while (e > 0) {
        if (e & 1) r *= x;
        x *= x;
        e >>= 1;
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

// ---- time segments of a call: the helpers of the two-pass segment kernels (qh_tiled.hpp, qh_qdemod.hpp) --------------------
static constexpr int kSegWaves = 16;                // wavefronts (time segments) per channel in the two-pass kernels
static constexpr int kSegThreads = 64 * kSegWaves;

// ---- DPP scans ------------------------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_fetch_d(double v)
{
    // lanes the control leaves without a source (shifted in from outside the row, rows outside ROWMASK) read 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// v_i = m v_{i-1} + u_i over the 64 lanes, zero carry-in: Kogge-Stone inside each row of 16 (row_shr 1, 2, 4, 8), then
// lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast15), then lane 31 into rows 2 and 3 (row_bcast31).
struct PoleScan { double m1, m2, m4, m8, pa, pb, pw; };     // pa = m^((lane & 15) + 1), pb = m^((lane & 31) + 1), pw = m^(lane + 1)
__device__ __forceinline__ PoleScan make_pole_scan(double m, int lane)
{
    PoleScan p;
    p.m1 = m; p.m2 = m * m; p.m4 = p.m2 * p.m2; p.m8 = p.m4 * p.m4;
    p.pa = lane_pow(m, (lane & 15) + 1); p.pb = lane_pow(m, (lane & 31) + 1); p.pw = lane_pow(m, lane + 1);
    return p;
}
__device__ __forceinline__ double scan_pole_dpp(double u, const PoleScan &p)
{
    u = __builtin_fma(p.m1, dpp_fetch_d<0x111, 0xf>(u), u);
    u = __builtin_fma(p.m2, dpp_fetch_d<0x112, 0xf>(u), u);
    u = __builtin_fma(p.m4, dpp_fetch_d<0x114, 0xf>(u), u);
    u = __builtin_fma(p.m8, dpp_fetch_d<0x118, 0xf>(u), u);
    u = __builtin_fma(p.pa, dpp_fetch_d<0x142, 0xa>(u), u);
    u = __builtin_fma(p.pb, dpp_fetch_d<0x143, 0xc>(u), u);
    return u;
}

// the segment of wavefront `wave`: batches of 64 samples [b0, b1) of the ceil(n / 64) in the call
__device__ __forceinline__ void seg_range(int n, int wave, int &b0, int &b1, int nseg = kSegWaves)
{
    const int nb = (n + 63) >> 6;
    b0 = (int)((long long)wave * nb / nseg);
    b1 = (int)((long long)(wave + 1) * nb / nseg);
}
// Segment kernels over SEVERAL workgroups per channel.  One workgroup = 16 wavefronts = 16 time segments fills one CU; a call
// with fewer channels than the chip has CUs (BASELINE config 4: 85 FM channels) leaves two thirds of them idle.  With G
// workgroups per channel (grid = channels x G, segment index 16 g + wave of S = 16 G) the two passes become two launches:
//   MODE 1  pass 1 only: every segment's response to its own samples from a zero state -> summary row in global memory
//   MODE 2  chains the summaries of the segments before its own (S - 1 at most, read from global memory) and runs pass 2
//   MODE 0  the one-launch form (G = 1, summaries in LDS, a barrier between the passes)
// kSegSumW doubles per summary; row (slot * S + sidx).
static constexpr int kSegSumW = 6;
static constexpr int kSegMaxGroups = 8;
// Batches of segment w = floor((w + 1) nb / S) - floor(w nb / S), walked in order without a division per segment: q = nb / S each,
// one more whenever the running remainder wraps (SegWalk w; ... w.next() for segments 0, 1, 2, ...).
struct SegWalk {
    int q, r, S, acc;
    __device__ __forceinline__ SegWalk(int n, int nseg) : q(((n + 63) >> 6) / nseg), r(((n + 63) >> 6) % nseg), S(nseg), acc(0) {}
    __device__ __forceinline__ int next()
    {
        acc += r;
        const int extra = acc >= S ? 1 : 0;
        acc -= extra ? S : 0;
        return q + extra;
    }
};
__device__ __forceinline__ int seg_samples(int n, int b0, int b1)
{
    const int lo = b0 * 64 < n ? b0 * 64 : n, hi = b1 * 64 < n ? b1 * 64 : n;
    return hi - lo;
}
__device__ __forceinline__ int seg_samples_of(int n, int w, int nseg)
{
    int b0, b1;
    seg_range(n, w, b0, b1, nseg);
    return seg_samples(n, b0, b1);
}

// Batches in flight per wavefront: a segment is walked in groups of kSegGroup batches, the next group's loads issued ahead
// of the work on the current one (the recurrences chain the batches, the loads do not).
static constexpr int kSegGroup = 8;
template <typename V>
__device__ __forceinline__ void seg_load(V (&z)[kSegGroup], int b, int b1, int n, int lane, const V *p)
{
#pragma unroll
    for (int k = 0; k < kSegGroup; k++) {
        const int i = (b + k) * 64 + lane;
        z[k] = V{};
        if (b + k < b1 && i < n) z[k] = p[i];
    }
}

__device__ __forceinline__ void seg_load_re(double (&z)[kSegGroup], int b, int b1, int n, int lane, const double2 *p)
{
#pragma unroll
    for (int k = 0; k < kSegGroup; k++) {
        const int i = (b + k) * 64 + lane;
        z[k] = 0.0;
        if (b + k < b1 && i < n) z[k] = p[i].x;
    }
}

}  // namespace qh
