// qh_ps_kernels.hpp -- the steps of quisk_process_samples (quisk.c:2289-2742) that are not engines of their own, as kernels over
// `nch` receivers: blockIdx.y = receiver, rows `stride` samples apart.  The one-receiver block API (qh_quisk_rx_compat.cpp) runs
// them with nch = 1, the receiver bank of the whole function (qh_qps.hip) with its receiver count: one set of kernels.
#pragma once
#include <cmath>
#include <vector>
#include "qh_internal.hpp"

namespace qh_ps {

using u64 = unsigned long long;

__device__ __forceinline__ double2 turns_phasor(u64 ph)
{
    double s, c;
    sincospi(2.0 * ((double)(ph >> 11) * (1.0 / 9007199254740992.0)), &s, &c);
    return make_double2(c, s);
}

// AddTestTone (quisk.c:1258-1303) and the spectrum inversion (quisk.c:2441-2446) in one pass, out of place: the split
// receiver keeps the raw block (orig_cSamples is copied ahead of both, quisk.c:2361-2363).
//   kind 0: x += A e^{j th}        1 (AM): x += A e^{j th} (1 + cos a)        2 (FM): x += A e^{j th} e^{j cos a}        -1: no tone
// th / a: phases of testtoneVector / audioVector in 2^-64 turns, advanced per sample by dth / da (the same for every receiver:
// one add_tone, one sample clock).
static __global__ void prep_kernel(const double2 *in, long long in_stride, double2 *out, long long out_stride, int n, int kind, u64 th0, u64 dth,
                            u64 a0, u64 da, int invert)
{
    const long long row = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double2 x = in[row * in_stride + i];
        if (kind >= 0) {
            const double A = 21474836.47;                   // -40 dB, quisk.c:1263
            double2 t = turns_phasor(th0 + dth * (u64)i);
            if (kind == 1) {
                const double g = 1.0 + turns_phasor(a0 + da * (u64)i).x;
                t.x *= g; t.y *= g;
            } else if (kind == 2) {
                double s, c;
                sincos(turns_phasor(a0 + da * (u64)i).x, &s, &c);
                t = make_double2(t.x * c - t.y * s, t.x * s + t.y * c);
            }
            x.x += A * t.x; x.y += A * t.y;
        }
        if (invert) x.y = -x.y;
        out[row * out_stride + i] = x;
    }
}

// cFracDecim (quisk.c:622-665): a 4-point Lagrange interpolator stepped by fdecim input samples per output.  The reference
// carries `dindex`: +(fdecim - 1) per output, -1 per skipped input.  Unrolled, output m of a call sits at
//   w = d0 + m (fdecim - 1),  input index i_m = m + floor(w) - 1,  position within (c0..c3) = w - floor(w) + 1  in [1, 2)
// with d0 the carried dindex: one lane per output.  hist[row][3] = the last three inputs of the call before (c0, c1, c2).
// Every receiver of a bank sees the same rates and block lengths, so d0 and the output count are the bank's.
// (nout is the host's count, from the reference's running sum -- fracdecim_walk below; where this closed form puts the call's last output one
// input further on, behind the call's end, the output is the same point seen from the input before: position 2 instead of 1)
static __global__ void fracdecim_kernel(const double2 *in, long long in_stride, const double2 *hist, int nout, double d0, double step, double2 *out,
                                 long long out_stride, int na)
{
    const long long row = blockIdx.y;
    const double2 *x = in + row * in_stride, *h = hist + row * 3;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < nout; m += gridDim.x * blockDim.x) {
        const double w = fma(step, (double)m, d0);
        double fl = floor(w);
        int i = m + (int)fl - 1;
        if (i > na - 1) { fl -= 1.0; i -= 1; }
        const double d = w - fl + 1.0;
        double2 c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int j = i - 3 + k; c[k] = j >= 0 ? x[j] : h[3 + j]; }
        const double xm0 = d, xm1 = d - 1, xm2 = d - 2, xm3 = d - 3;
        const double w0 = xm1 * xm2 * xm3 / -6.0, w1 = xm0 * xm2 * xm3 / 2.0, w2 = xm0 * xm1 * xm3 / -2.0, w3 = xm0 * xm1 * xm2 / 6.0;
        out[row * out_stride + m] = make_double2(w0 * c[0].x + w1 * c[1].x + w2 * c[2].x + w3 * c[3].x,
                                                 w0 * c[0].y + w1 * c[1].y + w2 * c[2].y + w3 * c[3].y);
    }
}
// one workgroup of 64 lanes per receiver (grid = nch)
static __global__ void fd_hist_kernel(const double2 *in, long long in_stride, int n, const double2 *hist_old, double2 *hist_new)
{
    const long long row = blockIdx.x;
    const int k = threadIdx.x;                          // 3 lanes
    if (k >= 3) return;
    const int j = n - 3 + k;
    hist_new[row * 3 + k] = j >= 0 ? in[row * in_stride + j] : hist_old[row * 3 + 3 + j];
}

// kill_audio / the squelch of either output channel (quisk.c:2712-2728), then the key-up envelope (quisk.c:2729-2738):
//   keyupEnvelope += 1 / (playback_rate 5e-3) per sample until it passes 1.0 -- sample i is scaled by env0 plus (i + 1) steps, added
//   one at a time like the reference does (at most 5 ms of samples: no table, no upload; the host steps its copy of the state).
// flag_real / flag_imag: the squelch flag (an int on the device) that mutes the real / imaginary part of receiver 0, receiver r's
// `step` ints further on; or null.  flags_out[row][2] = squelch_real, squelch_imag as the reference leaves them.  In place or from
// `src` to `dst`.
static __global__ void epilogue_kernel(const double2 *src, long long src_stride, double2 *dst, long long dst_stride, int n, const int *flag_real,
                                       int fr_step, const int *flag_imag, int fi_step, int kill, double env0, double env_step, int env_n,
                                       int *flags_out)
{
    const long long row = blockIdx.y;
    int sr = flag_real ? flag_real[row * fr_step] : 0, si = flag_imag ? flag_imag[row * fi_step] : 0;
    if (kill) sr = si = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) { flags_out[2 * row] = sr; flags_out[2 * row + 1] = si; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double2 v = src[row * src_stride + i];
        if (sr) v.x = 0.0;
        if (si) v.y = 0.0;
        if (i < env_n) {
            double e = env0;
            for (int k = 0; k <= i; k++) e += env_step;
            v.x *= e; v.y *= e;
        }
        dst[row * dst_stride + i] = v;
    }
}

inline unsigned grid_x(long long n, unsigned cap = 4096u)
{
    const long long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : g > cap ? cap : g);
}

// 2^-64 turns per sample for a tone of `freq` Hz at `rate` (any sign)
inline u64 turns_step(double freq, double rate)
{
    long double t = (long double)freq / (long double)rate;
    t -= floorl(t);
    const long double sc = t * 18446744073709551616.0L;
    return sc >= 18446744073709551616.0L ? 0ull : (u64)sc;
}

// cFracDecim's loop over a call's inputs as the reference runs it (quisk.c:631-662): returns the output count and leaves the dindex the next
// call starts from.  dindex is a RUNNING SUM in double, and whether an input yields an output (`dindex < 2`) is decided on that sum: at a rate
// whose dindex comes back to whole numbers (185 185 / 3 = 61 728 -> 48 000: step 143 / 500, dindex = 2 "exactly" every 500 outputs) a closed
// form d0 + m step rounds the other way now and then, and when that input is a call's last one the output lands in the other call (seed
// 910087 of the api walk: 4512 samples against the reference's 4510, two fewer in the call after).  A few ns per input on the host.
inline int fracdecim_walk(int na, double &dindex, double fdecim)
{
    const double step = fdecim - 1;
    int nout = 0;
    for (int i = 0; i < na; i++) {
        if (dindex < 2) { nout++; dindex += step; }
        else dindex -= 1;
    }
    return nout;
}

// HalfBand7 (.. 8, 9) chained (quisk.c:2666-2677) = one polyphase interpolator: g = h * up2(h) * up4(h), gain 2 per stage
inline std::vector<double> playback_interp_taps(int ratio)
{
    double t[43];
    qh_hb45_taps(t);                                        // t[2k] = coef[k]
    std::vector<double> h45(45, 0.0);
    for (int k = 0; k < 11; k++) { h45[(size_t)(2 * k + 1)] = t[2 * k]; h45[(size_t)(43 - 2 * k)] = t[2 * k]; }
    h45[22] = 0.5;
    auto up = [](const std::vector<double> &a, int f) {
        std::vector<double> r((a.size() - 1) * (size_t)f + 1, 0.0);
        for (size_t i = 0; i < a.size(); i++) r[i * (size_t)f] = a[i];
        return r;
    };
    auto conv = [](const std::vector<double> &a, const std::vector<double> &b) {
        std::vector<double> r(a.size() + b.size() - 1, 0.0);
        for (size_t i = 0; i < a.size(); i++) for (size_t j = 0; j < b.size(); j++) r[i + j] += a[i] * b[j];
        return r;
    };
    std::vector<double> taps = h45;                         // the stage that runs at the highest rate is applied last
    if (ratio == 4) taps = conv(up(h45, 2), h45);
    if (ratio == 8) taps = conv(conv(up(h45, 4), up(h45, 2)), h45);
    return taps;
}

}  // namespace qh_ps
