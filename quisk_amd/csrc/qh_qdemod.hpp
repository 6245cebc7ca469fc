// qh_qdemod.hpp -- Quisk-native detectors (quisk_process_demodulate, quisk.c:2002-2068): AM envelope with DC
// remover, FM phase-difference discriminator with de-emphasis.  One wave per channel, recurrences by scans.
#pragma once
#include "qh_wave.hpp"

namespace qh {

// Quisk's AM detector (quisk.c:2005-2012): di = |z|; d = di + 0.99*dc; out = d - dc; dc = d.  The DC remover is the
// linear recurrence dc_n = di_n + 0.99*dc_{n-1}: wave scan.  In place, (out, out).  One wave per channel.
static __global__ __launch_bounds__(64) void q_am_env_kernel(double2 *buf, long long stride, int n, double *dc_state)
{
    const int ch = blockIdx.x, lane = threadIdx.x;
    double2 *p = buf + (long long)ch * stride;
    double carry = dc_state[ch];
    const double pw = lane_pow(0.99, lane + 1);
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        double2 z = make_double2(0, 0);
        if (lane < cnt) z = p[base + lane];
        const double di = hypot(z.x, z.y);
        const double dc = scan_pole(di, 0.99, lane) + pw * carry;
        double prev = __shfl_up(dc, 1, 64);
        if (lane == 0) prev = carry;
        const double out = dc - prev;
        if (lane < cnt) p[base + lane] = make_double2(out, out);
        carry = lane_bcast(dc, cnt - 1);
    }
    if (lane == 0) dc_state[ch] = carry;
}

// Quisk's FM detector (quisk.c:2032-2064): di = arg(z * conj(z_prev)) * 20e5, then the one-pole de-emphasis
// y = di*a0 + x1*a1 - y1*b1.  state: {z_prev.re, z_prev.im, x1, y1}.  In place, (y, 0).
struct QFmParam { double a0, a1, b1; };
static __global__ __launch_bounds__(64) void q_fm_disc_kernel(double2 *buf, long long stride, int n, double4 *state, QFmParam q)
{
    const int ch = blockIdx.x, lane = threadIdx.x;
    double2 *p = buf + (long long)ch * stride;
    double4 st = state[ch];
    const double pole = -q.b1;
    const double pw = lane_pow(pole, lane + 1);
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        double2 z = make_double2(0, 0);
        if (lane < cnt) z = p[base + lane];
        double pr = __shfl_up(z.x, 1, 64), pi = __shfl_up(z.y, 1, 64);
        if (lane == 0) { pr = st.x; pi = st.y; }
        // cx * conj(fm_1)
        const double re = z.x * pr + z.y * pi, im = z.y * pr - z.x * pi;
        const double di = atan2(im, re) * 20e5;
        double dm1 = __shfl_up(di, 1, 64);
        if (lane == 0) dm1 = st.z;
        const double u = di * q.a0 + dm1 * q.a1;
        const double y = scan_pole(u, pole, lane) + pw * st.w;
        if (lane < cnt) p[base + lane] = make_double2(y, 0.0);
        const int last = cnt - 1;
        st.x = lane_bcast(z.x, last); st.y = lane_bcast(z.y, last);
        st.z = lane_bcast(di, last); st.w = lane_bcast(y, last);
    }
    if (lane == 0) state[ch] = st;
}


// FM squelch (quisk.c:2032-2033,2076-2085): the mean |cx| of the Rx-filtered samples over at least 2400 of them
// (evaluated once per call, like the reference) in dB re full scale; active while it is below squelch_level.
// One wave per channel; `buf` is the Rx filter's output of this call.
struct QSquelchState { double rf_sum, squelch; int rf_count, active; };
static __global__ __launch_bounds__(64) void q_fm_squelch_kernel(const double2 *buf, long long stride, int n, QSquelchState *state,
                                                             const double *level)
{
    const int ch = blockIdx.x, lane = threadIdx.x;
    const double2 *p = buf + (long long)ch * stride;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += hypot(p[i].x, p[i].y);
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane == 0) {
        QSquelchState st = state[ch];
        st.rf_sum += s; st.rf_count += n;
        if (st.rf_count >= 2400) {
            double v = st.rf_sum / st.rf_count / 2147483647.0;
            st.squelch = v > 1.E-10 ? 20 * log10(v) : -200.0;
            st.rf_sum = 0; st.rf_count = 0;
        }
        st.active = st.squelch < level[ch];
        state[ch] = st;
    }
}

// squelch_real && squelch_imag: the block goes out as zeros (quisk.c:2716-2719)
static __global__ __launch_bounds__(256) void q_mute_kernel(double2 *out, long long stride, int n, const QSquelchState *state)
{
    const int ch = blockIdx.y;
    if (!state[ch].active) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[(long long)ch * stride + i] = make_double2(0.0, 0.0);
}


// SSB squelch (ssb_squelch, quisk.c:1086-1180): the real audio at the filter rate is cut into 512-sample blocks
// (Hanning window, quisk.c:1110); per block the spectral flatness of the passband bins -- log of the arithmetic mean
// minus the mean of the logs of |X / CLIP16|^2, bins below 1e-4 skipped -- is compared with level * 0.005; a
// block above it opens the squelch for one second.  Per call: sq_open -= n, active = (sq_open == 0).  One
// workgroup per channel; a lane owns one passband bin and evaluates it by a rotation-recurrence DFT (only the
// passband bins are needed, at most 257).
struct QSsbSqState { int index, sq_open; };
struct QSsbSqParam { int samp_rate, bw1, bw2; double thresh; };
static __global__ __launch_bounds__(256) void q_ssb_squelch_kernel(const double2 *buf, long long stride, int n, QSsbSqState *state,
                                                               double *ring, QSquelchState *flag, QSsbSqParam q)
{
    __shared__ double blk[512];
    __shared__ double red[8];
    const int ch = blockIdx.x, t = threadIdx.x;
    const double2 *p = buf + (long long)ch * stride;
    double *rg = ring + (long long)ch * 512;
    QSsbSqState st = state[ch];
    int pos = 0, idx = st.index;
    const int nb = q.bw2 - q.bw1;
    while (idx + (n - pos) >= 512) {
        const int need = 512 - idx;
        for (int j = t; j < 512; j += 256) {
            const double v = j < idx ? rg[j] : p[pos + j - idx].x;
            blk[j] = v * (0.50 - 0.50 * cospi(2.0 * j / 512.0));
        }
        __syncthreads();
        double arith = 0.0, geom = 0.0;
        for (int i = q.bw1 + t; i < q.bw2; i += 256) {
            double wr, wi, cr = 1.0, ci = 0.0, xr = 0.0, xi = 0.0;
            sincospi(-2.0 * i / 512.0, &wi, &wr);
            for (int j = 0; j < 512; j++) {
                xr = __builtin_fma(blk[j], cr, xr); xi = __builtin_fma(blk[j], ci, xi);
                const double nr = cr * wr - ci * wi;
                ci = cr * wi + ci * wr; cr = nr;
            }
            xr /= 32767.0; xi /= 32767.0;           // CLIP16, quisk.h:14
            const double d = xr * xr + xi * xi;
            if (d > 1E-4) { arith += d; geom += log(d); }
        }
        for (int d2 = 32; d2 > 0; d2 >>= 1) { arith += __shfl_down(arith, d2, 64); geom += __shfl_down(geom, d2, 64); }
        if ((t & 63) == 0) { red[t >> 6] = arith; red[4 + (t >> 6)] = geom; }
        __syncthreads();
        const double a_sum = red[0] + red[1] + red[2] + red[3], g_sum = red[4] + red[5] + red[6] + red[7];
        const double ratio = a_sum > 1E-4 ? log(a_sum / nb) - g_sum / nb : 1.0;
        if (ratio > q.thresh) st.sq_open = q.samp_rate;         // one second timer
        pos += need; idx = 0;
        __syncthreads();
    }
    for (int k = t; k < n - pos; k += 256) rg[idx + k] = p[pos + k].x;
    if (t == 0) {
        st.index = idx + (n - pos);
        st.sq_open -= n;
        if (st.sq_open < 0) st.sq_open = 0;
        state[ch] = st;
        flag[ch].active = st.sq_open == 0;
    }
}

// d_delay (quisk.c:1057-1084) by 512 samples: dst = the stream delayed, dl_new = the 512 samples still inside the line
static __global__ __launch_bounds__(256) void q_delay_kernel(const double2 *src, long long src_stride, double2 *dst, long long dst_stride,
                                                         int n, const double2 *dl_old, double2 *dl_new)
{
    const int ch = blockIdx.y;
    const double2 *x = src + (long long)ch * src_stride, *o = dl_old + (long long)ch * 512;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n + 512; i += gridDim.x * 256) {
        if (i < n) dst[(long long)ch * dst_stride + i] = i < 512 ? o[i] : x[i - 512];
        else {
            const int k = i;                        // element n + j of [dl_old, src], j = i - n
            dl_new[(long long)ch * 512 + (i - n)] = k < 512 ? o[k] : x[k - 512];
        }
    }
}

}  // namespace qh
