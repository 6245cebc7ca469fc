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

}  // namespace qh
