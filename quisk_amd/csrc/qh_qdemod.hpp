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

// Pass 1 of the two-pass kernels below needs a segment's response to its own samples only as far back as the pole remembers:
// contributions older than kTail batches are below pole^(64 kTail) <= 1e-22 of full scale, far under the last bit of what they are
// added to.  Batches of a segment ahead of that are skipped in pass 1 (a segment shorter than the tail is walked whole: exact).
__host__ __device__ __forceinline__ int seg_tail_batches(double pole)
{
    const double t = -50.66 / (64.0 * log(fabs(pole)));        // ln 1e-22 = -50.66
    return t < 1.0e6 ? (int)t + 1 : 1000000;
}

// The same over kSegWaves time segments, one wavefront each (long calls: the sequential form leaves one wavefront per receiver
// busy for milliseconds): pass 1 = each segment's response to its own magnitudes from a zero state, chained; pass 2 = the scan
// from the true carry and the first difference.
static __global__ __launch_bounds__(kSegThreads) void q_am_env_tiled_kernel(double2 *buf, long long stride, int n, double *dc_state)
{
    __shared__ double s_e[kSegWaves];
    __shared__ int s_n[kSegWaves];
    const int ch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double2 *p = buf + (long long)ch * stride;
    int b0, b1;
    seg_range(n, wave, b0, b1);
    const PoleScan sc = make_pole_scan(0.99, lane);
    const double m64 = lane_pow(0.99, 64);
    double acc = 0.0;
    double2 zn[kSegGroup];
    const int tail = seg_tail_batches(0.99), bs = b1 - b0 > tail ? b1 - tail : b0;
    seg_load(zn, bs, b1, n, lane, (const double2 *)p);
    for (int b = bs; b < b1; b += kSegGroup) {
        double2 zz[kSegGroup];
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) zz[k] = zn[k];
        seg_load(zn, b + kSegGroup, b1, n, lane, (const double2 *)p);
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) {
            if (b + k >= b1) break;
            acc = __builtin_fma(acc, m64, hypot(zz[k].x, zz[k].y));        // lanes past the end of the call hold zeros
        }
    }
    const double e = wave_sum_d(acc * lane_pow(0.99, 63 - lane));
    if (lane == 0) { s_e[wave] = e; s_n[wave] = seg_samples(n, b0, b1); }
    const double c_in = dc_state[ch];                    // ahead of the barrier: the last wavefront stores the new carry at its end
    __syncthreads();
    double c = c_in;
    for (int w = 0; w < wave; w++) if (s_n[w]) c = __builtin_fma(c, pow(0.99, (double)s_n[w]), s_e[w]);
    seg_load(zn, b0, b1, n, lane, (const double2 *)p);
    for (int b = b0; b < b1; b += kSegGroup) {
        double2 zz[kSegGroup];
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) zz[k] = zn[k];
        seg_load(zn, b + kSegGroup, b1, n, lane, (const double2 *)p);
#pragma unroll
        for (int k = 0; k < kSegGroup; k++) {
            if (b + k >= b1) break;
            const int base = (b + k) * 64, cnt = n - base < 64 ? n - base : 64;
            const double di = lane < cnt ? hypot(zz[k].x, zz[k].y) : 0.0;
            const double dc = scan_pole_dpp(di, sc) + sc.pw * c;
            double prev = wave_shr1(dc);
            if (lane == 0) prev = c;
            const double out = dc - prev;
            if (lane < cnt) p[base + lane] = make_double2(out, out);
            c = lane_bcast(dc, cnt - 1);
        }
    }
    int last = kSegWaves - 1;
    while (last > 0 && s_n[last] == 0) last--;
    if (wave == last && lane == 0 && n > 0) dc_state[ch] = c;
}

// Quisk's FM detector (quisk.c:2032-2064): di = arg(z * conj(z_prev)) * 20e5, then the one-pole de-emphasis
// y = di*a0 + x1*a1 - y1*b1.  state: {z_prev.re, z_prev.im, x1, y1}.  In place, (y, 0).
struct QFmParam { double a0, a1, b1; };
struct QSquelchState { double rf_sum, squelch; int rf_count, active; };     // FM squelch, see q_fm_squelch_kernel
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


// the squelch's per-call update from the call's sum of |cx| (quisk.c:2076-2085)
// defer: a PIECE of a call (qh_qps.hip cuts long calls into pieces) only adds to the sums -- the reference looks at its count once per
// call of quisk_process_samples, so the windows of >= 2400 samples must end where its calls end; q_squelch_close_kernel does that look
__device__ __forceinline__ void q_squelch_update(QSquelchState *state, const double *level, int ch, double s, int n, int defer = 0)
{
    QSquelchState st = state[ch];
    st.rf_sum += s; st.rf_count += n;
    if (defer) { state[ch] = st; return; }
    if (st.rf_count >= 2400) {
        double v = st.rf_sum / st.rf_count / 2147483647.0;
        st.squelch = v > 1.E-10 ? 20 * log10(v) : -200.0;
        st.rf_sum = 0; st.rf_count = 0;
    }
    st.active = st.squelch < level[ch];
    state[ch] = st;
}

// The same detector for long calls, over a grid of time segments: one wavefront per segment of seg_b batches of 64 samples.  The
// discriminator needs the two samples ahead of a segment, the de-emphasis (pole 0.96) what the tail_b batches ahead of it leave
// (seg_tail_batches: older input is below 1e-22 of full scale) -- a wavefront walks those first from a zero state without storing,
// a segment that begins within tail_b batches of the call's start walks from sample 0 and the carried state and is exact.
// src != dst: the neighbours' samples are read while they write.  The squelch's sum of |cx| (quisk.c:2032) rides along, one
// partial sum per segment; q_fm_disc_finish_kernel adds them in order and moves the new state in.
// grid (segments / 4 rounded up, receivers), 256 threads.
static __global__ __launch_bounds__(256) void q_fm_disc_grid_kernel(const double2 *src, long long sstride, double2 *dst, long long dstride, int n,
                                                                   const double4 *state, double4 *state_new, QFmParam q, int seg_b, int tail_b,
                                                                   double *sq_part, int nseg)
{
    const int ch = blockIdx.y, lane = threadIdx.x & 63, seg = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long s0 = (long long)seg * seg_b * 64;
    if (s0 >= n) return;
    const double2 *p = src + (long long)ch * sstride;
    double2 *o = dst + (long long)ch * dstride;
    const long long s1 = s0 + (long long)seg_b * 64 < n ? s0 + (long long)seg_b * 64 : n;
    long long start = s0 - (long long)tail_b * 64;
    const double pole = -q.b1;
    const PoleScan sc = make_pole_scan(pole, lane);
    auto disc = [&](double2 z, double2 zp) -> double { return atan2(z.y * zp.x - z.x * zp.y, z.x * zp.x + z.y * zp.y) * 20e5; };
    double2 zc;
    double dc, c;
    if (start <= 0) {
        const double4 st0 = state[ch];
        start = 0; zc = make_double2(st0.x, st0.y); dc = st0.z; c = st0.w;
    } else {
        zc = p[start - 1]; dc = disc(zc, p[start - 2]); c = 0.0;
    }
    double sq = 0.0;
    double2 zn = start + lane < n ? p[start + lane] : make_double2(0.0, 0.0);
    for (long long base = start; base < s1; base += 64) {
        const int cnt = (int)(n - base < 64 ? n - base : 64);
        const double2 z = zn;
        if (base + 64 < s1) zn = base + 64 + lane < n ? p[base + 64 + lane] : make_double2(0.0, 0.0);       // the next batch is on its way
        double2 zp = make_double2(wave_shr1(z.x), wave_shr1(z.y));
        if (lane == 0) zp = zc;
        const double di = disc(z, zp);
        double dm1 = wave_shr1(di);
        if (lane == 0) dm1 = dc;
        const double y = scan_pole_dpp(lane < cnt ? di * q.a0 + dm1 * q.a1 : 0.0, sc) + sc.pw * c;
        if (base >= s0 && lane < cnt) {
            o[base + lane] = make_double2(y, 0.0);
            sq += hypot(z.x, z.y);
        }
        c = lane_bcast(y, cnt - 1);
        zc = make_double2(lane_bcast(z.x, cnt - 1), lane_bcast(z.y, cnt - 1));
        dc = lane_bcast(di, cnt - 1);
    }
    sq = wave_sum_d(sq);
    if (lane == 0) {
        sq_part[(long long)ch * nseg + seg] = sq;
        if (s1 == n) state_new[ch] = make_double4(zc.x, zc.y, dc, c);
    }
}
// one thread per receiver: the squelch from the segments' sums in order (quisk.c:2076-2085), the detector's new state
static __global__ void q_fm_disc_finish_kernel(int nch, double4 *state, const double4 *state_new, const double *sq_part, int nseg, int n,
                                               QSquelchState *sq_state, const double *sq_level, int defer = 0)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nch) return;
    state[ch] = state_new[ch];
    double s = 0.0;
    for (int k = 0; k < nseg; k++) s += sq_part[(long long)ch * nseg + k];
    q_squelch_update(sq_state, sq_level, ch, s, n, defer);
}
static __global__ void q_squelch_close_kernel(int nch, QSquelchState *sq_state, const double *sq_level)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < nch) q_squelch_update(sq_state, sq_level, ch, 0.0, 0);
}

// FM squelch (quisk.c:2032-2033,2076-2085): the mean |cx| of the Rx-filtered samples over at least 2400 of them
// (evaluated once per call, like the reference) in dB re full scale; active while it is below squelch_level.
// One wave per channel; `buf` is the Rx filter's output of this call.
static __global__ __launch_bounds__(kSegThreads) void q_fm_squelch_kernel(const double2 *buf, long long stride, int n, QSquelchState *state,
                                                             const double *level, int defer = 0)
{
    // blockDim.x = 64 (short calls) or kSegThreads: the lanes stride over the call, a fixed reduction order joins them
    __shared__ double s_part[kSegWaves];
    const int ch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const double2 *p = buf + (long long)ch * stride;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += hypot(p[i].x, p[i].y);
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane == 0) s_part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0.0;
        for (int w = 0; w < nw; w++) s += s_part[w];
        q_squelch_update(state, level, ch, s, n, defer);
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

// dAutoNotch (quisk.c:786-963): overlap-save on 2048-sample blocks of the real audio (510 old + 1538 new), the two
// strongest averaged bins tracked with a hysteresis count each, a 511-tap notch designed by frequency sampling when
// the pair changes.  One workgroup per stream; the 2048-point transforms are TileFft<2048> on (x, 0) pairs -- the
// audio runs at 6..48 ksps, one block per 32..256 ms of signal, so real-input tricks would buy nothing.  The
// per-block decisions (argmax with the reference's first-wins tie rule, counts, filter signature) are taken by
// lane 0 from block-wide reductions.  State lives in global memory in the reference's layout (data_in / data_out
// rings with the write index, average_fft, fltr_fft), so a call may end anywhere inside a block.
struct QNotchState {
    int index, fltrSig, old1, count1, old2, count2, pad0, pad1;
    double data_in[2048], data_out[2048], average_fft[1025];
    double2 fltr_fft[2048];         // bins 0..1024 are the reference's array; the upper half is its Hermitian mirror
};

__device__ __forceinline__ void notch_argmax_reduce(double &v, int &i, double *sv, int *si)
{
    // larger value wins, the lower index on equal values (the reference scans upwards with a strict >)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double ov = __shfl_xor(v, d, 64);
        const int oi = __shfl_xor(i, d, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    const int t = threadIdx.x;
    __syncthreads();
    if ((t & 63) == 0) { sv[t >> 6] = v; si[t >> 6] = i; }
    __syncthreads();
    v = sv[0]; i = si[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
        if (sv[w] > v || (sv[w] == v && si[w] < i)) { v = sv[w]; i = si[w]; }
}

// dAutoNotch(NULL, 0, 0, 0), quisk.c:826-834: everything but the filter spectrum starts over
static __global__ __launch_bounds__(256) void q_autonotch_init_kernel(QNotchState *state)
{
    QNotchState &st = state[blockIdx.x];
    for (int j = threadIdx.x; j < 2048; j += 256) { st.data_in[j] = 0.0; st.data_out[j] = 0.0; if (j < 1025) st.average_fft[j] = 0.0; }
    if (threadIdx.x == 0) { st.average_fft[1024] = 0.0; st.index = 510; st.fltrSig = -1; st.old1 = st.old2 = 0; st.count1 = st.count2 = -4; }
}

static __global__ __launch_bounds__(256) void q_autonotch_kernel(double2 *buf, long long stride, int n, QNotchState *state,
                                                                 const double2 *tw2048, int sidetone, int rate, int dup)
{
    using Fwd = TileFft<2048, false, double2>;
    using Inv = TileFft<2048, true, double2>;
    extern __shared__ __align__(16) unsigned char notch_smem[];
    __shared__ double s_avg[1025], s_cos[512], s_c2r[512], s_v[4];
    __shared__ int s_i[4], s_dec[4];
    const int ch = blockIdx.x, t = threadIdx.x;
    double2 *p = buf + (long long)ch * stride;
    QNotchState &st = state[ch];
    int pos = 0, idx = st.index;
    for (int j = t; j < 512; j += 256) s_cos[j] = cospi(2.0 * j / 512.0);
    while (pos < n) {
        int take = 2048 - idx;
        if (take > n - pos) take = n - pos;
        for (int j = t; j < take; j += 256) {                   // newest sample in, filtered sample out (quisk.c:840-841)
            const double v = p[pos + j].x, o = st.data_out[idx + j];
            st.data_in[idx + j] = v;
            p[pos + j].x = o;
            if (dup) p[pos + j].y = o;
        }
        idx += take; pos += take;
        if (idx < 2048) break;
        idx = 510;                                              // NOTCH_DATA_START_SIZE
        __syncthreads();                                        // data_in of this block is complete (same workgroup wrote it)
        // ---- forward transform of the block
        double2 x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) x[r] = make_double2(st.data_in[t + 256 * r], 0.0);
        __syncthreads();
        Fwd::run(x, notch_smem, Fwd::load(tw2048));
        // ---- averaged magnitudes of bins 0..1024 (lane t holds bins t + 256 r)
        const int delta_sig = (300 * 2 * 1025 + rate / 2) / rate, delta_i1 = (400 * 2 * 1025 + rate / 2) / rate;
        const int signal = sidetone != 0 ? ((sidetone < 0 ? -sidetone : sidetone) * 2 * 1025 + rate / 2) / rate : -999;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int i = t + 256 * r;
            if (i <= 1024) {
                const double a = 0.5 * st.average_fft[i] + 0.5 * hypot(x[r].x, x[r].y);
                st.average_fft[i] = a;
                s_avg[i] = a;
            }
        }
        __syncthreads();
        double v1 = 0.0; int i1 = 0;                            // first maximum (quisk.c:857-869): d1 = 0, i1 = 0 to start
        for (int i = t; i <= 1024; i += 256) {
            const int ds = i - signal < 0 ? signal - i : i - signal;
            if (ds > delta_sig && (s_avg[i] > v1)) { v1 = s_avg[i]; i1 = i; }
        }
        if (!(v1 > 0.0)) i1 = 0;
        notch_argmax_reduce(v1, i1, s_v, s_i);
        double v2 = 0.0; int i2 = 0;                            // next maximum not near the first (quisk.c:881-888)
        for (int i = t; i <= 1024; i += 256) {
            const int ds = i - signal < 0 ? signal - i : i - signal, d1 = i - i1 < 0 ? i1 - i : i - i1;
            if (ds > delta_sig && d1 > delta_i1 && s_avg[i] > v2) { v2 = s_avg[i]; i2 = i; }
        }
        if (!(v2 > 0.0)) i2 = 0;
        notch_argmax_reduce(v2, i2, s_v, s_i);
        if (t == 0) {
            int c1 = st.count1, c2 = st.count2;
            const int a1 = i1 - st.old1 < 0 ? st.old1 - i1 : i1 - st.old1, a2 = i2 - st.old2 < 0 ? st.old2 - i2 : i2 - st.old2;
            c1 += a1 < 3 ? 1 : -1;
            if (c1 > 4) c1 = 4; else if (c1 < -1) c1 = -1;
            if (c1 < 0) st.old1 = i1;
            c2 += a2 < 3 ? 1 : -1;
            if (c2 > 4) c2 = 4; else if (c2 < -2) c2 = -2;
            if (c2 < 0) st.old2 = i2;
            st.count1 = c1; st.count2 = c2;
            const int sig = (c1 > 0 && c2 > 0) ? i1 + 10000 * i2 : c1 > 0 ? i1 : 0;
            s_dec[0] = st.fltrSig != sig;
            s_dec[1] = c1 > 0;
            s_dec[2] = c1 > 0 && c2 > 0;
            st.fltrSig = sig;
        }
        __syncthreads();
        if (s_dec[0]) {
            // ---- design (quisk.c:905-942): c2r of a 0/1 spectrum of 256 bins (+ bin 256 = whatever the last design
            // left in fltr_fft[256]), centred with the reference's memmove / mirror, Hanning window, r2c at 2048
            int half_width = (100 * 2 * 256 + rate / 2) / rate;
            if (half_width < 3) half_width = 3;
            const int k1 = (i1 + 2) / 4, k2 = (i2 + 2) / 4, on1 = s_dec[1], on2 = s_dec[2];
            const double f256 = st.fltr_fft[256].x;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int m = t + 256 * h;
                double acc = 0.0;
                for (int k = 1; k < 256; k++) {
                    const int da = k - k1 < 0 ? k1 - k : k - k1, db = k - k2 < 0 ? k2 - k : k - k2;
                    const bool zero = (on1 && da <= half_width) || (on2 && db <= half_width);
                    if (!zero) acc += s_cos[(k * m) & 511];
                }
                const bool zero0 = (on1 && k1 <= half_width) || (on2 && k2 <= half_width);      // bin 0 inside a notch
                s_c2r[m] = (zero0 ? 0.0 : 1.0) + ((m & 1) ? -f256 : f256) + 2.0 * acc;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int i = t + 256 * r;
                double f = 0.0;
                if (i < 511) {
                    // fltr_out after "memmove(fltr_out + 255, fltr_out, 254 doubles)" and the mirror loop
                    const double o = i >= 509 ? s_c2r[i] : i >= 255 ? s_c2r[i - 255] : i >= 2 ? s_c2r[255 - i] : s_c2r[510 - i];
                    f = o * (0.50 - 0.50 * cospi(2.0 * i / 511.0)) / 2048.0 / 4.0;
                }
                x[r] = make_double2(f, 0.0);
            }
            __syncthreads();
            Fwd::run(x, notch_smem, Fwd::load(tw2048));
#pragma unroll
            for (int r = 0; r < 8; r++) st.fltr_fft[t + 256 * r] = x[r];
            __syncthreads();
            // redo the block's transform (a design is rare; keeping 8 more complex registers live all the time is not)
#pragma unroll
            for (int r = 0; r < 8; r++) x[r] = make_double2(st.data_in[t + 256 * r], 0.0);
            Fwd::run(x, notch_smem, Fwd::load(tw2048));
        }
        // ---- apply the filter and transform back (quisk.c:946-952)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double2 f = st.fltr_fft[t + 256 * r];
            x[r] = make_double2(x[r].x * f.x - x[r].y * f.y, x[r].x * f.y + x[r].y * f.x);
        }
        __syncthreads();
        Inv::run(x, notch_smem, Inv::load(tw2048));
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int i = t + 256 * r;
            if (i >= 510) st.data_out[i] = x[r].x / 102.0;      // NOTCH_DATA_SIZE / 20 in integers: "Empirical"
        }
        // memmove(data_in, data_in + NOTCH_DATA_OUTPUT_SIZE, NOTCH_DATA_START_SIZE doubles)
        double mv0 = 0.0, mv1 = 0.0;
        if (t < 510) mv0 = st.data_in[1538 + t];
        if (t + 256 < 510) mv1 = st.data_in[1538 + t + 256];
        __syncthreads();
        if (t < 510) st.data_in[t] = mv0;
        if (t + 256 < 510) st.data_in[t + 256] = mv1;
        __syncthreads();
    }
    if (t == 0) st.index = idx;
}

}  // namespace qh
