// qh_analyzer.hip -- WDSP's display engine (wdsp/analyzer.c) for a bank of displays (include/quiskhip.h group 11).
//
// The reference keeps a float sample ring per (display, sub-span), a dispatcher thread that polls the rings and a worker
// thread per frame: window, FFT (FFTW), |X|^2 of the kept bins in display order (Celiminate / eliminate), the sub-spans
// concatenated (stitch), reduced to pixels by one of five detectors -- or interpolated when the pixels outnumber the bins
// -- averaged in one of five ways, converted to dB (mlog10) and published for GetPixels.
//
// Here a call moves every frame that has become complete, of every display, through three launches:
//   ana_fft_kernel      one workgroup per (residue, frame x sub-span, display): N = R * M points by decimation in frequency,
//                       X[R k + r] = FFT_M{ W_N^(m r) sum_q W_R^(q r) w[m + M q] x[m + M q] }[k], M <= 8192 in LDS, R <= 64;
//                       writes |X|^2 in natural bin order
//   ana_detect_kernel   one lane per pixel (or rosenfell segment): everything that does not depend on the data -- which
//                       bins feed which pixel, the interpolation weights, the order of the clipped / flipped / stitched
//                       bins -- is tabulated by the host when SetAnalyzer runs, WITH THE REFERENCE'S OWN EXPRESSIONS
//                       (truncations, the offset-less "next pixel" of the rosenfell detector, the accumulated pixel
//                       position of the interpolator), so the lanes only gather and reduce, in the reference's order
//   ana_average_kernel  one lane per pixel walks the call's frames in time order through the averaging recurrence
// The sample streams stay on the device (float pairs, as the reference's dINREAL rings outside Thetis, comm.h:128-132).
#include <algorithm>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>
#include "qh_internal.hpp"
#include "qh_design.hpp"
#include "qh_fft.hpp"
#include "qh_wave.hpp"

using namespace qh;

namespace {

constexpr int kMaxStitch = 4, kMaxPixels = 16384, kMaxAverage = 60, kMaxPixouts = 4, kMaxN = 100, kMaxCalSets = 2;   // comm.h:123-139
constexpr int kMaxR = 64;

// ---- kernels --------------------------------------------------------------------------------------------------------
// old tail + new samples -> the other stream buffer; new samples arrive as (I, Q) doubles and are narrowed to float
__global__ void ana_append_kernel(const float2 *old_buf, long long old_stride, int drop, int keep, const double2 *src, long long src_stride,
                                  int n, int swap_iq, float2 *dst, long long dst_stride)
{
    const int d = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= keep + n) return;
    float2 v;
    if (i < keep) v = old_buf[(long long)d * old_stride + drop + i];
    else {
        const double2 z = src[(long long)d * src_stride + (i - keep)];
        v = swap_iq ? make_float2((float)z.y, (float)z.x) : make_float2((float)z.x, (float)z.y);
    }
    dst[(long long)d * dst_stride + i] = v;
}

struct AnaFftArgs {
    const float2 *sbuf[kMaxStitch];     // per sub-span: [ndisp][stride]
    long long stride[kMaxStitch];
    const int *starts;                  // [nframes][nss]: index of the frame's first sample in the sub-span's buffer
    const double *window;
    const double2 *tw;
    double *pw;                         // [ndisp][nframes][nss][N]
    int nss, ss0, N, R, real_input, nframes;
    // SnapSpectrum (analyzer.c:708-713,1337-1346): the transform itself of one (display, sub-span), first frame of the call, natural bin order
    double2 *snap;
    int snap_disp, snap_ss;
};

template <int M> __global__ __launch_bounds__(NT) void ana_fft_kernel(AnaFftArgs a)
{
    using C = double2;
    using Fwd = TileFft<M, false, C>;
    constexpr int E = M / NT;
    extern __shared__ __align__(16) unsigned char ana_smem[];
    const int t = threadIdx.x, r = blockIdx.x, fs = blockIdx.y, d = blockIdx.z;
    const int f = fs / a.nss, s = fs - f * a.nss;
    const float2 *x = a.sbuf[a.ss0 + s] + (long long)d * a.stride[a.ss0 + s] + a.starts[f * kMaxStitch + a.ss0 + s];
    C acc[E];
#pragma unroll
    for (int j = 0; j < E; j++) acc[j] = make_double2(0.0, 0.0);
    for (int q = 0; q < a.R; q++) {
        double wr = 1.0, wi = 0.0;                      // W_R^(q r)
        if (a.R > 1) sincospi(-2.0 * (double)((q * r) % a.R) / (double)a.R, &wi, &wr);
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int idx = t + NT * j + M * q;
            const float2 v = x[idx];
            const double w = a.window[idx];
            const double re = w * (double)v.x, im = a.real_input ? 0.0 : w * (double)v.y;     // analyzer.c:617,691-692
            acc[j].x += re * wr - im * wi;
            acc[j].y += re * wi + im * wr;
        }
    }
    if (a.R > 1) {
#pragma unroll
        for (int j = 0; j < E; j++) {                   // W_N^(m r)
            double c, sn;
            sincospi(-2.0 * (double)(((long long)(t + NT * j) * r) % a.N) / (double)a.N, &sn, &c);
            const C v = acc[j];
            acc[j] = make_double2(v.x * c - v.y * sn, v.x * sn + v.y * c);
        }
    }
    Fwd::run(acc, ana_smem, Fwd::load(a.tw));
    double *out = a.pw + (((long long)d * a.nframes + f) * a.nss + s) * a.N;
#pragma unroll
    for (int j = 0; j < E; j++) {
        const int k = t + NT * j;
        out[(long long)a.R * k + r] = acc[j].x * acc[j].x + acc[j].y * acc[j].y;           // analyzer.c:200,243
    }
    if (a.snap && f == 0 && d == a.snap_disp && a.ss0 + s == a.snap_ss) {                  // workgroup-uniform
#pragma unroll
        for (int j = 0; j < E; j++) a.snap[(long long)a.R * (t + NT * j) + r] = acc[j];
    }
}

struct AnaDetArgs {
    const double *pw;           // [ndisp][nframes][span] with span = nss * N
    const int *src;             // [m] stitched position -> ss_local * N + bin
    const int *lo, *hi;         // [npix] bins of a pixel (pix_per_bin <= 1)
    const int *seg;             // rosenfell: [nseg][4] first bin, last bin, pixel, written
    const int *ip_i;            // interpolation: left bin per pixel, -1 = not reached
    const double *ip_frac;
    double *tframes;            // [ndisp][nframes][npix]
    double inv_enb;
    long long span;
    int det, npix, nseg, nframes, interp, m;
};

__global__ void ana_detect_kernel(AnaDetArgs a)
{
#pragma clang fp contract(off)
    const int d = blockIdx.z, f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const double *pw = a.pw + ((long long)d * a.nframes + f) * a.span;
    double *tp = a.tframes + ((long long)d * a.nframes + f) * a.npix;
    auto bin = [&](int j) -> double { return pw[a.src[j]]; };
    if (a.interp) {                                     // analyzer.c:444-459
        if (i >= a.npix) return;
        const int b = a.ip_i[i];
        if (b < 0) return;
        const double frac = a.ip_frac[i];
        double v = bin(b) * (1.0 - frac) + bin(b + 1) * frac;
        if (a.det == 2 || a.det == 3 || a.det == 4) v *= a.inv_enb;
        tp[i] = v;
        return;
    }
    if (a.det == 1) {                                   // rosenfell, analyzer.c:322-365: one lane per finished pixel segment
        if (i >= a.nseg) return;
        const int b0 = a.seg[4 * i], b1 = a.seg[4 * i + 1], pix = a.seg[4 * i + 2];
        if (!a.seg[4 * i + 3]) return;                  // a later segment lands on the same pixel
        double mini = 1.0e300, maxi = -1.0e300, prev_maxi = -1.0e300;
        bool rose = false, fell = false;
        double cur = bin(b0);
        for (int j = b0; j <= b1; j++) {
            if (cur < mini) mini = cur;
            if (cur > maxi) maxi = cur;
            if (j < b1) {
                const double nx = bin(j + 1);
                if (nx > cur) rose = true;
                if (nx < cur) fell = true;
                cur = nx;
            }
        }
        if (i > 0) for (int j = a.seg[4 * (i - 1)]; j <= a.seg[4 * (i - 1) + 1]; j++) { const double v = bin(j); if (v > prev_maxi) prev_maxi = v; }
        double v = maxi;
        if (rose && fell) v = (pix & 1) ? (prev_maxi > maxi ? prev_maxi : maxi) : mini;
        tp[pix] = v;
        return;
    }
    if (i >= a.npix) return;
    const int lo = a.lo[i], hi = a.hi[i];
    if (a.det == 0) {                                   // positive peak, analyzer.c:308-320
        double px = -1.0e300;
        for (int j = lo; j < hi; j++) { const double v = bin(j); if (v > px) px = v; }
        tp[i] = px;
        return;
    }
    if (hi <= lo) return;                               // no bin lands here: the pixel keeps its last value
    if (a.det == 3) { tp[i] = bin(hi - 1 - (hi - lo) / 2) * a.inv_enb; return; }       // sample, analyzer.c:392-412
    double psum = 0.0;
    for (int j = lo; j < hi; j++) { const double v = bin(j); psum += a.det == 2 ? v : v * v; }
    const double mean = psum / (double)(hi - lo);
    tp[i] = (a.det == 2 ? mean : sqrt(mean)) * a.inv_enb;                               // analyzer.c:367-390,414-439
}

struct AnaAvgArgs {
    const double *tframes;      // [ndisp][nframes][npix] of this output's detector
    const unsigned char *valid; // [npix]
    double *t_persist;          // [ndisp][kMaxPixels]
    double *av_sum;             // [ndisp][kMaxPixels]
    double *av_buff;            // [ndisp][kMaxAverage][kMaxPixels] (mode 2)
    const double *cd;           // [npix]
    float *rows;                // [ndisp][nframes][npix]
    float *latest;              // [ndisp][kMaxPixels]
    double scale, back, norm_oneHz;
    int mode, npix, nframes, avail, num_average, in_idx, out_idx, normalize;
};

__global__ void ana_average_kernel(AnaAvgArgs a)      // avenger, analyzer.c:463-553
{
#pragma clang fp contract(off)
    const int d = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.npix) return;
    const long long di = (long long)d * kMaxPixels + i;     // the per-pixel state keeps its place when num_pixels changes (dMAX_PIXELS arrays, analyzer.c:1196-1203)
    double t = a.t_persist[di], sum = a.av_sum[di];
    const double cd = a.cd[i], onem = 1.0 - a.back;
    const bool valid = a.valid[i] != 0;
    int avail = a.avail, in_idx = a.in_idx, out_idx = a.out_idx;
    float px = 0.0f;
    for (int f = 0; f < a.nframes; f++) {
        if (valid) t = a.tframes[((long long)d * a.nframes + f) * a.npix + i];
        switch (a.mode) {
        case -1:
            if (t > sum) sum = t;
            px = (float)(10.0 * mlog10_dev(a.scale * cd * sum + 1.0e-60));
            break;
        case 1:
            sum = a.back * sum + onem * t;
            px = (float)(10.0 * mlog10_dev(a.scale * cd * sum + 1.0e-60));
            break;
        case 2: {
            double *ring = a.av_buff + (long long)d * kMaxAverage * kMaxPixels + i;
            double factor;
            if (avail < a.num_average) {
                factor = a.scale / (double)++avail;
                sum += t;
            } else {
                factor = a.scale / (double)avail;
                sum += t - ring[(long long)out_idx * kMaxPixels];
                if (++out_idx == kMaxAverage) out_idx = 0;
            }
            ring[(long long)in_idx * kMaxPixels] = t;
            if (++in_idx == kMaxAverage) in_idx = 0;
            px = (float)(10.0 * mlog10_dev(cd * sum * factor + 1.0e-60));
            break; }
        case 3:
            sum = a.back * sum + onem * (10.0 * mlog10_dev(a.scale * cd * t + 1e-60));
            px = (float)sum;
            break;
        default:
            px = (float)(10.0 * mlog10_dev(a.scale * cd * t + 1.0e-60));
            break;
        }
        if (a.normalize) px += (float)a.norm_oneHz;
        a.rows[((long long)d * a.nframes + f) * a.npix + i] = px;
    }
    a.t_persist[di] = t; a.av_sum[di] = sum;
    if (a.nframes > 0) a.latest[di] = px;
}

__global__ void ana_fill_kernel(double *p, long long n, double v)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- host side ------------------------------------------------------------------------------------------------------
double bessi0(double x)         // analyzer.c:33-50 (Abramowitz & Stegun 9.8.1, 9.8.2)
{
    static const double s[7] = {1.0, 3.5156229, 3.0899424, 1.2067492, 0.2659732, 0.360768e-1, 0.45813e-2};
    static const double l[9] = {0.39894228, 0.1328592e-1, 0.225319e-2, -0.157565e-2, 0.916281e-2, -0.2057706e-1, 0.2635537e-1, -0.1647633e-1,
                                0.392377e-2};
    const double ax = std::fabs(x);
    if (ax < 3.75) {
        double y = x / 3.75; y *= y;
        double p = s[6];
        for (int k = 5; k >= 0; k--) p = s[k] + y * p;
        return p;
    }
    const double y = 3.75 / ax;
    double p = l[8];
    for (int k = 7; k >= 0; k--) p = l[k] + y * p;
    return (std::exp(ax) / std::sqrt(ax)) * p;
}

double host_mlog10(double val)  // wdsp/meterlog10.c:29-32,547-554
{
    unsigned long long N;
    std::memcpy(&N, &val, 8);
    const int e = (int)((N >> 52) & 2047) - 1023, m = (int)((N >> (52 - 11)) & 2047);
    return 0.301029995663981 * (e + std::log2(1.0 + m / 2048.0));
}

template <typename T> struct DevVec {
    T *p = nullptr;
    size_t cap = 0;
    ~DevVec() { (void)hipFree(p); }
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        (void)hipFree(p); p = nullptr; cap = 0;
        const hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    hipError_t upload(const std::vector<T> &v, hipStream_t s)
    {
        hipError_t e = ensure(v.size() ? v.size() : 1);
        if (e == hipSuccess && !v.empty()) e = hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
        return e;
    }
};

struct DetTables {              // one per detector type in use
    int det = -1;
    int nseg = 0;
    DevVec<int> lo, hi, seg, ip_i;
    DevVec<double> ip_frac, tframes, t_persist;
    DevVec<unsigned char> valid;
};

}  // namespace

struct qh_ana {
    std::recursive_mutex mu;
    // SnapSpectrum: one request at a time -- armed for (display, sub-span), filled by the next frame of that pair
    std::mutex snap_mu;
    std::condition_variable snap_cv;
    bool snap_armed = false, snap_done = false;
    int snap_disp = 0, snap_ss = 0;
    DevVec<double2> d_snap;
    std::vector<double> snap_host;      // the transform, fft-shifted as analyzer.c:710-711 copies it: bins N/2 .. N-1, then 0 .. N/2-1
    int device = 0, ndisp = 0, max_size = 0, max_stitch = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false, configured = false;
    // SetAnalyzer
    int num_pixout = 1, type = 1, flip = 0, size = -1, out_size = 0, buff_size = 0, window_type = -1, overlap = 0, clip = 0, num_pixels = -1,
        num_stitch = 1, cal_set = -1, max_writeahead = 0, incr = 0, begin_ss = 0, end_ss = 0, fscL = 0, fscH = 0, cal_changed = 0, sample_rate = 0;
    double PiAlpha = 0, fsclipL = 0, fsclipH = 0, f_min = -1.0, f_max = -1.0, pix_per_bin = 0, bin_per_pix = 0, det_offset = 0, scale = 1, inv_enb = 1,
           inv_coherent_gain = 1, inherent_power_gain = 1, norm_oneHz = 0;
    int R = 1, M = 0, m_bins = 0;
    std::vector<double> h_window, h_cd;
    int n_freqs[kMaxCalSets] = {0, 0};
    std::vector<double> freqs[kMaxCalSets], ac3[kMaxCalSets], ac2[kMaxCalSets], ac1[kMaxCalSets], ac0[kMaxCalSets];
    // per pixel output
    int det_type[kMaxPixouts] = {0, 0, 0, 0}, av_mode[kMaxPixouts] = {0, 0, 0, 0}, num_average[kMaxPixouts] = {0, 0, 0, 0},
        normalize[kMaxPixouts] = {0, 0, 0, 0}, avail_frames[kMaxPixouts] = {0, 0, 0, 0}, av_in_idx[kMaxPixouts] = {0, 0, 0, 0},
        av_out_idx[kMaxPixouts] = {0, 0, 0, 0};
    double av_backmult[kMaxPixouts] = {0, 0, 0, 0};
    std::vector<unsigned char> fresh[kMaxPixouts];     // [ndisp]: a row GetPixels has not handed out yet
    DevVec<double> av_sum[kMaxPixouts], av_buff[kMaxPixouts];
    DevVec<float> rows[kMaxPixouts], latest[kMaxPixouts];
    int rows_frames = 0;
    // device tables
    DevVec<double> d_window, d_cd, d_pw;
    DevVec<double2> d_tw;
    DevVec<int> d_src, d_starts;
    DetTables dets[kMaxPixouts];
    bool tables_dirty = true;
    // sample streams
    DevVec<float2> sbuf[kMaxStitch][2];
    int cur[kMaxStitch] = {0, 0, 0, 0};
    long long stride[kMaxStitch][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    long long base_pos[kMaxStitch] = {0, 0, 0, 0}, in_pos[kMaxStitch] = {0, 0, 0, 0}, out_pos[kMaxStitch] = {0, 0, 0, 0}, start_pos[kMaxStitch] = {0, 0, 0, 0};
    int have[kMaxStitch] = {0, 0, 0, 0};
    bool ready[kMaxStitch] = {false, false, false, false}, busy[kMaxStitch] = {false, false, false, false};
    unsigned stitch_flag = 0;
    std::vector<int> pending;           // [frames][kMaxStitch] start positions relative to base_pos
    DevVec<double2> staging;            // host-pointer entry points
    std::vector<float> h_open_I[kMaxStitch], h_open_Q[kMaxStitch];
    long long frames_total = 0;

    ~qh_ana()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

void new_window(qh_ana &a, int type, int size, double PiAlpha)         // analyzer.c:52-176
{
    std::vector<double> &w = a.h_window;
    w.assign((size_t)size, 1.0);
    const double step = 2.0 * 3.1415926535897932 / ((double)size - 1.0);
    double cg = 0.0, ig = 0.0;
    for (int i = 0; i < size; i++) {
        const double arg = step * (double)i;
        double v = 1.0;
        switch (type) {
        case 1: v = 0.35875 - 0.48829 * std::cos(arg) + 0.14128 * std::cos(2.0 * arg) - 0.01168 * std::cos(3.0 * arg); break;
        case 2: v = 0.5 * (1.0 - std::cos((double)i * step)); break;
        case 3: v = 0.21557895 - 0.41663158 * std::cos(arg) + 0.277263158 * std::cos(2.0 * arg) - 0.083578947 * std::cos(3.0 * arg)
                    + 0.006947368 * std::cos(4.0 * arg); break;
        case 4: v = 0.54 - 0.46 * std::cos((double)i * step); break;
        case 5: v = bessi0(PiAlpha * std::sqrt(1.0 - std::pow(2.0 * (double)i / (double)(size - 1) - 1.0, 2))) / bessi0(PiAlpha); break;
        case 6: {
            const double c = std::cos(arg);
            v = 6.3964424114390378e-02 + c * (-2.3993864599352804e-01 + c * (3.5015956323820469e-01 + c * (-2.4774111897080783e-01
                + c * (8.5438256055858031e-02 + c * (-1.2320203369293225e-02 + c * 4.3778825791773474e-04)))));
            break; }
        default: break;
        }
        w[(size_t)i] = v; cg += v; ig += v * v;
    }
    a.inv_coherent_gain = 1.0;
    if (type == 0) ig = (double)size;
    else {
        a.inv_coherent_gain = (double)size / cg;
        for (double &v : w) v *= a.inv_coherent_gain;
    }
    a.inherent_power_gain = ig / (double)size;
    a.inv_enb = 1.0 / (a.inherent_power_gain * a.inv_coherent_gain * a.inv_coherent_gain);
}

void interpolate_cal(qh_ana &a, int set, double fmin, double fmax, int num_pixels)     // analyzer.c:747-798
{
    const int n = a.n_freqs[set];
    const std::vector<double> &fr = a.freqs[set];
    int k = 0, kmin = 0, kmax = n - 1;
    for (int i = 0; i < num_pixels; i++) {
        const double f = fmin + (double)i * (fmax - fmin) / (double)(num_pixels - 1);
        if (f < fr[0]) k = 0;
        else if (f > fr[(size_t)n - 1]) k = n - 2;
        else {
            int kdelta = 1;
            while (f < fr[(size_t)kmin]) { kmin = std::max(0, kmin - kdelta); kdelta += kdelta; }
            while (f > fr[(size_t)kmax]) { kmax = std::min(n - 1, kmax + kdelta); kdelta += kdelta; }
            while (kmax - kmin > 1) {
                k = (kmin + kmax) / 2;
                if (f > fr[(size_t)k]) kmin = k; else kmax = k--;
            }
        }
        const double dx = f - fr[(size_t)k];
        const double mag = ((a.ac3[set][(size_t)k] * dx + a.ac2[set][(size_t)k]) * dx + a.ac1[set][(size_t)k]) * dx + a.ac0[set][(size_t)k];
        a.h_cd[(size_t)i] = mag * mag;
    }
}

// the bins one sub-span hands to the stitcher, in its order (eliminate / Celiminate with one LO, analyzer.c:179-279)
void kept_bins(const qh_ana &a, int ss, std::vector<int> &out)
{
    const int ilim = a.out_size - 1, base = (ss - a.begin_ss) * a.size;
    auto run = [&](int from, int to, bool flip) {
        if (flip) for (int i = ilim - from; i > ilim - to; i--) out.push_back(base + i);
        else for (int i = from; i < to; i++) out.push_back(base + i);
    };
    if (a.type == 0) {
        const int begin = ss == a.begin_ss ? a.fscL + a.clip : a.clip;
        const int end = ss == a.end_ss ? a.out_size - 1 - a.clip - a.fscH : a.out_size - 1 - a.clip;
        run(begin, end, a.flip != 0);
        return;
    }
    int begin0, end0, begin1, end1;
    if (ss == a.begin_ss) {
        begin0 = a.out_size / 2 + 1 + a.clip + a.fscL;
        begin1 = begin0 > a.out_size ? begin0 - a.out_size : 0;
    } else { begin0 = a.out_size / 2 + 1 + a.clip; begin1 = 0; }
    if (ss == a.end_ss) {
        end1 = a.out_size / 2 - a.clip - a.fscH;
        end0 = end1 < 0 ? a.out_size + end1 : a.out_size;
    } else { end0 = a.out_size; end1 = a.out_size / 2 - a.clip; }
    run(begin0, end0, a.flip != 0);
    run(begin1, end1, a.flip != 0);
}

int build_tables(qh_ana &a)
{
    // stitched order
    std::vector<int> src;
    for (int ss = a.begin_ss; ss <= a.end_ss; ss++) kept_bins(a, ss, src);
    for (int v : src) if (v < 0 || v >= (a.end_ss - a.begin_ss + 1) * a.size) return set_error(QH_ERR_INVALID, "SetAnalyzer: clip / span settings leave no valid bins");
    const int m = (int)src.size();
    a.m_bins = m;
    if (m < 2) return set_error(QH_ERR_INVALID, "SetAnalyzer: fewer than two bins are kept");
    QH_HIP(a.d_src.upload(src, a.stream));
    QH_HIP(a.d_window.upload(a.h_window, a.stream));
    QH_HIP(a.d_cd.upload(a.h_cd, a.stream));
    const int npix = a.num_pixels;
    // detectors: one slot per distinct type among the outputs
    int nslot = 0;
    for (int o = 0; o < a.num_pixout; o++) {
        bool seen = false;
        for (int s = 0; s < nslot; s++) if (a.dets[s].det == a.det_type[o]) seen = true;
        if (!seen) a.dets[nslot++].det = a.det_type[o];
    }
    for (int s = nslot; s < kMaxPixouts; s++) a.dets[s].det = -1;
    const bool interp = !(a.pix_per_bin <= 1.0);
    std::vector<int> lo((size_t)npix, 0), hi((size_t)npix, 0);
    std::vector<int> seg;
    std::vector<int> ip_i((size_t)npix, -1);
    std::vector<double> ip_frac((size_t)npix, 0.0);
    if (!interp) {
        const int imin = a.fsclipL == std::floor(a.fsclipL) ? 0 : 1, ilim = a.fsclipH == std::floor(a.fsclipH) ? m : m - 1;
        int prev = -1, seg_start = imin;
        std::vector<int> last_writer((size_t)npix, -1);
        for (int i = imin; i < ilim; i++) {
            int pc = (int)(a.det_offset + (double)i * a.pix_per_bin);           // analyzer.c:315,331,373
            if (pc >= npix) pc = npix - 1;
            if (pc < 0) pc = 0;
            if (pc != prev) { lo[(size_t)pc] = i; prev = pc; }
            hi[(size_t)pc] = i + 1;
            const int next = (int)((double)(i + 1) * a.pix_per_bin);            // analyzer.c:334: no offset, no clamp
            if (!(next == pc && i < ilim - 1)) {
                last_writer[(size_t)pc] = (int)seg.size() / 4;
                seg.push_back(seg_start); seg.push_back(i); seg.push_back(pc); seg.push_back(0);
                seg_start = i + 1;
            }
        }
        for (int p = 0; p < npix; p++) if (last_writer[(size_t)p] >= 0) seg[(size_t)last_writer[(size_t)p] * 4 + 3] = 1;
    } else {
        double pix_pos = a.fsclipL - std::floor(a.fsclipL);
        int pix_count = 0;
        for (int i = 1; i < m; i++)
            while (pix_pos < (double)i + 1.0e-06 && pix_count < npix) {        // analyzer.c:449-458
                ip_i[(size_t)pix_count] = i - 1;
                ip_frac[(size_t)pix_count] = pix_pos - (double)(i - 1);
                pix_count++;
                pix_pos += a.bin_per_pix;
            }
    }
    for (int s = 0; s < nslot; s++) {
        DetTables &t = a.dets[s];
        std::vector<unsigned char> valid((size_t)npix, 0);
        if (interp) for (int p = 0; p < npix; p++) valid[(size_t)p] = ip_i[(size_t)p] >= 0;
        else if (t.det == 0) std::fill(valid.begin(), valid.end(), 1);
        else if (t.det == 1) { for (size_t k = 0; k + 3 < seg.size(); k += 4) valid[(size_t)seg[k + 2]] = 1; }
        else for (int p = 0; p < npix; p++) valid[(size_t)p] = hi[(size_t)p] > lo[(size_t)p];
        t.nseg = (int)seg.size() / 4;
        QH_HIP(t.lo.upload(lo, a.stream)); QH_HIP(t.hi.upload(hi, a.stream)); QH_HIP(t.seg.upload(seg, a.stream));
        QH_HIP(t.ip_i.upload(ip_i, a.stream)); QH_HIP(t.ip_frac.upload(ip_frac, a.stream)); QH_HIP(t.valid.upload(valid, a.stream));
        const size_t need = (size_t)a.ndisp * kMaxPixels;
        if (t.t_persist.cap < need) {
            QH_HIP(t.t_persist.ensure(need));
            QH_HIP(hipMemsetAsync(t.t_persist.p, 0, need * sizeof(double), a.stream));
        }
    }
    QH_HIP(hipStreamSynchronize(a.stream));             // the host vectors go out of scope
    a.tables_dirty = false;
    return QH_OK;
}

int fill(qh_ana &a, double *p, size_t n, double v)
{
    hipLaunchKernelGGL(ana_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, a.stream, p, (long long)n, v);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int reset_average(qh_ana &a, int o)                     // SetDisplayAverageMode's state reset, analyzer.c:1594-1623
{
    const size_t n = (size_t)a.ndisp * kMaxPixels;
    if (a.av_sum[o].cap < n) {                          // malloc0 in XCreateAnalyzer (analyzer.c:1196); mode 2 keeps what is there
        QH_HIP(a.av_sum[o].ensure(n));
        QH_HIP(hipMemsetAsync(a.av_sum[o].p, 0, n * sizeof(double), a.stream));
    }
    const int mode = a.av_mode[o];
    if (mode == 2) { a.avail_frames[o] = a.av_in_idx[o] = a.av_out_idx[o] = 0; return QH_OK; }
    return fill(a, a.av_sum[o].p, n, mode == 1 ? 1.0e-12 : mode == 3 ? -160.0 : 0.0);
}

template <int M> int launch_fft(qh_ana &a, const AnaFftArgs &args, int nframes, int nss)
{
    constexpr size_t lds = TileFft<M, false, double2>::kLdsBytes;
    static bool attr_done = false;
    if (!attr_done) {
        QH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ana_fft_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((ana_fft_kernel<M>), dim3((unsigned)a.R, (unsigned)(nframes * nss), (unsigned)a.ndisp), dim3(NT), lds, a.stream, args);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

// every frame set that is complete: FFTs, detectors, averaging
int run_pending(qh_ana &a)
{
    const int nframes = (int)a.pending.size() / kMaxStitch;
    if (nframes == 0) return QH_OK;
    if (a.tables_dirty) if (int e = build_tables(a)) return e;
    const int nss = a.end_ss - a.begin_ss + 1, npix = a.num_pixels;
    QH_HIP(a.d_starts.upload(a.pending, a.stream));
    QH_HIP(a.d_pw.ensure((size_t)a.ndisp * nframes * nss * a.size));
    AnaFftArgs fa{};
    for (int s = 0; s < kMaxStitch; s++) { fa.sbuf[s] = a.sbuf[s][a.cur[s]].p; fa.stride[s] = a.stride[s][a.cur[s]]; }
    fa.starts = a.d_starts.p; fa.window = a.d_window.p; fa.tw = a.d_tw.p; fa.pw = a.d_pw.p;
    fa.nss = nss; fa.ss0 = a.begin_ss; fa.N = a.size; fa.R = a.R; fa.real_input = a.type == 0; fa.nframes = nframes;
    const bool snapping = a.snap_armed && a.snap_ss >= a.begin_ss && a.snap_ss <= a.end_ss;
    if (snapping) {
        QH_HIP(a.d_snap.ensure((size_t)a.size));
        fa.snap = a.d_snap.p; fa.snap_disp = a.snap_disp; fa.snap_ss = a.snap_ss;
    }
    int rc;
    switch (a.M) {
    case 512: rc = launch_fft<512>(a, fa, nframes, nss); break;
    case 1024: rc = launch_fft<1024>(a, fa, nframes, nss); break;
    case 2048: rc = launch_fft<2048>(a, fa, nframes, nss); break;
    case 4096: rc = launch_fft<4096>(a, fa, nframes, nss); break;
    default: rc = launch_fft<8192>(a, fa, nframes, nss); break;
    }
    if (rc) return rc;
    if (snapping) {
        std::vector<double> x((size_t)a.size * 2);
        QH_HIP(hipMemcpyAsync(x.data(), a.d_snap.p, (size_t)a.size * 16, hipMemcpyDeviceToHost, a.stream));
        QH_HIP(hipStreamSynchronize(a.stream));
        {
            std::lock_guard<std::mutex> sl(a.snap_mu);
            a.snap_host.resize((size_t)a.size * 2);
            const size_t half = (size_t)a.size;              // doubles in half of the transform (size complex values = 2 size doubles)
            std::copy(x.begin() + (long)half, x.end(), a.snap_host.begin());
            std::copy(x.begin(), x.begin() + (long)half, a.snap_host.begin() + (long)half);
            a.snap_armed = false; a.snap_done = true;
        }
        a.snap_cv.notify_all();
    }
    const bool interp = !(a.pix_per_bin <= 1.0);
    for (int s = 0; s < kMaxPixouts && a.dets[s].det >= 0; s++) {
        DetTables &t = a.dets[s];
        QH_HIP(t.tframes.ensure((size_t)a.ndisp * nframes * npix));
        AnaDetArgs da{};
        da.pw = a.d_pw.p; da.src = a.d_src.p; da.lo = t.lo.p; da.hi = t.hi.p; da.seg = t.seg.p; da.ip_i = t.ip_i.p; da.ip_frac = t.ip_frac.p;
        da.tframes = t.tframes.p; da.inv_enb = a.inv_enb; da.span = (long long)nss * a.size; da.det = t.det; da.npix = npix; da.nseg = t.nseg;
        da.nframes = nframes; da.interp = interp; da.m = a.m_bins;
        const int lanes = (!interp && t.det == 1) ? t.nseg : npix;
        hipLaunchKernelGGL(ana_detect_kernel, dim3((unsigned)((lanes + 255) / 256), (unsigned)nframes, (unsigned)a.ndisp), dim3(256), 0, a.stream, da);
        QH_HIP(hipGetLastError());
    }
    for (int o = 0; o < a.num_pixout; o++) {
        DetTables *t = nullptr;
        for (int s = 0; s < kMaxPixouts; s++) if (a.dets[s].det == a.det_type[o]) { t = &a.dets[s]; break; }
        if (!t) return set_error(QH_ERR_INVALID, "analyzer: detector tables missing");
        QH_HIP(a.rows[o].ensure((size_t)a.ndisp * nframes * npix));
        QH_HIP(a.latest[o].ensure((size_t)a.ndisp * kMaxPixels));
        if (a.av_sum[o].cap == 0) if (int e = reset_average(a, o)) return e;
        if (a.av_mode[o] == 2 && a.av_buff[o].cap < (size_t)a.ndisp * kMaxAverage * kMaxPixels) {
            QH_HIP(a.av_buff[o].ensure((size_t)a.ndisp * kMaxAverage * kMaxPixels));
            QH_HIP(hipMemsetAsync(a.av_buff[o].p, 0, a.av_buff[o].cap * sizeof(double), a.stream));
        }
        AnaAvgArgs aa{};
        aa.tframes = t->tframes.p; aa.valid = t->valid.p; aa.t_persist = t->t_persist.p; aa.av_sum = a.av_sum[o].p; aa.av_buff = a.av_buff[o].p;
        aa.cd = a.d_cd.p; aa.rows = a.rows[o].p; aa.latest = a.latest[o].p; aa.scale = a.scale; aa.back = a.av_backmult[o];
        aa.norm_oneHz = a.norm_oneHz; aa.mode = a.av_mode[o]; aa.npix = npix; aa.nframes = nframes; aa.avail = a.avail_frames[o];
        aa.num_average = a.num_average[o]; aa.in_idx = a.av_in_idx[o]; aa.out_idx = a.av_out_idx[o]; aa.normalize = a.normalize[o];
        hipLaunchKernelGGL(ana_average_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)a.ndisp), dim3(256), 0, a.stream, aa);
        QH_HIP(hipGetLastError());
        if (a.av_mode[o] == 2)
            for (int f = 0; f < nframes; f++) {         // the counters of avenger's case 2, analyzer.c:508-531
                if (a.avail_frames[o] < a.num_average[o]) a.avail_frames[o]++;
                else if (++a.av_out_idx[o] == kMaxAverage) a.av_out_idx[o] = 0;
                if (++a.av_in_idx[o] == kMaxAverage) a.av_in_idx[o] = 0;
            }
        a.fresh[o].assign((size_t)a.ndisp, 1);
    }
    a.rows_frames = nframes;
    a.frames_total += nframes;
    a.pending.clear();
    return QH_OK;
}

// the dispatcher's scan (sendbuf, analyzer.c:884-917) until nothing more can start; complete sets go to `pending`
void dispatch(qh_ana &a)
{
    bool started = true;
    while (started) {
        started = false;
        for (int ss = 0; ss < a.num_stitch; ss++)
            if (!a.busy[ss] && a.ready[ss]) {
                a.busy[ss] = true;
                a.start_pos[ss] = a.out_pos[ss];
                a.out_pos[ss] += a.incr;
                if ((a.have[ss] -= a.incr) < a.size) a.ready[ss] = false;
                a.stitch_flag |= 1u << ss;
                if (a.stitch_flag == (1u << a.num_stitch) - 1u) {
                    a.stitch_flag = 0;
                    for (int s = 0; s < kMaxStitch; s++) a.busy[s] = false;
                    for (int s = 0; s < kMaxStitch; s++) a.pending.push_back(s < a.num_stitch ? (int)(a.start_pos[s] - a.base_pos[s]) : 0);
                }
                started = true;
            }
    }
}

// n = k * buff_size new samples of sub-span ss for every display: device (I, Q) doubles
int push(qh_ana &a, int ss, const double2 *d_src, long long src_stride, int n, int swap_iq)
{
    const int from = a.cur[ss], to = from ^ 1;
    // samples a frame may still need: from the next frame's start, or from the start of a frame of this sub-span that
    // waits for the other sub-spans of its set (the reference's worker has transformed it already; here the whole set
    // is transformed together)
    long long keep_from = a.out_pos[ss];
    if (a.busy[ss] && a.start_pos[ss] < keep_from) keep_from = a.start_pos[ss];
    if (keep_from > a.in_pos[ss]) keep_from = a.in_pos[ss];
    const long long drop = keep_from - a.base_pos[ss];
    const int keep = (int)(a.in_pos[ss] - keep_from);
    const long long need = (long long)keep + n;
    long long cap = a.stride[ss][to];
    if (cap < need) {
        cap = need + a.size;
        QH_HIP(a.sbuf[ss][to].ensure((size_t)(cap * a.ndisp)));
        a.stride[ss][to] = cap;
    }
    hipLaunchKernelGGL(ana_append_kernel, dim3((unsigned)((need + 255) / 256), (unsigned)a.ndisp), dim3(256), 0, a.stream,
                       (const float2 *)a.sbuf[ss][from].p, a.stride[ss][from], (int)drop, keep, d_src, src_stride, n, swap_iq, a.sbuf[ss][to].p, cap);
    QH_HIP(hipGetLastError());
    a.cur[ss] = to;
    a.base_pos[ss] = keep_from;
    for (int done = 0; done < n; done += a.buff_size) {                         // CloseBuffer's bookkeeping, analyzer.c:1422-1448
        if (a.have[ss] > a.max_writeahead) {
            a.out_pos[ss] += a.have[ss] - a.max_writeahead;
            a.have[ss] = a.max_writeahead;
        }
        if ((a.have[ss] += a.buff_size) >= a.size) a.ready[ss] = true;
        a.in_pos[ss] += a.buff_size;
        dispatch(a);
    }
    return QH_OK;
}

int check_config(qh_ana *h, const char *who)
{
    if (!h) return set_error(QH_ERR_INVALID, "%s: null analyzer", who);
    if (!h->configured) return set_error(QH_ERR_INVALID, "%s: SetAnalyzer has not run", who);
    return QH_OK;
}

}  // namespace

extern "C" {

qh_ana *qh_ana_create(int device, int ndisp, int max_size, int max_stitch, void *stream)
{
    if (ndisp <= 0 || max_size < 512 || max_stitch < 1 || max_stitch > kMaxStitch) {
        set_error(QH_ERR_INVALID, "qh_ana_create: bad arguments (ndisp > 0, max_size >= 512, 1 <= max_stitch <= 4)");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    qh_ana *h = new qh_ana();
    h->device = device; h->ndisp = ndisp; h->max_size = max_size; h->max_stitch = max_stitch;
    h->h_cd.assign(kMaxPixels, 1.0);
    if (hipSetDevice(device) != hipSuccess) { set_error(QH_ERR_HIP, "hipSetDevice failed"); delete h; return nullptr; }
    hipStream_t s = (hipStream_t)stream;
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { set_error(QH_ERR_HIP, "stream creation failed"); delete h; return nullptr; }
        h->own_stream = true;
    }
    h->stream = s;
    return h;
}

void qh_ana_destroy(qh_ana *h) { delete h; }

// SetAnalyzer, analyzer.c:999-1137
int qh_ana_set_analyzer(qh_ana *h, int n_pixout, int n_fft, int typ, const int *flp, int sz, int bf_sz, int win_type, double pi, int ovrlp, int clp,
                        double fscLin, double fscHin, int n_pix, int n_stch, int calset, double fmin, double fmax, int max_w)
{
    if (!h) return set_error(QH_ERR_INVALID, "SetAnalyzer: null analyzer");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    qh_ana &a = *h;
    if (n_fft != 1) return set_error(QH_ERR_INVALID, "SetAnalyzer: one LO per sub-span (dMAX_NUM_FFT is 1, wdsp/comm.h:125)");
    if (n_pixout < 1 || n_pixout > kMaxPixouts || (typ != 0 && typ != 1) || n_stch < 1 || n_stch > a.max_stitch || n_pix < 2 || n_pix > kMaxPixels ||
        win_type < 0 || win_type > 6 || ovrlp < 0 || ovrlp >= sz || clp < 0 || bf_sz < 1 || calset < 0 || calset >= kMaxCalSets || !flp)
        return set_error(QH_ERR_INVALID, "SetAnalyzer: argument out of range");
    // the transform: sz = R * M, M a power of two in 512 .. 4096 (8192 for the largest sizes), R a power of two up to 64
    if (sz < 512 || sz > a.max_size || (sz & (sz - 1))) return set_error(QH_ERR_UNSUPPORTED, "SetAnalyzer: fft size %d (powers of two from 512 to max_size)", sz);
    int M = sz, R = 1;
    while (M > 4096) { M >>= 1; R <<= 1; }              // 4096-point tiles: four workgroups per CU (the 8192-point transform holds one)
    if (R > kMaxR) { M <<= 1; R >>= 1; }
    if (R > kMaxR) return set_error(QH_ERR_UNSUPPORTED, "SetAnalyzer: fft size %d above %d", sz, 8192 * kMaxR);
    if ((a.max_size * 2) % bf_sz) return set_error(QH_ERR_INVALID, "SetAnalyzer: buff_size must divide the sample buffer (2 * max_size)");
    QH_HIP(hipSetDevice(a.device));
    a.num_pixout = n_pixout; a.type = typ; a.buff_size = bf_sz; a.flip = flp[0]; a.overlap = ovrlp; a.clip = clp;
    a.fsclipL = fscLin; a.fsclipH = fscHin; a.num_stitch = n_stch;
    if (sz != a.size || win_type != a.window_type || pi != a.PiAlpha) new_window(a, win_type, sz, pi);
    if (M != a.M) {
        const std::vector<cd> tw = fft_twiddle_table(M);
        std::vector<double2> t2(tw.size());
        for (size_t i = 0; i < tw.size(); i++) t2[i] = make_double2(tw[i].real(), tw[i].imag());
        QH_HIP(a.d_tw.upload(t2, a.stream));
        QH_HIP(hipStreamSynchronize(a.stream));
    }
    a.size = sz; a.M = M; a.R = R; a.window_type = win_type; a.PiAlpha = pi; a.max_writeahead = max_w;
    a.norm_oneHz = 10.0 * host_mlog10(1.0 / ((double)a.sample_rate / (double)a.size));         // CalcBandwidthNormalization, :919-924
    if ((fmin != a.f_min || fmax != a.f_max) && fmin == 0.0 && fmax == 0.0) std::fill(a.h_cd.begin(), a.h_cd.end(), 1.0);
    if ((fmax != 0.0 || fmin != 0.0) && (n_pix != a.num_pixels || fmin != a.f_min || fmax != a.f_max || calset != a.cal_set || a.cal_changed)) {
        if (a.n_freqs[calset] < 2) return set_error(QH_ERR_INVALID, "SetAnalyzer: calibration set %d has no table", calset);
        interpolate_cal(a, calset, fmin, fmax, n_pix);
    }
    a.incr = a.size - a.overlap;
    a.num_pixels = n_pix; a.f_min = fmin; a.f_max = fmax; a.cal_set = calset; a.cal_changed = 0;
    if (a.type == 0) { a.out_size = a.size / 2 + 1; a.scale = 4.0 / ((double)a.size * (double)a.size); }
    else { a.out_size = a.size; a.scale = 1.0 / ((double)a.size * (double)a.size); }
    const int span = a.out_size - 1 - 2 * a.clip;
    if (span < 1) return set_error(QH_ERR_INVALID, "SetAnalyzer: clip leaves no bins");
    a.begin_ss = 0; a.end_ss = a.num_stitch - 1;
    a.fscL = (int)a.fsclipL; a.fscH = (int)a.fsclipH;
    while (a.fscL >= span && a.begin_ss < a.num_stitch) { a.fscL -= span; a.begin_ss++; }
    while (a.fscH >= span && a.end_ss >= 0) { a.fscH -= span; a.end_ss--; }
    if (a.begin_ss > a.end_ss) return set_error(QH_ERR_INVALID, "SetAnalyzer: the span clips remove every sub-span");
    a.pix_per_bin = (double)a.num_pixels / ((double)(a.num_stitch * span) - a.fsclipL - a.fsclipH - 1.0);
    a.det_offset = -a.pix_per_bin * (a.fsclipL - std::floor(a.fsclipL));
    a.bin_per_pix = ((double)(a.num_stitch * span) - 1.0 - a.fsclipL - a.fsclipH) / ((double)a.num_pixels - 1.0);
    // the rings start over (analyzer.c:1102-1131)
    a.stitch_flag = 0;
    for (int s = 0; s < kMaxStitch; s++) {
        a.busy[s] = a.ready[s] = false; a.have[s] = 0;
        a.base_pos[s] = a.in_pos[s] = a.out_pos[s] = a.start_pos[s] = 0;
    }
    a.pending.clear();
    for (int o = 0; o < kMaxPixouts; o++) a.fresh[o].assign((size_t)a.ndisp, 0);
    a.tables_dirty = true;
    a.configured = true;
    return build_tables(a);
}

// SetCalibration, analyzer.c:1380-1410 with build_interpolants :800-882 (dMAX_M = 1): rows of (frequency, value)
int qh_ana_set_calibration(qh_ana *h, int set_num, int n_points, const double *cal)
{
    if (!h || set_num < 0 || set_num >= kMaxCalSets || n_points < 3 || n_points > kMaxN || !cal)
        return set_error(QH_ERR_INVALID, "SetCalibration: bad arguments (3 .. 100 points)");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    qh_ana &a = *h;
    std::vector<std::pair<double, double>> rows((size_t)n_points);
    for (int i = 0; i < n_points; i++) rows[(size_t)i] = {cal[2 * i], cal[2 * i + 1]};
    std::stable_sort(rows.begin(), rows.end(), [](const std::pair<double, double> &p, const std::pair<double, double> &q) { return p.first < q.first; });
    std::vector<double> x, y;
    for (int i = 0; i < n_points; i++)
        if (i == n_points - 1 || rows[(size_t)i].first != rows[(size_t)i + 1].first) { x.push_back(rows[(size_t)i].first); y.push_back(rows[(size_t)i].second); }
    const int n = (int)x.size();
    if (n < 3) return set_error(QH_ERR_INVALID, "SetCalibration: fewer than three distinct frequencies");
    std::vector<double> dx((size_t)n), idx((size_t)n), dmain((size_t)n), dsub((size_t)n), dsup((size_t)n), d((size_t)n), S((size_t)n), b((size_t)n), v((size_t)n);
    for (int i = 0; i < n - 1; i++) {
        dx[(size_t)i] = x[(size_t)i + 1] - x[(size_t)i];
        if (dx[(size_t)i] < 1e-30) return set_error(QH_ERR_INVALID, "SetCalibration: frequencies too close");
        idx[(size_t)i] = 1.0 / dx[(size_t)i];
    }
    for (int i = 1; i <= n - 2; i++) {
        const size_t k = (size_t)i;
        if (i == 1) { dsub[k] = 0.0; dmain[k] = 3.0 * dx[k - 1] + 2.0 * dx[k]; dsup[k] = dx[k]; }
        else if (i == n - 2) { dsub[k] = dx[k - 1]; dmain[k] = 2.0 * dx[k - 1] + 3.0 * dx[k]; dsup[k] = 0.0; }
        else { dsub[k] = dx[k - 1]; dmain[k] = 2.0 * (dx[k - 1] + dx[k]); dsup[k] = dx[k]; }
        d[k] = 6.0 * ((y[k + 1] - y[k]) * idx[k] - (y[k] - y[k - 1]) * idx[k - 1]);
    }
    b[1] = dmain[1]; v[1] = d[1];
    for (int i = 2; i <= n - 2; i++) {
        const size_t k = (size_t)i;
        const double t = dsub[k] / b[k - 1];
        b[k] = dmain[k] - t * dsup[k - 1];
        v[k] = d[k] - t * v[k - 1];
    }
    S[(size_t)n - 2] = v[(size_t)n - 2] / b[(size_t)n - 2];
    for (int i = n - 3; i >= 1; i--) S[(size_t)i] = (v[(size_t)i] - dsup[(size_t)i] * S[(size_t)i + 1]) / b[(size_t)i];
    S[0] = S[1]; S[(size_t)n - 1] = S[(size_t)n - 2];
    a.freqs[set_num] = x;
    a.ac3[set_num].assign((size_t)n, 0.0); a.ac2[set_num].assign((size_t)n, 0.0); a.ac1[set_num].assign((size_t)n, 0.0); a.ac0[set_num].assign((size_t)n, 0.0);
    for (int i = 0; i < n - 1; i++) {
        const size_t k = (size_t)i;
        a.ac3[set_num][k] = (S[k + 1] - S[k]) / (6.0 * dx[k]);
        a.ac2[set_num][k] = 0.5 * S[k];
        a.ac1[set_num][k] = (y[k + 1] - y[k]) * idx[k] - (2.0 * dx[k] * S[k] + dx[k] * S[k + 1]) / 6.0;
        a.ac0[set_num][k] = y[k];
    }
    a.n_freqs[set_num] = n;
    a.cal_changed = 1;
    return QH_OK;
}

int qh_ana_set_detector_mode(qh_ana *h, int pixout, int mode)          // SetDisplayDetectorMode, analyzer.c:1582
{
    if (!h || pixout < 0 || pixout >= kMaxPixouts || mode < 0 || mode > 4) return set_error(QH_ERR_INVALID, "SetDisplayDetectorMode: bad arguments");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->det_type[pixout] != mode) { h->det_type[pixout] = mode; h->tables_dirty = true; }
    return QH_OK;
}

int qh_ana_set_average_mode(qh_ana *h, int pixout, int mode)           // SetDisplayAverageMode, analyzer.c:1594
{
    if (!h || pixout < 0 || pixout >= kMaxPixouts || mode < -1 || mode > 3) return set_error(QH_ERR_INVALID, "SetDisplayAverageMode: bad arguments");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->av_mode[pixout] == mode && h->av_sum[pixout].cap) return QH_OK;
    QH_HIP(hipSetDevice(h->device));
    h->av_mode[pixout] = mode;
    return reset_average(*h, pixout);
}

int qh_ana_set_num_average(qh_ana *h, int pixout, int num)             // SetDisplayNumAverage, analyzer.c:1626
{
    if (!h || pixout < 0 || pixout >= kMaxPixouts || num < 1 || num > kMaxAverage) return set_error(QH_ERR_INVALID, "SetDisplayNumAverage: bad arguments (1 .. 60)");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->num_average[pixout] != num) { h->num_average[pixout] = num; h->avail_frames[pixout] = h->av_in_idx[pixout] = h->av_out_idx[pixout] = 0; }
    return QH_OK;
}

int qh_ana_set_av_backmult(qh_ana *h, int pixout, double mult)         // SetDisplayAvBackmult, analyzer.c:1641
{
    if (!h || pixout < 0 || pixout >= kMaxPixouts) return set_error(QH_ERR_INVALID, "SetDisplayAvBackmult: bad arguments");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    h->av_backmult[pixout] = mult;
    return QH_OK;
}

int qh_ana_set_sample_rate(qh_ana *h, int rate)                        // SetDisplaySampleRate, analyzer.c:1653
{
    if (!h || rate <= 0) return set_error(QH_ERR_INVALID, "SetDisplaySampleRate: bad arguments");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    h->sample_rate = rate;
    if (h->size > 0) h->norm_oneHz = 10.0 * host_mlog10(1.0 / ((double)rate / (double)h->size));
    return QH_OK;
}

int qh_ana_set_norm_onehz(qh_ana *h, int pixout, int norm)             // SetDisplayNormOneHz, analyzer.c:1666
{
    if (!h || pixout < 0 || pixout >= kMaxPixouts) return set_error(QH_ERR_INVALID, "SetDisplayNormOneHz: bad arguments");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    h->normalize[pixout] = norm ? 1 : 0;
    return QH_OK;
}

double qh_ana_get_enb(qh_ana *h) { return h ? 1.0 / h->inv_enb : 0.0; }  // GetDisplayENB, analyzer.c:1678

// ResetPixelBuffers, analyzer.c:927-996
int qh_ana_reset_pixel_buffers(qh_ana *h)
{
    if (!h) return set_error(QH_ERR_INVALID, "ResetPixelBuffers: null analyzer");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    qh_ana &a = *h;
    QH_HIP(hipSetDevice(a.device));
    for (int s = 0; s < kMaxPixouts; s++)
        if (a.dets[s].t_persist.cap) QH_HIP(hipMemsetAsync(a.dets[s].t_persist.p, 0, a.dets[s].t_persist.cap * sizeof(double), a.stream));
    for (int o = 0; o < kMaxPixouts; o++) {
        if (a.av_sum[o].cap) {
            const int mode = a.av_mode[o];
            if (mode != 2) if (int e = fill(a, a.av_sum[o].p, a.av_sum[o].cap, mode == 1 ? 1.0e-12 : mode == 3 ? -160.0 : 0.0)) return e;
        }
        if (a.av_buff[o].cap) QH_HIP(hipMemsetAsync(a.av_buff[o].p, 0, a.av_buff[o].cap * sizeof(double), a.stream));
        a.avail_frames[o] = a.av_in_idx[o] = a.av_out_idx[o] = 0;
        a.fresh[o].assign((size_t)a.ndisp, 0);
    }
    a.stitch_flag = 0;
    for (int s = 0; s < kMaxStitch; s++) {
        a.busy[s] = a.ready[s] = false; a.have[s] = 0;
        a.base_pos[s] = a.in_pos[s] = a.out_pos[s] = a.start_pos[s] = 0;
    }
    a.pending.clear();
    return QH_OK;
}

// Feed n = k * buff_size samples to sub-span ss of every display ((I, Q) doubles on the device, [ndisp][stride]) and run
// every frame that completes.  *frames = frames published by this call.
int qh_ana_feed(qh_ana *h, int ss, const void *d_iq, long long disp_stride, int n, int *frames)
{
    if (int e = check_config(h, "qh_ana_feed")) return e;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    qh_ana &a = *h;
    if (frames) *frames = 0;
    if (ss < 0 || ss >= a.num_stitch || n < 0 || (n > 0 && (!d_iq || disp_stride < n)) || n % a.buff_size)
        return set_error(QH_ERR_INVALID, "qh_ana_feed: bad arguments (n must be a multiple of buff_size)");
    if (n == 0) return QH_OK;
    QH_HIP(hipSetDevice(a.device));
    if (int e = push(a, ss, (const double2 *)d_iq, disp_stride, n, 0)) return e;
    const int nf = (int)a.pending.size() / kMaxStitch;
    if (int e = run_pending(a)) return e;
    if (frames) *frames = nf;
    return QH_OK;
}

// the same from host memory; swap_iq = 1 for Spectrum0's (Q, I) pair order (analyzer.c:1550-1553)
int qh_ana_feed_host(qh_ana *h, int ss, const double *h_iq, long long disp_stride, int n, int swap_iq, int *frames)
{
    if (int e = check_config(h, "qh_ana_feed_host")) return e;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    qh_ana &a = *h;
    if (frames) *frames = 0;
    if (ss < 0 || ss >= a.num_stitch || n < 0 || (n > 0 && (!h_iq || disp_stride < n)) || n % a.buff_size)
        return set_error(QH_ERR_INVALID, "qh_ana_feed_host: bad arguments (n must be a multiple of buff_size)");
    if (n == 0) return QH_OK;
    QH_HIP(hipSetDevice(a.device));
    QH_HIP(a.staging.ensure((size_t)a.ndisp * n));
    QH_HIP(hipMemcpy2DAsync(a.staging.p, (size_t)n * 16, h_iq, (size_t)disp_stride * 16, (size_t)n * 16, (size_t)a.ndisp, hipMemcpyHostToDevice, a.stream));
    QH_HIP(hipStreamSynchronize(a.stream));             // the caller's buffer is free again when this returns
    if (int e = push(a, ss, a.staging.p, n, n, swap_iq)) return e;
    const int nf = (int)a.pending.size() / kMaxStitch;
    if (int e = run_pending(a)) return e;
    if (frames) *frames = nf;
    return QH_OK;
}

// SnapSpectrum, analyzer.c:1337-1367: the next frame's transform of (display, sub-span) -- size complex values, fft-shifted (the second
// half of fft_out first, analyzer.c:710-711).  qh_ana_snap_arm asks for it, the feed call that completes that frame takes it,
// qh_ana_snap_take hands it over (*flag = 0: not there yet); qh_ana_snap_wait blocks like the reference (another thread feeds),
// timeout_ms < 0 = for ever.
int qh_ana_snap_arm(qh_ana *h, int disp, int ss)
{
    if (int e = check_config(h, "SnapSpectrum")) return e;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (disp < 0 || disp >= h->ndisp || ss < 0 || ss >= h->num_stitch) return set_error(QH_ERR_INVALID, "SnapSpectrum: bad display or sub-span");
    if (ss < h->begin_ss || ss > h->end_ss) return set_error(QH_ERR_UNSUPPORTED, "SnapSpectrum: sub-span %d is outside the displayed range, no transform is made of it", ss);
    std::lock_guard<std::mutex> sl(h->snap_mu);
    h->snap_armed = true; h->snap_done = false; h->snap_disp = disp; h->snap_ss = ss;
    return QH_OK;
}
int qh_ana_snap_take(qh_ana *h, double *snap_buff, int *flag)
{
    if (!h || !snap_buff || !flag) return set_error(QH_ERR_INVALID, "qh_ana_snap_take: bad arguments");
    std::lock_guard<std::mutex> sl(h->snap_mu);
    *flag = 0;
    if (!h->snap_done) return QH_OK;
    std::copy(h->snap_host.begin(), h->snap_host.end(), snap_buff);
    h->snap_done = false;
    *flag = 1;
    return QH_OK;
}
int qh_ana_snap_wait(qh_ana *h, double *snap_buff, int timeout_ms, int *flag)
{
    if (!h || !snap_buff) return set_error(QH_ERR_INVALID, "qh_ana_snap_wait: bad arguments");
    std::unique_lock<std::mutex> sl(h->snap_mu);
    bool ok = true;
    if (timeout_ms < 0) h->snap_cv.wait(sl, [&] { return h->snap_done; });
    else ok = h->snap_cv.wait_for(sl, std::chrono::milliseconds(timeout_ms), [&] { return h->snap_done; });
    if (ok) { std::copy(h->snap_host.begin(), h->snap_host.end(), snap_buff); h->snap_done = false; }
    else h->snap_armed = false;                              // SnapSpectrumTimeout resets the request (analyzer.c:1364)
    if (flag) *flag = ok ? 1 : 0;
    return QH_OK;
}

// GetPixels, analyzer.c:1315-1334: the newest row of one display if it has not been read yet
int qh_ana_get_pixels(qh_ana *h, int disp, int pixout, float *pix, int *flag)
{
    if (int e = check_config(h, "GetPixels")) return e;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    qh_ana &a = *h;
    if (disp < 0 || disp >= a.ndisp || pixout < 0 || pixout >= a.num_pixout || !pix || !flag) return set_error(QH_ERR_INVALID, "GetPixels: bad arguments");
    *flag = 0;
    if (a.fresh[pixout].size() != (size_t)a.ndisp || !a.fresh[pixout][(size_t)disp]) return QH_OK;
    QH_HIP(hipSetDevice(a.device));
    QH_HIP(hipMemcpyAsync(pix, a.latest[pixout].p + (size_t)disp * kMaxPixels, (size_t)a.num_pixels * sizeof(float), hipMemcpyDeviceToHost, a.stream));
    QH_HIP(hipStreamSynchronize(a.stream));
    a.fresh[pixout][(size_t)disp] = 0;
    *flag = 1;
    return QH_OK;
}

// every row the last feed call produced: rows[ndisp][frames][num_pixels] floats on the device
int qh_ana_rows(qh_ana *h, int pixout, const float **d_rows, int *frames, int *num_pixels)
{
    if (int e = check_config(h, "qh_ana_rows")) return e;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (pixout < 0 || pixout >= h->num_pixout || !d_rows || !frames || !num_pixels) return set_error(QH_ERR_INVALID, "qh_ana_rows: bad arguments");
    *d_rows = h->rows[pixout].p; *frames = h->rows_frames; *num_pixels = h->num_pixels;
    return QH_OK;
}

int qh_ana_rows_host(qh_ana *h, int pixout, float *out, int max_frames, int *frames)
{
    if (int e = check_config(h, "qh_ana_rows_host")) return e;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    qh_ana &a = *h;
    if (pixout < 0 || pixout >= a.num_pixout || !out || !frames) return set_error(QH_ERR_INVALID, "qh_ana_rows_host: bad arguments");
    *frames = a.rows_frames;
    if (a.rows_frames > max_frames) return set_error(QH_ERR_INVALID, "qh_ana_rows_host: %d frames, room for %d", a.rows_frames, max_frames);
    if (a.rows_frames == 0) return QH_OK;
    QH_HIP(hipSetDevice(a.device));
    QH_HIP(hipMemcpyAsync(out, a.rows[pixout].p, (size_t)a.ndisp * a.rows_frames * a.num_pixels * sizeof(float), hipMemcpyDeviceToHost, a.stream));
    QH_HIP(hipStreamSynchronize(a.stream));
    return QH_OK;
}

void *qh_ana_stream(qh_ana *h) { return h ? (void *)h->stream : nullptr; }
long long qh_ana_frames(qh_ana *h) { return h ? h->frames_total : 0; }
int qh_ana_buff_size(qh_ana *h) { return h ? h->buff_size : 0; }
int qh_ana_num_pixels(qh_ana *h) { return h ? h->num_pixels : 0; }


// ---- WDSP's own names (wdsp/analyzer.h:100-193): one bank of a single display per display id ------------------------
namespace {
constexpr int kMaxDisplays = 64;        // dMAX_DISPLAYS, comm.h:123
qh_ana *g_disp[kMaxDisplays];
std::mutex g_disp_mu;
qh_ana *disp_of(int disp, const char *who)
{
    std::lock_guard<std::mutex> lk(g_disp_mu);
    if (disp < 0 || disp >= kMaxDisplays || !g_disp[disp]) { set_error(QH_ERR_INVALID, "%s: display %d does not exist", who, disp); return nullptr; }
    return g_disp[disp];
}
}  // namespace

void XCreateAnalyzer(int disp, int *success, int m_size, int m_LO, int m_stitch, char *app_data_path)      // analyzer.c:1140
{
    (void)app_data_path;
    if (success) *success = -1;
    if (disp < 0 || disp >= kMaxDisplays || m_LO != 1) { set_error(QH_ERR_INVALID, "XCreateAnalyzer: display id 0..63, one LO per sub-span"); return; }
    qh_ana *h = qh_ana_create(0, 1, m_size, m_stitch, nullptr);
    if (!h) return;
    std::lock_guard<std::mutex> lk(g_disp_mu);
    delete g_disp[disp];
    g_disp[disp] = h;
    if (success) *success = 0;
}

void DestroyAnalyzer(int disp)                          // analyzer.c:1239
{
    std::lock_guard<std::mutex> lk(g_disp_mu);
    if (disp < 0 || disp >= kMaxDisplays) return;
    delete g_disp[disp];
    g_disp[disp] = nullptr;
}

void SetAnalyzer(int disp, int n_pixout, int n_fft, int typ, int *flp, int sz, int bf_sz, int win_type, double pi, int ovrlp, int clp, double fscLin,
                 double fscHin, int n_pix, int n_stch, int calset, double fmin, double fmax, int max_w)
{
    if (qh_ana *h = disp_of(disp, "SetAnalyzer"))
        (void)qh_ana_set_analyzer(h, n_pixout, n_fft, typ, flp, sz, bf_sz, win_type, pi, ovrlp, clp, fscLin, fscHin, n_pix, n_stch, calset, fmin, fmax, max_w);
}

void SetCalibration(int disp, int set_num, int n_points, double (*cal)[2])     // dMAX_M + 1 = 2 columns
{
    if (qh_ana *h = disp_of(disp, "SetCalibration")) (void)qh_ana_set_calibration(h, set_num, n_points, cal ? &cal[0][0] : nullptr);
}

void Spectrum0(int run, int disp, int ss, int LO, double *pbuff)               // analyzer.c:1536: (Q, I) pairs of doubles
{
    (void)LO;
    if (!run) return;
    if (qh_ana *h = disp_of(disp, "Spectrum0")) (void)qh_ana_feed_host(h, ss, pbuff, qh_ana_buff_size(h), qh_ana_buff_size(h), 1, nullptr);
}

void Spectrum2(int run, int disp, int ss, int LO, float *pbuff)                // analyzer.c:1490: (Q, I) pairs of dINREAL
{
    (void)LO;
    if (!run) return;
    qh_ana *h = disp_of(disp, "Spectrum2");
    if (!h || !pbuff) return;
    const int n = qh_ana_buff_size(h);
    std::vector<double> tmp((size_t)2 * n);
    for (int i = 0; i < 2 * n; i++) tmp[(size_t)i] = (double)pbuff[i];
    (void)qh_ana_feed_host(h, ss, tmp.data(), n, n, 1, nullptr);
}

void Spectrum(int disp, int ss, int LO, float *pI, float *pQ)                  // analyzer.c:1451
{
    (void)LO;
    qh_ana *h = disp_of(disp, "Spectrum");
    if (!h || !pI || !pQ) return;
    const int n = qh_ana_buff_size(h);
    std::vector<double> tmp((size_t)2 * n);
    for (int i = 0; i < n; i++) { tmp[(size_t)2 * i] = (double)pI[i]; tmp[(size_t)2 * i + 1] = (double)pQ[i]; }
    (void)qh_ana_feed_host(h, ss, tmp.data(), n, n, 0, nullptr);
}

void OpenBuffer(int disp, int ss, int LO, void **Ipointer, void **Qpointer)    // analyzer.c:1412: room for buff_size samples
{
    (void)LO;
    qh_ana *h = disp_of(disp, "OpenBuffer");
    if (!h || ss < 0 || ss >= kMaxStitch || !Ipointer || !Qpointer) return;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    h->h_open_I[ss].resize((size_t)h->buff_size); h->h_open_Q[ss].resize((size_t)h->buff_size);
    *Ipointer = h->h_open_I[ss].data(); *Qpointer = h->h_open_Q[ss].data();
}

void CloseBuffer(int disp, int ss, int LO)                                     // analyzer.c:1422
{
    qh_ana *h = disp_of(disp, "CloseBuffer");
    if (!h || ss < 0 || ss >= kMaxStitch || (int)h->h_open_I[ss].size() != h->buff_size) return;
    Spectrum(disp, ss, LO, h->h_open_I[ss].data(), h->h_open_Q[ss].data());
}

void GetPixels(int disp, int pixout, float *pix, int *flag)                    // analyzer.c:1315
{
    if (flag) *flag = 0;
    if (qh_ana *h = disp_of(disp, "GetPixels")) (void)qh_ana_get_pixels(h, 0, pixout, pix, flag);
}

void SnapSpectrum(int disp, int ss, int LO, double *snap_buff)                 // analyzer.c:1337: blocks until the next frame (fed by another thread)
{
    (void)LO;
    qh_ana *h = disp_of(disp, "SnapSpectrum");
    if (h && snap_buff && qh_ana_snap_arm(h, 0, ss) == QH_OK) (void)qh_ana_snap_wait(h, snap_buff, -1, nullptr);
}
void SnapSpectrumTimeout(int disp, int ss, int LO, double *snap_buff, unsigned int timeout, int *flag)     // analyzer.c:1349
{
    (void)LO;
    if (flag) *flag = 0;
    qh_ana *h = disp_of(disp, "SnapSpectrumTimeout");
    if (h && snap_buff && qh_ana_snap_arm(h, 0, ss) == QH_OK) (void)qh_ana_snap_wait(h, snap_buff, (int)timeout, flag);
}

void ResetPixelBuffers(int disp) { if (qh_ana *h = disp_of(disp, "ResetPixelBuffers")) (void)qh_ana_reset_pixel_buffers(h); }
void SetDisplayDetectorMode(int disp, int pixout, int mode) { if (qh_ana *h = disp_of(disp, "SetDisplayDetectorMode")) (void)qh_ana_set_detector_mode(h, pixout, mode); }
void SetDisplayAverageMode(int disp, int pixout, int mode) { if (qh_ana *h = disp_of(disp, "SetDisplayAverageMode")) (void)qh_ana_set_average_mode(h, pixout, mode); }
void SetDisplayNumAverage(int disp, int pixout, int num) { if (qh_ana *h = disp_of(disp, "SetDisplayNumAverage")) (void)qh_ana_set_num_average(h, pixout, num); }
void SetDisplayAvBackmult(int disp, int pixout, double mult) { if (qh_ana *h = disp_of(disp, "SetDisplayAvBackmult")) (void)qh_ana_set_av_backmult(h, pixout, mult); }
void SetDisplaySampleRate(int disp, int rate) { if (qh_ana *h = disp_of(disp, "SetDisplaySampleRate")) (void)qh_ana_set_sample_rate(h, rate); }
void SetDisplayNormOneHz(int disp, int pixout, int norm) { if (qh_ana *h = disp_of(disp, "SetDisplayNormOneHz")) (void)qh_ana_set_norm_onehz(h, pixout, norm); }
double GetDisplayENB(int disp) { qh_ana *h = disp_of(disp, "GetDisplayENB"); return h ? qh_ana_get_enb(h) : 0.0; }

}  // extern "C"
