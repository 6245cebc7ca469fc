// qh_quisk_rx_compat.cpp -- the Quisk native block API for ONE receiver (include/quiskhip.h group 9).
//
// quisk.c keeps its receive path behind `int quisk_process_samples(complex double *cSamples, int nSamples)`
// (quisk.h:375, quisk.c:2289): in place, returns the output count at the playback rate, parameters arrive through
// globals that the GUI thread sets with set_tune / set_rx_mode / set_filters / set_agc ... and reads with
// get_filter_rate / get_graph.  This layer offers exactly that shape -- one process-wide receiver, the same call names
// with a qh_quisk_ prefix -- and restates the WHOLE function (quisk.c:2289-2742) in its order:
//
//   key-down replacement (host: nothing is demodulated)             quisk.c:2368-2433
//   AddTestTone, spectrum inversion                                  quisk.c:1258-1303,2438-2446
//   NoiseBlanker (not while the key is down)                         quisk.c:2448-2449
//   FFT ring producer                                                quisk.c:2454-2475
//   tune + quisk_process_decimate + quisk_process_demodulate, bank 0 quisk.c:2477-2530
//   measure_freq on the decimated samples                            quisk.c:2527-2528,5579-5649
//   second channel on bank 1: split Rx/Tx or the played sub-receiver, Buffer2Chan   quisk.c:2537-2621,1577-1611
//   sub-receiver 1 on a digital output device, bank 2, Agc3          quisk.c:2630-2651
//   cFracDecim to 48 ksps                                            quisk.c:2654-2659,622-665
//   wdspFexchange0                                                   quisk.c:2660-2661
//   HB45 interpolation x2 / x4 / x8 to the playback rate             quisk.c:2663-2682
//   process_agc (Agc1, Agc2), kill_audio / squelch, key-up envelope  quisk.c:2686-2738
//
// The block is uploaded ONCE and stays on the device until the playback samples come back: every step above is a kernel
// (or one of the library's engines) on one HIP stream -- the optional WDSP hand-off too: the shim's ring and fexchange0's rings are
// device FIFOs in front of the RXA engine (qh_wdsp_fexchange0_device), enqueued behind this stream.
// Mode, bandwidth class or rate changes rebuild a bank (filter histories restart: a few ms of transient where the
// reference keeps its static histories); the tuning oscillators keep their phase (one per purpose, as the reference's
// rxTuneVector / txTuneVector / aux1TuneVector / aux2TuneVector).
#include <cmath>
#include <complex>
#include <cstring>
#include <mutex>
#include <vector>
#include "qh_internal.hpp"
#include "qh_ps_kernels.hpp"      // the steps shared with the receiver bank of the whole function (qh_qps.hip): this file runs them with nch = 1


namespace {

using u64 = unsigned long long;
constexpr int kBuf2Chan = 12000;            // BUF2CHAN_SIZE, quisk.c:1576
constexpr int kMfSize = 12000;              // measure_freq's fft_size, quisk.c:5586

// The stereo join of the two demodulated streams (quisk.c:2548-2620) behind Buffer2Chan (quisk.c:1577-1611).  Stream k is the
// concatenation of what Buffer2Chan held back for it (buf_k, nbuf_k samples) and the real parts of this call's bank output.
// sel_re / sel_im: which stream goes to the real / imaginary output (0 = bank 0, 1 = bank 1).
__device__ __forceinline__ double b2c_at(const double *buf, int nbuf, const double2 *samp, int i)
{
    return i < nbuf ? buf[i] : samp[i - nbuf].x;
}
__global__ void ps_join_kernel(const double *buf1, int nbuf1, const double2 *s1, const double *buf2, int nbuf2, const double2 *s2,
                               int nout, int sel_re, int sel_im, double2 *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nout) return;
    const double a = b2c_at(buf1, nbuf1, s1, i), b = b2c_at(buf2, nbuf2, s2, i);
    out[i] = make_double2(sel_re ? b : a, sel_im ? b : a);
}
// what stays in Buffer2Chan's buffers: samples nout .. total_k - 1 of either stream
__global__ void ps_b2c_keep_kernel(const double *buf1, int nbuf1, const double2 *s1, int keep1, double *new1, const double *buf2, int nbuf2,
                                   const double2 *s2, int keep2, double *new2, int nout)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < keep1) new1[i] = b2c_at(buf1, nbuf1, s1, nout + i);
    if (i < keep2) new2[i] = b2c_at(buf2, nbuf2, s2, nout + i);
}

// the two output channels as two real streams for Agc1 / Agc2 and back (quisk.c:2690-2698)
__global__ void ps_split_kernel(const double2 *x, int n, double2 *a, double2 *b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 v = x[i];
    a[i] = make_double2(v.x, 0.0); b[i] = make_double2(v.y, 0.0);
}
__global__ void ps_merge_kernel(const double2 *a, const double2 *b, int n, double2 *x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] = make_double2(a[i].x, b[i].x);
}
// ---- get_filter (quisk.c:5481-5568), the "RX Filter" screen's curve: three small kernels, run when the user opens that screen
// the multitone: 0.5 + sum_{f = 1 .. nf} cos(2 pi f t / W), the terms added in the reference's order (frequency by frequency)
__global__ void gf_multitone_kernel(double *x, int total, int W, int nf)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    double s = 0.5;
    for (int f = 1; f <= nf; f++) s += cospi(2.0 * (double)(((long long)f * t) % W) / (double)W);
    x[t] = s;
}
// its own copy of the cRxFilterOut loop (quisk.c:5516-5530): tap 0 on the newest sample, taps 1 .. N - 1 on the oldest .. second newest;
// output m is the one of time N + m, times fft_window[m] (record_app's window of fft_size points, quisk.c:6008, its FIRST W entries)
__global__ void gf_filter_kernel(const double *x, const double *fI, const double *fQ, int N, int W, int fft_size, double2 *y)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= W) return;
    const int t = N + m;
    double aI = 0.0, aQ = 0.0;
    if (N > 0) { aI = x[t] * fI[0]; aQ = x[t] * fQ[0]; }
    for (int k = 1; k < N; k++) { const double v = x[t - N + k]; aI += v * fI[k]; aQ += v * fQ[k]; }
    if (N == 0) { aI = aQ = 0.0; }
    const double w = 0.5 + 0.5 * cospi(2.0 * (double)(m - fft_size / 2) / (double)fft_size);
    y[m] = make_double2(aI * w, aQ * w);
}
// the W-point transform by its definition (W is the graph's width in pixels, any number; once per screen refresh), |X| / W in dB with
// the -140 dB floor, negative frequencies first
__global__ void gf_dft_kernel(const double2 *y, int W, double *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= W) return;
    double re = 0.0, im = 0.0;
    for (int n = 0; n < W; n++) {
        double s, c;
        sincospi(-2.0 * (double)(((long long)k * n) % W) / (double)W, &s, &c);
        re += y[n].x * c - y[n].y * s;
        im += y[n].x * s + y[n].y * c;
    }
    const double a = hypot(re, im) * (1.0 / W);
    const double db = a <= 1e-7 ? -140.0 : 20.0 * log10(a);
    const int half = W / 2;
    out[k >= half ? k - half : k + (W - half)] = db;
}

// the tune of quisk.c:2481-2487 on its own (mode EXT has no decimator behind it to carry the oscillator): x[i] *= e^{j 2 pi (ph + i dph)}
__global__ void ps_tune_kernel(const double2 *in, double2 *out, int n, unsigned long long ph, unsigned long long dph)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s, c;
    sincospi(2.0 * ((double)((ph + dph * (unsigned long long)i) >> 11) * (1.0 / 9007199254740992.0)), &s, &c);
    const double2 x = in[i];
    out[i] = make_double2(x.x * c - x.y * s, x.x * s + x.y * c);
}

inline unsigned grid_for(int n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

using qh_ps::turns_step;

template <typename T> struct DevBuf {           // grows on demand; contents are not kept across a growth
    T *p = nullptr;
    size_t cap = 0;
    int need(size_t n)
    {
        if (n <= cap) return QH_OK;
        if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 4 + 64;
        if (hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess) return qh::set_error(QH_ERR_HIP, "qh_quisk: hipMalloc of %zu bytes failed", want * sizeof(T));
        cap = want;
        return QH_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <typename T> struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    int need(size_t n)
    {
        if (n <= cap) return QH_OK;
        if (p) { (void)hipDeviceSynchronize(); (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 4 + 64;
        if (hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault) != hipSuccess) return qh::set_error(QH_ERR_HIP, "qh_quisk: pinned allocation failed");
        cap = want;
        return QH_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// one receiver bank of the reference (Storage[bank] of quisk_process_decimate / _demodulate and of c/dRxFilterOut)
struct Bank {
    qh_qrx *rx = nullptr;
    int mode = -1, cls = -1, tune = 0x7fffffff;
    int filt_n = -1;                    // which of cFilterI/Q[nFilter] its Rx filter stage holds ...
    long long filt_epoch = -1;          // ... as of which set_filters call
    int notch_applied = -1;
    int purpose = -1;                   // whose tune vector its oscillator carries (index into QuiskRx::phase)
    double sq_level = -1e300;           // squelch settings the bank has been given
    int ssb_en = -1, ssb_lv = -1, ssb_bw = -1;
};

struct QuiskRx {
    std::mutex mtx;
    int sample_rate = 0, playback_rate = 48000, mode = 3 /* USB */, tune = 0;
    double agc_gain = 80.0;                     // agcReleaseGain, quisk.c:191
    std::vector<double> tables[13];
    qh_qrx_tables t{};
    bool have_tables = false;
    hipStream_t stream = nullptr;
    // cFilterI / cFilterQ [nFilter], filter_bandwidth[nFilter] and the ONE sizeFilter of set_filters (quisk.c:127-129,4551-4594)
    std::vector<double> filtI[3], filtQ[3];
    int filt_bw[3] = { 2700, 2700, 2700 }, size_filter = 0;
    long long filt_epoch = 0;
    Bank bank[3];
    // the tune vectors (quisk.c:2308-2311) as oscillator phases: 0 rx, 1 tx (split), 2 aux1 (played sub-receiver), 3 aux2;
    // 4: the measurement receiver's own
    u64 phase[5] = { 0, 0, 0, 0, 0 };
    bool ssb_planned = false;                   // ssb_squelch's `plan` static: one for all banks (quisk.c:1091,1104)
    // panadapter (record_app's fft_size / data_width, quisk.c:5946)
    qh_pan *pan = nullptr;
    int fft_size = 0, data_width = 0;
    qh_nb *nb = nullptr;                        // NoiseBlanker's statics (quisk.c:682-687) outlive mode changes
    int nb_level = 0;
    int auto_notch = 0, rit_freq = 0;
    bool notch_reset = false;
    qh_qagc *agc[3] = { nullptr, nullptr, nullptr };        // Agc1, Agc2, Agc3 = {0.7, 0, 0} (quisk.c:2321): rate = playback rate
    double agc_gain_set[3] = { -1.0, -1.0, -1.0 };
    int tx_tune = 0, split_rxtx = 0;            // set_tune's second argument, set_split_rxtx (quisk.c:4702,4694)
    int play_channel = -1, play_method = 0;     // set_multirx_play_channel / _method (quisk.c:4856,4846)
    static constexpr int kMaxSub = 9;           // QUISK_MAX_SUB_RECEIVERS, quisk.h
    int sub_freq[kMaxSub] = { 0 }, sub_mode[kMaxSub] = { 0 };
    std::vector<double> sub_samples[kMaxSub];   // multirx_cSamples[] for the coming call
    int sub_have[kMaxSub] = { 0 };
    int multirx_count = 0, sub_rx1_driver = 0;
    int old_split = 0, old_play = 0;            // "static int old_multirx_play_channel = 0", quisk.c:2303
    // key handling (quisk.c:2368-2433)
    int key_down = 0, cw_key_down = 0, active_sidetone = 0, is_fdx = 0, kill_audio = 0, invert_spectrum = 0;
    int txrx_silence_ms = 50;
    double sidetone_volume = 0.0;
    std::complex<double> sidetone_phase{ 1.0, 0.0 }, sidetone_vec{ 2.2e9, 0.0 };
    double out_counter = 0.0, sidetone_env = 0.0, keyup_env = 1.0;
    int sidetone_on = 0, play_silence = 0;
    // squelches of quisk_process_demodulate (set_squelch quisk.c:4721, set_ssb_squelch quisk.c:4729)
    double squelch_level = -999.0;
    int ssb_squelch_enabled = 0, ssb_squelch_level = 0;
    // AddTestTone (quisk.c:1258-1303, add_tone quisk.c:3203)
    bool tone_on = false;
    u64 tone_phase = 0, tone_step = 0, audio_phase = 0;
    // Buffer2Chan (quisk.c:1577-1611): device buffers, counts on the host
    DevBuf<double> b2c[2][2];
    int b2c_cur = 0, nbuf1 = 0, nbuf2 = 0;
    // cFracDecim's statics (quisk.c:626-629)
    double fd_dindex = 1.0;
    double2 *fd_hist[2] = { nullptr, nullptr };
    int fd_cur = 0;
    // the interpolator to the playback rate (HalfBand7..9, quisk.c:2663-2682) as one polyphase filter
    qh_rat *up = nullptr;
    int up_ratio = 1;
    // measure_freq (quisk.c:5579-5649)
    int measure_mode = 0, mf_index = 0, mf_count = 0;
    double measured_frequency = 0.0;
    Bank mf_bank;
    qh_fir *mf_dec = nullptr;
    qh_pan *mf_fft = nullptr;
    std::vector<double> mf_avg;
    // work buffers
    DevBuf<double2> d_raw, d_x, d_nb, d_o0, d_o1, d_mix, d_fd, d_up, d_a, d_b, d_sub, d_sub0, d_s1, d_mf, d_mf8, d_wd;
    int *d_flags = nullptr;
    PinBuf<double> h_in, h_out, h_sub1;
    int *h_flags = nullptr;
    std::vector<double> sub1_out;               // sub-receiver 1's audio of the last call (what play_sound_interface got)
    int sub1_n = 0;
    int squelch_real = 0, squelch_imag = 0;
    // mode EXT (quisk.c:2490-2493): the user's quisk_extern_demod (extdemod.c:13), a host function by definition
    int (*ext_demod)(double *cSamples, int nSamples, double decim) = nullptr;
    DevBuf<double2> d_ext;
    long long failed_calls = 0;                 // calls whose device chain failed: the caller got 0 samples, which alone says nothing
};

QuiskRx g;

int bw_class(int mode, int bw)     // what of the bandwidth the bank's structure depends on (quisk.c:2089,2143)
{
    if (mode == 7 || mode == 8 || mode == 11 || mode == 12) return bw < 3000 ? 0 : 1;
    if (mode == 9) return bw < 19000 ? 0 : 1;
    return 0;
}
bool has_ssb_squelch(int mode) { return mode <= 4 || mode == 10; }       // CW, SSB, AM (IMD takes the SSB path)
bool is_fm_mode(int mode) { return mode == 5 || mode == 13; }

void destroy_bank(Bank &b, bool keep_phase)
{
    if (b.rx) {
        if (keep_phase && b.purpose >= 0) (void)qh_qrx_get_nco_phase(b.rx, 0, &g.phase[b.purpose]);
        qh_qrx_destroy(b.rx);
    }
    b = Bank();
}

// Bank `b` ready for this call: the receiver structure of (mode, bandwidth class), the purpose's oscillator phase, the tune
// frequency, cFilterI/Q[nFilter][0 .. sizeFilter) and the squelch settings.
int ensure_bank(Bank &b, int mode, int nFilter, int tune, int purpose, int bw_force = -1)
{
    if (mode == 6) return qh::set_error(QH_ERR_UNSUPPORTED, "mode EXT calls the user's quisk_extern_demod (extdemod.c:13): not provided");
    const int bw = bw_force >= 0 ? bw_force : g.filt_bw[nFilter];
    const int cls = bw_class(mode, bw);
    if (!b.rx || b.mode != mode || b.cls != cls) {
        destroy_bank(b, true);
        b.rx = qh_qrx_create_ex(0, 1, g.sample_rate, mode, bw, &g.t, g.stream);
        if (!b.rx) return QH_ERR_HIP;
        b.mode = mode; b.cls = cls;
        if (int rc = qh_qrx_set_mute_deferred(b.rx, 1)) return rc;
    }
    if (b.tune != tune) { if (int rc = qh_qrx_set_tune(b.rx, 0, tune)) return rc; b.tune = tune; }
    if (b.purpose != purpose) {
        if (b.purpose >= 0) if (int rc = qh_qrx_get_nco_phase(b.rx, 0, &g.phase[b.purpose])) return rc;
        if (int rc = qh_qrx_set_nco_phase(b.rx, 0, g.phase[purpose])) return rc;
        b.purpose = purpose;
    }
    if (b.filt_n != nFilter || b.filt_epoch != g.filt_epoch) {
        std::vector<double> fI((size_t)g.size_filter, 0.0), fQ((size_t)g.size_filter, 0.0);
        for (int i = 0; i < g.size_filter; i++) {
            if ((size_t)i < g.filtI[nFilter].size()) { fI[(size_t)i] = g.filtI[nFilter][(size_t)i]; fQ[(size_t)i] = g.filtQ[nFilter][(size_t)i]; }
        }
        if (int rc = qh_qrx_set_filters(b.rx, 0, fI.data(), fQ.data(), g.size_filter)) return rc;
        b.filt_n = nFilter; b.filt_epoch = g.filt_epoch;
    }
    if (is_fm_mode(mode)) {
        if (b.sq_level != g.squelch_level) { if (int rc = qh_qrx_set_squelch(b.rx, 0, g.squelch_level)) return rc; b.sq_level = g.squelch_level; }
    } else if (has_ssb_squelch(mode)) {
        if (b.ssb_en != g.ssb_squelch_enabled || b.ssb_lv != g.ssb_squelch_level) {
            if (int rc = qh_qrx_set_ssb_squelch(b.rx, g.ssb_squelch_enabled, g.ssb_squelch_level)) return rc;
            b.ssb_en = g.ssb_squelch_enabled; b.ssb_lv = g.ssb_squelch_level;
        }
        if (g.ssb_planned) (void)qh_qrx_ssb_squelch_planned(b.rx, 1);
        if (b.ssb_bw != g.filt_bw[0]) {         // "bw = filter_bandwidth[0]" in every bank, whichever filter set it runs (quisk.c:1120)
            if (int rc = qh_qrx_set_ssb_squelch_bandwidth(b.rx, g.filt_bw[0])) return rc;
            b.ssb_bw = g.filt_bw[0];
        }
    }
    return QH_OK;
}

int ensure_agc(int k)
{
    if (!g.agc[k]) {
        g.agc[k] = qh_qagc_create(0, 1, g.playback_rate, 0.7, 1.0, 0, g.stream);       // struct AgcState {0.7, 0, 0}, quisk.c:2321,2174
        if (!g.agc[k]) return QH_ERR_HIP;
        g.agc_gain_set[k] = -1.0;
    }
    if (g.agc_gain_set[k] != g.agc_gain) { if (int rc = qh_qagc_set_gain(g.agc[k], -1, g.agc_gain)) return rc; g.agc_gain_set[k] = g.agc_gain; }
    return QH_OK;
}
int run_agc(int k, int is_cpx, double2 *buf, int n)
{
    if (int rc = ensure_agc(k)) return rc;
    if (int rc = qh_qagc_set_cpx(g.agc[k], is_cpx)) return rc;
    return qh_qagc_process(g.agc[k], buf, n, n);
}

void free_all()
{
    for (Bank &b : g.bank) destroy_bank(b, false);
    destroy_bank(g.mf_bank, false);
    if (g.pan) { qh_pan_destroy(g.pan); g.pan = nullptr; }
    if (g.nb) { qh_nb_destroy(g.nb); g.nb = nullptr; }
    for (qh_qagc *&a : g.agc) if (a) { qh_qagc_destroy(a); a = nullptr; }
    if (g.up) { qh_rat_destroy(g.up); g.up = nullptr; }
    if (g.mf_dec) { qh_fir_destroy(g.mf_dec); g.mf_dec = nullptr; }
    if (g.mf_fft) { qh_pan_destroy(g.mf_fft); g.mf_fft = nullptr; }
    g.d_raw.release(); g.d_x.release(); g.d_nb.release(); g.d_o0.release(); g.d_o1.release(); g.d_mix.release(); g.d_fd.release();
    g.d_up.release(); g.d_a.release(); g.d_b.release(); g.d_sub.release(); g.d_sub0.release(); g.d_s1.release(); g.d_mf.release();
    g.d_mf8.release(); g.d_ext.release(); g.d_wd.release();
    for (auto &row : g.b2c) for (auto &bb : row) bb.release();
    for (double2 *&h : g.fd_hist) { if (h) (void)hipFree(h); h = nullptr; }
    if (g.d_flags) { (void)hipFree(g.d_flags); g.d_flags = nullptr; }
    if (g.h_flags) { (void)hipHostFree(g.h_flags); g.h_flags = nullptr; }
    g.h_in.release(); g.h_out.release(); g.h_sub1.release();
    if (g.stream) { (void)hipStreamDestroy(g.stream); g.stream = nullptr; }
}

// The key is down (or was a moment ago): the block is not demodulated; sidetone or silence at the playback rate take its
// place (quisk.c:2368-2433).  Returns the number of samples written, or -1 when the block is radio sound.
int key_block(double *cSamples, int nSamples)
{
    auto take = [&]() {         // the block's share of playback samples, the fraction carried over (quisk.c:2372-2375)
        g.out_counter += (double)nSamples * g.playback_rate / g.sample_rate;
        const int nout = (int)g.out_counter;
        g.out_counter -= nout;
        return nout;
    };
    const double env_step = 1.0 / (g.playback_rate * 5e-3);     // 5 milliseconds
    if (g.key_down && !g.is_fdx) {
        const int nout = take();
        g.play_silence = (int)(g.playback_rate * 1E-3 * g.txrx_silence_ms);
        g.keyup_env = 0;
        if (g.active_sidetone == 2 && g.cw_key_down) {          // play sidetone instead of radio for CW
            if (!g.sidetone_on) { g.sidetone_on = 1; g.sidetone_env = 0; g.sidetone_vec = 2.2e9; }       // BIG_VOLUME, quisk.h:11
            for (int i = 0; i < nout; i++) {
                if (g.sidetone_env < 1.0) { g.sidetone_env += env_step; if (g.sidetone_env > 1.0) g.sidetone_env = 1.0; }
                const double d = g.sidetone_vec.real() * g.sidetone_volume * g.sidetone_env;
                cSamples[2 * i] = d; cSamples[2 * i + 1] = d;
                g.sidetone_vec *= g.sidetone_phase;
            }
        } else {
            std::memset(cSamples, 0, (size_t)nout * 2 * sizeof(double));
        }
        return nout;
    }
    if (g.sidetone_on) {        // the key is up: the sidetone fades, then silence
        const int nout = take();
        int i = 0;
        for (; i < nout; i++) {
            g.sidetone_env -= env_step;
            if (g.sidetone_env < 0) { g.sidetone_on = 0; g.sidetone_env = 0; break; }
            const double d = g.sidetone_vec.real() * g.sidetone_volume * g.sidetone_env;
            cSamples[2 * i] = d; cSamples[2 * i + 1] = d;
            g.sidetone_vec *= g.sidetone_phase;
        }
        for (; i < nout; i++) { cSamples[2 * i] = 0; cSamples[2 * i + 1] = 0; g.play_silence--; }
        return nout;
    }
    if (g.play_silence > 0) {
        const int nout = take();
        std::memset(cSamples, 0, (size_t)nout * 2 * sizeof(double));
        g.play_silence -= nout;
        return nout;
    }
    return -1;
}

// measure_freq (quisk.c:5579-5649) on the device: the decimated samples come from a receiver of their own (the main bank runs
// its decimators and its demodulator's as ONE filter, so the 48 ksps stream between them does not exist there): wide DGT-IQ is
// exactly tune + quisk_process_decimate.  Then HalfBand1..3 as one 293-tap /8 filter, 12000-sample blocks under the reference's
// window through the chirp-z transform of the panadapter engine, |X| summed in fftshift order.  The peak search over +-500 Hz
// and its three-point interpolation read the averaged spectrum back once every measure_freq_mode / 2 transforms.
int measure_freq(const double2 *d_in, int n)
{
    if (int rc = ensure_bank(g.mf_bank, 9, 0, g.tune, 4, 20000)) return rc;     // wide DGT-IQ: no Rx filter (quisk.c:2143)
    if (!g.mf_dec) {
        double t[43];
        qh_hb45_taps(t);
        std::vector<double> c1(t, t + 43), c2(85, 0.0), c4(169, 0.0);
        for (int i = 0; i < 43; i++) { c2[2 * (size_t)i] = t[i]; c4[4 * (size_t)i] = t[i]; }
        auto conv = [](const std::vector<double> &a, const std::vector<double> &b) {
            std::vector<double> r(a.size() + b.size() - 1, 0.0);
            for (size_t i = 0; i < a.size(); i++) for (size_t j = 0; j < b.size(); j++) r[i + j] += a[i] * b[j];
            return r;
        };
        const std::vector<double> h = conv(conv(c1, c2), c4);
        g.mf_dec = qh_fir_create(0, 1, h.data(), nullptr, (int)h.size(), 8, QH_F64, g.stream);
        if (!g.mf_dec) return QH_ERR_HIP;
    }
    if (!g.mf_fft) {
        g.mf_fft = qh_pan_create(0, 1, kMfSize, 16, 6000.0, g.stream);
        if (!g.mf_fft) return QH_ERR_HIP;
        std::vector<double> w((size_t)kMfSize);
        for (int i = 0; i < kMfSize; i++) w[(size_t)i] = 0.50 - 0.50 * std::cos(2. * M_PI * i / (kMfSize - 1));     // quisk.c:5600-5601
        if (int rc = qh_pan_set_window(g.mf_fft, w.data())) return rc;
        g.mf_avg.assign((size_t)kMfSize, 0.0);
    }
    const int cap = qh_qrx_out_count(g.mf_bank.rx, n);
    if (int rc = g.d_mf.need((size_t)cap + 1)) return rc;
    int nd = 0;
    if (int rc = qh_qrx_process(g.mf_bank.rx, reinterpret_cast<const double *>(d_in), n, n, reinterpret_cast<double *>(g.d_mf.p), (long long)g.d_mf.cap, &nd)) return rc;
    const int srate = qh_qrx_decim_rate(g.mf_bank.rx) / 8;
    if (nd <= 0) return QH_OK;
    const int n8cap = qh_fir_out_count(g.mf_dec, nd);
    if (int rc = g.d_mf8.need((size_t)n8cap + 1)) return rc;
    int n8 = 0;
    if (int rc = qh_fir_process(g.mf_dec, g.d_mf.p, nd, nd, g.d_mf8.p, (long long)g.d_mf8.cap, &n8)) return rc;
    // "for (i = 0; i < nSamples && index < fft_size; ...)": the transform runs when the array is full and the rest of the call is dropped
    const int take = n8 < kMfSize - g.mf_index ? n8 : kMfSize - g.mf_index;
    if (take > 0) if (int rc = qh_pan_feed(g.mf_fft, reinterpret_cast<const double *>(g.d_mf8.p), take, take)) return rc;
    g.mf_index += take;
    if (g.mf_index < kMfSize) return QH_OK;
    g.mf_index = 0;
    (void)qh_pan_drop_partial(g.mf_fft);
    g.mf_count++;
    if (g.mf_count < g.measure_mode / 2) return QH_OK;
    g.mf_count = 0;
    if (int rc = qh_pan_read_avg(g.mf_fft, g.mf_avg.data(), 1)) return rc;
    const double *avg = g.mf_avg.data();
    const int N = kMfSize;
    double dmax = 1.e-20;
    int ipeak = 0;
    const int center = N / 2 - g.rit_freq * N / srate;
    int k = 500;
    k = k * N / srate;
    for (int i = center - k; i <= center + k; i++)
        if (i >= 0 && i < N && avg[i] > dmax) { dmax = avg[i]; ipeak = i; }
    if (ipeak < 1 || ipeak > N - 2) return QH_OK;
    const double c3 = 1.36 * (avg[ipeak + 1] - avg[ipeak - 1]) / (avg[ipeak - 1] + avg[ipeak] + avg[ipeak + 1]);
    double freq = srate * (2 * (ipeak + c3) - N) / 2 / N;
    freq += g.tune;
    g.measured_frequency = freq;
    return QH_OK;
}

// the body of quisk_process_samples behind the key handling; returns the output count or -1 (error set)
int process_radio(double *cSamples, int nSamples)
{
    const int n = nSamples;
    hipStream_t s = g.stream;
    // ---- the block goes to the device once
    if (g.h_in.need((size_t)n * 2) || g.d_raw.need((size_t)n)) return -1;
    std::memcpy(g.h_in.p, cSamples, (size_t)n * 2 * sizeof(double));
    if (hipMemcpyAsync(g.d_raw.p, g.h_in.p, (size_t)n * 16, hipMemcpyHostToDevice, s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "upload failed"); return -1; }
    const double2 *cur = g.d_raw.p;
    // ---- AddTestTone and the inversion (quisk.c:2438-2446)
    if (g.tone_on || g.invert_spectrum) {
        if (g.d_x.need((size_t)n)) return -1;
        const int kind = !g.tone_on ? -1 : g.mode == 4 ? 1 : is_fm_mode(g.mode) ? 2 : 0;
        const u64 da = turns_step(1000.0, (double)g.sample_rate);
        hipLaunchKernelGGL(qh_ps::prep_kernel, dim3(grid_for(n), 1u), dim3(256), 0, s, cur, (long long)n, g.d_x.p, (long long)n, n, kind, g.tone_phase,
                           g.tone_step, g.audio_phase, da, g.invert_spectrum);
        if (g.tone_on) {
            g.tone_phase += g.tone_step * (u64)n;
            if (kind >= 1) g.audio_phase += da * (u64)n;
        }
        cur = g.d_x.p;
    }
    // ---- NoiseBlanker, not while the key is down (full duplex reaches this point with the key down; quisk.c:2448-2449)
    if (!g.key_down && (g.nb_level > 0 || g.nb)) {
        if (!g.nb && !(g.nb = qh_nb_create(0, 1, g.sample_rate, s))) return -1;
        if (qh_nb_set_level(g.nb, g.nb_level) || g.d_nb.need((size_t)n)) return -1;
        if (qh_nb_process(g.nb, cur, n, g.d_nb.p, n, n)) return -1;
        cur = g.d_nb.p;
    }
    // ---- the FFT ring producer (quisk.c:2454-2475)
    if (g.pan && qh_pan_feed(g.pan, reinterpret_cast<const double *>(cur), n, n)) return -1;
    // ---- mode EXT: tune, the user's demodulator, then straight to the AGC (quisk.c:2490-2493: "goto start_agc")
    if (g.mode == 6) {
        if (!g.ext_demod) { qh::set_error(QH_ERR_UNSUPPORTED, "mode EXT calls the user's quisk_extern_demod (extdemod.c:13): none registered (qh_quisk_set_extern_demod)"); return -1; }
        destroy_bank(g.bank[0], true);                           // the receive oscillator's phase comes back from the bank that carried it
        if (g.d_ext.need((size_t)n) || g.h_out.need((size_t)n * 2)) return -1;
        const u64 dph = g.tune != 0 ? turns_step(-(double)g.tune, (double)g.sample_rate) : 0ull;
        hipLaunchKernelGGL(ps_tune_kernel, dim3(grid_for(n)), dim3(256), 0, s, cur, g.d_ext.p, n, g.phase[0], dph);
        g.phase[0] += dph * (u64)n;
        // the plug-in is host code: the one round trip of this path
        if (hipMemcpyAsync(g.h_out.p, g.d_ext.p, (size_t)n * 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            qh::set_error(QH_ERR_HIP, "download failed"); return -1;
        }
        const int na = g.ext_demod(g.h_out.p, n, (double)g.sample_rate / g.playback_rate);      // "total decimation needed"
        if (na < 0 || na > n) { qh::set_error(QH_ERR_INVALID, "quisk_extern_demod returned %d samples for a block of %d", na, n); return -1; }
        double2 *audio = g.d_ext.p;
        if (na > 0) {
            if (hipMemcpyAsync(audio, g.h_out.p, (size_t)na * 16, hipMemcpyHostToDevice, s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "upload failed"); return -1; }
            if (run_agc(0, 1, audio, na)) return -1;             // "Ext and DGT-IQ stereo sound", quisk.c:2686-2688
        }
        int env_n = 0;
        const double env0 = g.keyup_env, env_step = 1. / (g.playback_rate * 5e-3);
        if (g.keyup_env < 1.0 && na > 0)
            for (int i = 0; i < na; i++) {
                g.keyup_env += env_step;
                if (g.keyup_env > 1.0) { g.keyup_env = 1.0; break; }
                env_n++;
            }
        hipLaunchKernelGGL(qh_ps::epilogue_kernel, dim3(grid_for(na), 1u), dim3(256), 0, s, (const double2 *)audio, 0LL, audio, 0LL, na, (const int *)nullptr, 0,
                           (const int *)nullptr, 0, g.kill_audio, env0, env_step, env_n, g.d_flags);
        if (na > 0 && hipMemcpyAsync(g.h_out.p, audio, (size_t)na * 16, hipMemcpyDeviceToHost, s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "download failed"); return -1; }
        if (hipMemcpyAsync(g.h_flags, g.d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipGetLastError() != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "qh_quisk_process_samples: the device chain failed"); return -1; }
        if (na > 0) std::memcpy(cSamples, g.h_out.p, (size_t)na * 2 * sizeof(double));
        g.squelch_real = g.h_flags[0]; g.squelch_imag = g.h_flags[1];
        return na;
    }
    // ---- bank 0: tune, decimate, demodulate
    Bank &b0 = g.bank[0];
    if (ensure_bank(b0, g.mode, 0, g.tune, 0)) return -1;
    if (g.notch_reset || b0.notch_applied != g.auto_notch) {     // set_auto_notch (or a fresh bank) starts the notch over
        if (g.mode != 9 /* DGT-IQ has no notch */) if (qh_qrx_set_auto_notch(b0.rx, g.auto_notch, g.rit_freq)) return -1;
        b0.notch_applied = g.auto_notch; g.notch_reset = false;
    }
    const int cap0 = qh_qrx_out_count(b0.rx, n);
    if (g.d_o0.need((size_t)cap0 + 1)) return -1;
    int n0 = 0;
    if (qh_qrx_process(b0.rx, reinterpret_cast<const double *>(cur), n, n, reinterpret_cast<double *>(g.d_o0.p), (long long)g.d_o0.cap, &n0)) return -1;
    const int decim_srate = qh_qrx_decim_rate(b0.rx);
    if (qh_qrx_ssb_squelch_planned(b0.rx, 0)) g.ssb_planned = true;
    if (g.measure_mode) {                                        // quisk.c:2527-2528 (the same tuned and decimated samples)
        if (measure_freq(cur, n)) return -1;
    }
    const bool stereo = g.mode == 9;                             // DGT-IQ is already stereo (quisk.c:2534)
    const int *flag0 = qh_qrx_squelch_flag(b0.rx, 0), *flag1 = nullptr;
    const int *flag_real = nullptr, *flag_imag = nullptr;
    double2 *audio = g.d_o0.p;
    int na = n0;
    // ---- a second channel: the same receiver on the transmit frequency, or the played sub-receiver (quisk.c:2537-2621)
    int second = 0;                                              // 1 split, 2 played sub-receiver
    if (!stereo && g.split_rxtx) second = 1;
    else if (!stereo && g.play_channel >= 0 && g.sub_have[g.play_channel] == n) second = 2;
    if (second) {
        Bank &b1 = g.bank[1];
        const double2 *src2;
        if (second == 1) {
            if (ensure_bank(b1, g.mode, 0, g.tx_tune + g.rit_freq, 1)) return -1;
            src2 = g.d_raw.p;                                    // orig_cSamples: copied ahead of the test tone, the inversion and the blanker
        } else {
            const int pc = g.play_channel;
            if (ensure_bank(b1, g.sub_mode[pc], 1, g.sub_freq[pc], 2)) return -1;
            if (g.d_sub.need((size_t)n)) return -1;
            if (hipMemcpyAsync(g.d_sub.p, g.sub_samples[pc].data(), (size_t)n * 16, hipMemcpyHostToDevice, s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "upload failed"); return -1; }
            if (hipStreamSynchronize(s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "upload failed"); return -1; }     // pageable source
            src2 = g.d_sub.p;
        }
        if (g.ssb_planned) (void)qh_qrx_ssb_squelch_planned(b1.rx, 1);
        const int cap1 = qh_qrx_out_count(b1.rx, n);
        if (g.d_o1.need((size_t)cap1 + 1)) return -1;
        int n1 = 0;
        if (qh_qrx_process(b1.rx, reinterpret_cast<const double *>(src2), n, n, reinterpret_cast<double *>(g.d_o1.p), (long long)g.d_o1.cap, &n1)) return -1;
        flag1 = qh_qrx_squelch_flag(b1.rx, 0);
        if (qh_qrx_ssb_squelch_planned(b1.rx, 0)) g.ssb_planned = true;
        // which stream is the real (left) output
        int sel_re = 0, sel_im = 1;
        if (second == 1) {
            switch (g.split_rxtx) {
            default:
            case 1: if (!(g.tx_tune < g.tune)) { sel_re = 1; sel_im = 0; } break;       // higher frequency is real
            case 2: if (!(g.tx_tune >= g.tune)) { sel_re = 1; sel_im = 0; } break;      // lower frequency is real
            case 3: sel_re = sel_im = 0; break;
            case 4: sel_re = sel_im = 1; break;
            }
        } else {
            switch (g.play_method) {
            default:
            case 0: sel_re = sel_im = 1; break;
            case 1: sel_re = 0; sel_im = 1; break;
            case 2: sel_re = 1; sel_im = 0; break;
            }
        }
        flag_real = sel_re ? flag1 : flag0;
        flag_imag = sel_im ? flag1 : flag0;
        // Buffer2Chan(dsamples, nSamples, dsamples2, n)
        int nout;
        if (g.nbuf1 == 0 && g.nbuf2 == 0 && n0 == n1) {
            nout = n0;
            if (g.d_mix.need((size_t)nout + 1)) return -1;
            if (nout > 0)
                hipLaunchKernelGGL(ps_join_kernel, dim3(grid_for(nout)), dim3(256), 0, s, (const double *)nullptr, 0, (const double2 *)g.d_o0.p,
                                   (const double *)nullptr, 0, (const double2 *)g.d_o1.p, nout, sel_re, sel_im, g.d_mix.p);
        } else {
            if (n0 + g.nbuf1 >= kBuf2Chan || n1 + g.nbuf2 >= kBuf2Chan) g.nbuf1 = g.nbuf2 = 0;      // overflow: the reference starts over
            const int t1 = g.nbuf1 + n0, t2 = g.nbuf2 + n1;
            nout = t1 <= t2 ? t1 : t2;
            const int keep1 = t1 - nout, keep2 = t2 - nout, nc = g.b2c_cur;
            if (g.d_mix.need((size_t)nout + 1) || g.b2c[0][nc ^ 1].need((size_t)keep1 + 1) || g.b2c[1][nc ^ 1].need((size_t)keep2 + 1)) return -1;
            if (nout > 0)
                hipLaunchKernelGGL(ps_join_kernel, dim3(grid_for(nout)), dim3(256), 0, s, (const double *)g.b2c[0][nc].p, g.nbuf1,
                                   (const double2 *)g.d_o0.p, (const double *)g.b2c[1][nc].p, g.nbuf2, (const double2 *)g.d_o1.p, nout, sel_re,
                                   sel_im, g.d_mix.p);
            const int kmax = keep1 > keep2 ? keep1 : keep2;
            if (kmax > 0)
                hipLaunchKernelGGL(ps_b2c_keep_kernel, dim3(grid_for(kmax)), dim3(256), 0, s, (const double *)g.b2c[0][nc].p, g.nbuf1,
                                   (const double2 *)g.d_o0.p, keep1, g.b2c[0][nc ^ 1].p, (const double *)g.b2c[1][nc].p, g.nbuf2,
                                   (const double2 *)g.d_o1.p, keep2, g.b2c[1][nc ^ 1].p, nout);
            g.nbuf1 = keep1; g.nbuf2 = keep2; g.b2c_cur = nc ^ 1;
        }
        audio = g.d_mix.p; na = nout;
    } else if (!stereo) {
        flag_real = flag_imag = flag0;                           // monophonic sound on both channels: the bank wrote (d, d)
    }
    // ---- sub-receiver 1 on a digital output device (quisk.c:2630-2651)
    g.sub1_n = 0;
    {
        const int m = g.sub_mode[0];
        if (g.multirx_count > 0 && (m == 7 || m == 8 || m == 9 || m == 13) && g.sub_rx1_driver && g.sub_have[0] == n) {
            Bank &b2 = g.bank[2];
            if (ensure_bank(b2, m, 2, g.sub_freq[0], 3)) return -1;
            if (g.d_sub0.need((size_t)n)) return -1;
            if (hipMemcpyAsync(g.d_sub0.p, g.sub_samples[0].data(), (size_t)n * 16, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "upload failed"); return -1; }
            const int cap2 = qh_qrx_out_count(b2.rx, n);
            if (g.d_s1.need((size_t)cap2 + 1)) return -1;
            int n2 = 0;
            if (qh_qrx_process(b2.rx, reinterpret_cast<const double *>(g.d_sub0.p), n, n, reinterpret_cast<double *>(g.d_s1.p), (long long)g.d_s1.cap, &n2)) return -1;
            if (n2 > 0) {
                if (run_agc(2, m == 9 ? 1 : 0, g.d_s1.p, n2)) return -1;    // the bank wrote (d, d) for the mono modes; process_agc(.., 0) scales both parts alike
                if (g.h_sub1.need((size_t)n2 * 2)) return -1;
                if (hipMemcpyAsync(g.h_sub1.p, g.d_s1.p, (size_t)n2 * 16, hipMemcpyDeviceToHost, s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "download failed"); return -1; }
            }
            g.sub1_n = n2;
        }
    }
    // ---- cFracDecim to 48 ksps (quisk.c:2654-2659)
    if (decim_srate != 48000 && na > 0) {
        const double fdecim = decim_srate / 48000.0, step = fdecim - 1;
        const double d0 = g.fd_dindex;
        const int M = qh_ps::fracdecim_walk(na, g.fd_dindex, fdecim);
        if (g.d_fd.need((size_t)M + 1)) return -1;
        if (M > 0) hipLaunchKernelGGL(qh_ps::fracdecim_kernel, dim3(grid_for(M), 1u), dim3(256), 0, s, (const double2 *)audio, 0LL, (const double2 *)g.fd_hist[g.fd_cur], M,
                                      d0, step, g.d_fd.p, 0LL, na);
        hipLaunchKernelGGL(qh_ps::fd_hist_kernel, dim3(1), dim3(64), 0, s, (const double2 *)audio, 0LL, na, (const double2 *)g.fd_hist[g.fd_cur], g.fd_hist[g.fd_cur ^ 1]);
        g.fd_cur ^= 1;
        audio = g.d_fd.p; na = M;
    }
    // ---- the WDSP hand-off (quisk.c:2660-2661) with the samples where they are: the shim's ring, fexchange0's rings and the DSP blocks
    // in device memory, enqueued behind this stream (qh_wdsp_fexchange0_device) -- no round trip, no wait
    if (qh_wdsp_shim_in_size(0) <= 0) (void)wdspFexchange0(0, nullptr, 0);      // not in use: the shim only rewinds its ring (quisk_wdsp.c:32-37)
    if (const int in_size = qh_wdsp_shim_in_size(0); in_size > 0 && na > 0) {
        if (g.d_wd.need((size_t)(na + in_size))) return -1;
        if (hipMemcpyAsync(g.d_wd.p, audio, (size_t)na * 16, hipMemcpyDeviceToDevice, s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "copy failed"); return -1; }
        const int nw = qh_wdsp_fexchange0_device(0, g.d_wd.p, na, s);
        if (qh_wdsp_status() != QH_OK) return -1;
        audio = g.d_wd.p; na = nw;
    }
    // ---- interpolation to the playback rate (quisk.c:2663-2682)
    if (g.up && na > 0) {
        const int nu = qh_rat_out_count(g.up, na);
        if (g.d_up.need((size_t)nu + 1)) return -1;
        int got = 0;
        if (qh_rat_process(g.up, audio, na, na, g.d_up.p, (long long)g.d_up.cap, &got)) return -1;
        audio = g.d_up.p; na = got;
    }
    // ---- AGC (quisk.c:2686-2702)
    if (na > 0) {
        if (stereo) {
            if (run_agc(0, 1, audio, na)) return -1;
        } else if (g.split_rxtx || g.play_channel >= 0) {        // separate AGC for left and right
            if (g.d_a.need((size_t)na) || g.d_b.need((size_t)na)) return -1;
            hipLaunchKernelGGL(ps_split_kernel, dim3(grid_for(na)), dim3(256), 0, s, (const double2 *)audio, na, g.d_a.p, g.d_b.p);
            if (run_agc(0, 0, g.d_a.p, na) || run_agc(1, 0, g.d_b.p, na)) return -1;
            hipLaunchKernelGGL(ps_merge_kernel, dim3(grid_for(na)), dim3(256), 0, s, (const double2 *)g.d_a.p, (const double2 *)g.d_b.p, na, audio);
        } else {
            if (run_agc(0, 0, audio, na)) return -1;
        }
    }
    // ---- kill_audio / squelch / key-up envelope (quisk.c:2712-2738)
    // (the envelope's factors are a closed form of the sample index for the kernel; the host only steps its copy of keyupEnvelope)
    int env_n = 0;
    const double env0 = g.keyup_env, env_step = 1. / (g.playback_rate * 5e-3);
    if (g.keyup_env < 1.0 && na > 0) {
        for (int i = 0; i < na; i++) {
            g.keyup_env += env_step;
            if (g.keyup_env > 1.0) { g.keyup_env = 1.0; break; }
            env_n++;
        }
    }
    hipLaunchKernelGGL(qh_ps::epilogue_kernel, dim3(grid_for(na), 1u), dim3(256), 0, s, (const double2 *)audio, 0LL, audio, 0LL, na, flag_real, 0, flag_imag, 0,
                       g.kill_audio, env0, env_step, env_n, g.d_flags);
    if (g.h_out.need((size_t)(na > 0 ? na : 1) * 2)) return -1;
    if (na > 0 && hipMemcpyAsync(g.h_out.p, audio, (size_t)na * 16, hipMemcpyDeviceToHost, s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "download failed"); return -1; }
    if (hipMemcpyAsync(g.h_flags, g.d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "download failed"); return -1; }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { qh::set_error(QH_ERR_HIP, "qh_quisk_process_samples: the device chain failed"); return -1; }
    if (na > 0) std::memcpy(cSamples, g.h_out.p, (size_t)na * 2 * sizeof(double));
    g.squelch_real = g.h_flags[0]; g.squelch_imag = g.h_flags[1];
    if (g.sub1_n > 0) g.sub1_out.assign(g.h_sub1.p, g.h_sub1.p + 2 * (size_t)g.sub1_n);
    return na;
}

}  // namespace

extern "C" {

// quisk_sound_state.sample_rate and .playback_rate (open_sound, quisk.c:4106) + the filters.h tables (data, passed in like to
// qh_qrx_create_ex) + record_app's fft_size and data_width (0, 0: no panadapter).
static int open_locked(int sample_rate, int playback_rate, const qh_qrx_tables *tables, int fft_size, int data_width);
int qh_quisk_open(int sample_rate, int playback_rate, const qh_qrx_tables *tables, int fft_size, int data_width)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    const int rc = open_locked(sample_rate, playback_rate, tables, fft_size, data_width);
    if (rc) {               // a receiver that failed to open is closed: the next qh_quisk_process_samples says so instead of launching on null buffers
        if (g.sample_rate) (void)hipSetDevice(0);
        free_all();
        g.sample_rate = 0; g.have_tables = false;
    }
    return rc;
}
static int open_locked(int sample_rate, int playback_rate, const qh_qrx_tables *tables, int fft_size, int data_width)
{
    static const int len[13] = { 98, 147, 245, 50, 36, 186, 309, 125, 55, 114, 136, 174, 189 };
    if (sample_rate <= 0 || !tables) return qh::set_error(QH_ERR_INVALID, "qh_quisk_open: bad arguments");
    const int ratio = playback_rate / 48000;
    if (playback_rate <= 0 || playback_rate % 48000 || (ratio != 1 && ratio != 2 && ratio != 4 && ratio != 8))
        return qh::set_error(QH_ERR_UNSUPPORTED, "Failure in quisk.c in integer interpolation: playback rate %d is not 48000 x 1, 2, 4 or 8 (quisk.c:2664-2681)",
                             playback_rate);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return qh::set_error(QH_ERR_NO_DEVICE, "no HIP device (libquiskhip has no CPU fallback)");
    QH_HIP(hipSetDevice(0));
    free_all();
    const double *src[13] = { tables->f48dec24, tables->f144d3, tables->f240d5, tables->audio24p4, tables->audio24p6, tables->lp48,
                              tables->fmhp, tables->f300d5, tables->sdriq53, tables->sdriq111, tables->sdriq133, tables->sdriq167,
                              tables->sdriq185 };
    const double *dst[13];
    for (int i = 0; i < 13; i++) {
        if (src[i]) { g.tables[i].assign(src[i], src[i] + len[i]); dst[i] = g.tables[i].data(); }
        else { g.tables[i].clear(); dst[i] = nullptr; }
    }
    g.t.f48dec24 = dst[0]; g.t.f144d3 = dst[1]; g.t.f240d5 = dst[2]; g.t.audio24p4 = dst[3]; g.t.audio24p6 = dst[4];
    g.t.lp48 = dst[5]; g.t.fmhp = dst[6]; g.t.f300d5 = dst[7]; g.t.sdriq53 = dst[8]; g.t.sdriq111 = dst[9];
    g.t.sdriq133 = dst[10]; g.t.sdriq167 = dst[11]; g.t.sdriq185 = dst[12];
    g.have_tables = true;
    g.sample_rate = sample_rate; g.playback_rate = playback_rate;
    g.fft_size = fft_size; g.data_width = data_width;
    QH_HIP(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    // everything starts like the reference's statics and globals at program start
    g.mode = 0; g.tune = 0; g.agc_gain = 80.0; g.nb_level = 0; g.auto_notch = 0; g.rit_freq = 0; g.notch_reset = false;
    for (int i = 0; i < 3; i++) { g.filtI[i].clear(); g.filtQ[i].clear(); g.filt_bw[i] = 0; }
    g.size_filter = 0; g.sidetone_volume = 0.0; g.sidetone_phase = 1.0; g.txrx_silence_ms = 50;
    g.tx_tune = 0; g.split_rxtx = 0; g.play_channel = -1; g.play_method = 0; g.old_split = 0; g.old_play = 0;
    for (int i = 0; i < QuiskRx::kMaxSub; i++) { g.sub_freq[i] = 0; g.sub_mode[i] = 0; g.sub_have[i] = 0; }
    g.multirx_count = 0; g.sub_rx1_driver = 0;
    for (u64 &p : g.phase) p = 0;
    g.key_down = g.cw_key_down = g.active_sidetone = g.is_fdx = g.kill_audio = g.invert_spectrum = 0;
    g.out_counter = 0.0; g.sidetone_env = 0.0; g.keyup_env = 1.0; g.sidetone_on = 0; g.play_silence = 0;
    g.sidetone_vec = 2.2e9;
    g.tone_on = false; g.tone_phase = 0; g.audio_phase = 0;
    g.nbuf1 = g.nbuf2 = 0; g.b2c_cur = 0;
    g.fd_dindex = 1.0; g.fd_cur = 0;
    g.measure_mode = 0; g.mf_index = 0; g.mf_count = 0; g.measured_frequency = 0.0;
    g.sub1_n = 0; g.squelch_real = g.squelch_imag = 0;
    g.squelch_level = -999.0; g.ssb_squelch_enabled = 0; g.ssb_squelch_level = 0; g.ssb_planned = false;
    for (int i = 0; i < 2; i++) {
        QH_HIP(hipMalloc((void **)&g.fd_hist[i], 3 * sizeof(double2)));
        QH_HIP(qh::dev_zero(g.fd_hist[i], 3 * sizeof(double2)));
    }
    QH_HIP(hipMalloc((void **)&g.d_flags, 2 * sizeof(int)));
    QH_HIP(qh::dev_zero(g.d_flags, 2 * sizeof(int)));
    QH_HIP(hipHostMalloc((void **)&g.h_flags, 2 * sizeof(int), hipHostMallocDefault));
    for (auto &row : g.b2c) for (auto &bb : row) if (int rc = bb.need(kBuf2Chan)) return rc;
    if (fft_size > 0 && data_width > 0) {
        g.pan = qh_pan_create(0, 1, fft_size, data_width, (double)sample_rate, g.stream);
        if (!g.pan) return QH_ERR_HIP;
    }
    g.up_ratio = ratio;
    if (ratio > 1) {
        const std::vector<double> taps = qh_ps::playback_interp_taps(ratio);
        g.up = qh_rat_create(0, 1, taps.data(), (int)taps.size(), ratio, 1, QH_F64, g.stream);
        if (!g.up) return QH_ERR_HIP;
    }
    g.filt_epoch++;
    return QH_OK;
}

void qh_quisk_close(void)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    if (g.sample_rate) (void)hipSetDevice(0);
    free_all();
    g.sample_rate = 0;
}

void qh_quisk_set_tune(int rx_tune_freq)            // set_tune, quisk.c:4702
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.tune = rx_tune_freq;
}

void qh_quisk_set_rx_mode(int mode)                 // set_rx_mode, quisk.c:4621
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.mode = mode;
}

// set_filters(filterI, filterQ, bandwidth, start_offset, nFilter), quisk.c:4551: the taps of filter set nFilter, its bandwidth,
// and the ONE global sizeFilter that every bank's filter then runs with
int qh_quisk_set_filters_n(const double *filtI, const double *filtQ, int size, int bandwidth, int nFilter)
{
    if (size < 0 || size >= 10001 || (size > 0 && (!filtI || !filtQ)))
        return qh::set_error(QH_ERR_INVALID, "Filter size must be less than 10001");    // MAX_FILTER_SIZE, quisk.c:4576
    if (nFilter < 0 || nFilter > 2) return qh::set_error(QH_ERR_INVALID, "nFilter must be 0, 1 or 2 (MAX_RX_FILTERS, quisk.c:127)");
    std::lock_guard<std::mutex> lk(g.mtx);
    std::vector<double> &fI = g.filtI[nFilter], &fQ = g.filtQ[nFilter];
    if (fI.size() < (size_t)size) { fI.resize((size_t)size, 0.0); fQ.resize((size_t)size, 0.0); }    // entries beyond `size` keep what earlier calls left
    for (int i = 0; i < size; i++) { fI[(size_t)i] = filtI[i]; fQ[(size_t)i] = filtQ[i]; }
    g.filt_bw[nFilter] = bandwidth;
    g.size_filter = size;
    g.filt_epoch++;
    return QH_OK;
}
int qh_quisk_set_filters(const double *filtI, const double *filtQ, int size, int bandwidth) { return qh_quisk_set_filters_n(filtI, filtQ, size, bandwidth, 0); }
int qh_quisk_set_filters2(const double *filtI, const double *filtQ, int size, int bandwidth) { return qh_quisk_set_filters_n(filtI, filtQ, size, bandwidth, 1); }

void qh_quisk_set_agc(double level)                 // set_agc, quisk.c:4543
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.agc_gain = level;
}

void qh_quisk_set_auto_notch(int on, int rit_freq)   // set_auto_notch, quisk.c:4596: the flag, and dAutoNotch(NULL, ...)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    (void)rit_freq;         // the reference's set_auto_notch takes the flag alone: rit_freq is the global set_sidetone writes (quisk.c:4712)
    g.auto_notch = on ? 1 : 0; g.notch_reset = true;
}

void qh_quisk_set_noise_blanker(int level)          // set_noise_blanker, quisk.c:4605
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.nb_level = level < 0 ? 0 : level;
}

int qh_quisk_get_filter_rate(void)                  // get_filter_rate(-1, 0): the rate the current Rx filter runs at
{
    std::lock_guard<std::mutex> lk(g.mtx);
    if (!g.sample_rate || !g.have_tables) { qh::set_error(QH_ERR_INVALID, "qh_quisk_open has not been called"); return 0; }
    if (ensure_bank(g.bank[0], g.mode, 0, g.tune, 0)) return 0;
    return qh_qrx_filter_rate(g.bank[0].rx);
}

// quisk_process_samples (quisk.c:2289): in place; returns the number of output samples at the playback rate
// (the buffer must have room for them: SAMP_BUFFER_SIZE in the reference); nSamples <= 0 is returned unchanged.
int qh_quisk_process_samples(double *cSamples, int nSamples)
{
    if (nSamples <= 0) return nSamples;                                  // quisk.c:2336-2337
    if (!cSamples) { qh::set_error(QH_ERR_INVALID, "null sample buffer"); return 0; }
    std::lock_guard<std::mutex> lk(g.mtx);
    if (!g.sample_rate || !g.have_tables) { qh::set_error(QH_ERR_INVALID, "qh_quisk_open has not been called"); return 0; }
    if (hipSetDevice(0) != hipSuccess) { qh::set_error(QH_ERR_NO_DEVICE, "no HIP device (libquiskhip has no CPU fallback)"); return 0; }
    // quisk.c:2360-2367: a new split or another play channel starts Buffer2Chan over -- ahead of the key handling
    if (g.split_rxtx && !g.old_split) g.nbuf1 = g.nbuf2 = 0;
    if (g.play_channel != g.old_play) g.nbuf1 = g.nbuf2 = 0;
    g.old_split = g.split_rxtx; g.old_play = g.play_channel;
    int out = key_block(cSamples, nSamples);
    if (out < 0) {
        out = process_radio(cSamples, nSamples);
        if (out < 0) { out = 0; g.failed_calls++; }             // qh_last_error() says why; qh_quisk_error_count() that it happened
    }
    for (int &h : g.sub_have) h = 0;
    return out;
}

// ---- the setters of the orchestration state (names after the QS calls they mirror)
void qh_quisk_set_tx_tune(int tx_tune_freq) { std::lock_guard<std::mutex> lk(g.mtx); g.tx_tune = tx_tune_freq; }           // set_tune's 2nd argument, quisk.c:4702
void qh_quisk_set_split_rxtx(int split) { std::lock_guard<std::mutex> lk(g.mtx); g.split_rxtx = split; }                  // quisk.c:4694
void qh_quisk_set_multirx_play_channel(int ch) { std::lock_guard<std::mutex> lk(g.mtx); g.play_channel = ch >= QuiskRx::kMaxSub ? -1 : ch; }    // quisk.c:4856
void qh_quisk_set_multirx_play_method(int m) { std::lock_guard<std::mutex> lk(g.mtx); g.play_method = m; }                 // quisk.c:4846
void qh_quisk_set_multirx_freq(int index, int freq) { std::lock_guard<std::mutex> lk(g.mtx); if (index >= 0 && index < QuiskRx::kMaxSub) g.sub_freq[index] = freq; }   // quisk.c:4826
void qh_quisk_set_multirx_mode(int index, int mode) { std::lock_guard<std::mutex> lk(g.mtx); if (index >= 0 && index < QuiskRx::kMaxSub) g.sub_mode[index] = mode; }   // quisk.c:4836
void qh_quisk_set_multirx_count(int n) { std::lock_guard<std::mutex> lk(g.mtx); g.multirx_count = n; }                     // quisk_multirx_count
void qh_quisk_set_sub_rx1_output(int on) { std::lock_guard<std::mutex> lk(g.mtx); g.sub_rx1_driver = on; }                 // quiskPlaybackDevices[QUISK_INDEX_SUB_RX1]->driver
// a sub-receiver's samples for the coming qh_quisk_process_samples call (the reference's sample source fills
// multirx_cSamples[index] with as many samples as the main receiver gets); kept for the played one and for sub-receiver 1
int qh_quisk_multirx_samples(int index, const double *cSamples, int nSamples)
{
    if (!cSamples || nSamples <= 0) return qh::set_error(QH_ERR_INVALID, "qh_quisk_multirx_samples: no samples");
    std::lock_guard<std::mutex> lk(g.mtx);
    if (index < 0 || index >= QuiskRx::kMaxSub) return QH_OK;
    if (index != g.play_channel && index != 0) return QH_OK;    // only these two are demodulated on this path
    g.sub_samples[index].assign(cSamples, cSamples + 2 * (size_t)nSamples);
    g.sub_have[index] = nSamples;
    return QH_OK;
}
// what play_sound_interface(quiskPlaybackDevices[QUISK_INDEX_SUB_RX1], ...) was handed in the last call (quisk.c:2651);
// returns its length (copies at most `cap` samples)
int qh_quisk_sub_rx1_audio(double *cSamples, int cap)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    const int n = g.sub1_n < cap ? g.sub1_n : cap;
    if (n > 0 && cSamples) std::memcpy(cSamples, g.sub1_out.data(), (size_t)n * 2 * sizeof(double));
    return g.sub1_n;
}
// quisk_is_key_down() / QUISK_CWKEY_DOWN / quisk_active_sidetone / quisk_isFDX as the caller sees them now; the sidetone
// of set_sidetone (quisk.c:4710: volume, |rit_freq| as its pitch) at the playback rate of qh_quisk_open
void qh_quisk_set_key_state(int key_down, int cw_key_down, int active_sidetone, int is_fdx)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.key_down = key_down; g.cw_key_down = cw_key_down; g.active_sidetone = active_sidetone; g.is_fdx = is_fdx;
}
void qh_quisk_set_sidetone(double volume, int rit_freq, int playback_rate, int txrx_silence_msec)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    (void)playback_rate;                                // the playback rate is qh_quisk_open's (open_sound sets it, quisk.c:4106)
    g.sidetone_volume = volume; g.rit_freq = rit_freq;
    if (g.mode == 0 || g.mode == 1) g.notch_reset = true;       // for CW, changing the RIT affects autonotch (quisk.c:4716-4717)
    if (txrx_silence_msec >= 0) g.txrx_silence_ms = txrx_silence_msec;
    g.sidetone_phase = std::exp(std::complex<double>(0.0, 2.0 * M_PI * std::abs(rit_freq) / g.playback_rate));
}
void qh_quisk_set_kill_audio(int kill) { std::lock_guard<std::mutex> lk(g.mtx); g.kill_audio = kill; }
void qh_quisk_invert_spectrum(int invert) { std::lock_guard<std::mutex> lk(g.mtx); g.invert_spectrum = invert; }           // quisk.c:4535
void qh_quisk_set_squelch(double level) { std::lock_guard<std::mutex> lk(g.mtx); g.squelch_level = level; }               // set_squelch, quisk.c:4721
void qh_quisk_set_ssb_squelch(int enabled, int level)                                                                     // set_ssb_squelch, quisk.c:4729
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.ssb_squelch_enabled = enabled; g.ssb_squelch_level = level;
}
// get_filter() (quisk.c:5481-5568): the response of cFilterI/Q[0] (all sizeFilter taps, up to MAX_FILTER_SIZE) in dB, data_width
// values, negative frequencies first.  Returns data_width, or 0 with qh_last_error() set.
int qh_quisk_get_filter(double *db)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    if (!g.sample_rate || !g.have_tables) { qh::set_error(QH_ERR_INVALID, "qh_quisk_open has not been called"); return 0; }
    if (!db || g.data_width <= 0 || g.fft_size < g.data_width) { qh::set_error(QH_ERR_INVALID, "qh_quisk_get_filter: the receiver was opened without a graph (data_width, fft_size)"); return 0; }
    if (hipSetDevice(0) != hipSuccess) { qh::set_error(QH_ERR_NO_DEVICE, "no HIP device (libquiskhip has no CPU fallback)"); return 0; }
    const int W = g.data_width, N = g.size_filter, total = W + N;
    int nf = 0;
    for (int f = 1; f < W / 2.0 - 10.0; f++) nf = f;
    std::vector<double> fI((size_t)(N > 0 ? N : 1), 0.0), fQ((size_t)(N > 0 ? N : 1), 0.0);
    for (int i = 0; i < N; i++) {
        if ((size_t)i < g.filtI[0].size()) { fI[(size_t)i] = g.filtI[0][(size_t)i]; fQ[(size_t)i] = g.filtQ[0][(size_t)i]; }
    }
    double *d = nullptr;
    const size_t words = (size_t)total + 2 * fI.size() + 2 * (size_t)W + (size_t)W;
    if (hipMalloc((void **)&d, words * sizeof(double)) != hipSuccess) { qh::set_error(QH_ERR_HIP, "qh_quisk_get_filter: hipMalloc failed"); return 0; }
    double *dx = d, *dfI = dx + total, *dfQ = dfI + fI.size(), *dy = dfQ + fI.size(), *dout = dy + 2 * (size_t)W;
    hipStream_t s = g.stream;
    bool ok = hipMemcpyAsync(dfI, fI.data(), fI.size() * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(dfQ, fQ.data(), fQ.size() * 8, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(gf_multitone_kernel, dim3(grid_for(total)), dim3(256), 0, s, dx, total, W, nf);
        hipLaunchKernelGGL(gf_filter_kernel, dim3(grid_for(W)), dim3(256), 0, s, (const double *)dx, (const double *)dfI, (const double *)dfQ, N, W, g.fft_size,
                           reinterpret_cast<double2 *>(dy));
        hipLaunchKernelGGL(gf_dft_kernel, dim3(grid_for(W)), dim3(256), 0, s, (const double2 *)reinterpret_cast<double2 *>(dy), W, dout);
        ok = hipMemcpyAsync(db, dout, (size_t)W * 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    (void)hipFree(d);
    if (!ok) { qh::set_error(QH_ERR_HIP, "qh_quisk_get_filter: the device chain failed"); return 0; }
    return W;
}

// The user's quisk_extern_demod (extdemod.c:13: `int quisk_extern_demod(complex double *cSamples, int nSamples, double decim)`, in place,
// returns the play-sample count), which the reference links in and calls in mode EXT (quisk.c:2490-2493).  NULL unregisters it.
void qh_quisk_set_extern_demod(int (*fn)(double *cSamples, int nSamples, double decim))
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.ext_demod = fn;
}

long long qh_quisk_error_count(void) { std::lock_guard<std::mutex> lk(g.mtx); return g.failed_calls; }
int qh_quisk_squelch_flags(void) { std::lock_guard<std::mutex> lk(g.mtx); return g.squelch_real | (g.squelch_imag << 1); }
// add_tone(freq) (quisk.c:3203-3216): a -40 dB test tone added to the samples; 0 switches it off
void qh_quisk_add_tone(int freq)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.tone_on = freq != 0 && g.sample_rate != 0;
    if (g.tone_on) g.tone_step = turns_step((double)freq, (double)g.sample_rate);
}
// measure_frequency(mode) (quisk.c:3181-3191): mode >= 0 sets measure_freq_mode (0 = off; a result every mode / 2 transforms of
// 12000 samples at decim_srate / 8); returns the last measured frequency
double qh_quisk_measure_frequency(int mode)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    if (mode >= 0) g.measure_mode = mode;
    return g.measured_frequency;
}

// get_graph(1, zoom, deltaf) (quisk.c:5142): data_width pixels in dB and the S-meter; returns the number of FFTs
// averaged (0: nothing new, pixels untouched -- the reference returns None).
int qh_quisk_get_graph(double zoom, double deltaf, double *pixels, double *smeter)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    if (!g.pan) { qh::set_error(QH_ERR_INVALID, "qh_quisk_open was called without a panadapter"); return 0; }
    int count = 0;
    if (qh_pan_graph(g.pan, zoom, deltaf, pixels, smeter, &count)) return 0;
    return count;
}

}  // extern "C"
