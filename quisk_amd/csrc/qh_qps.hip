// qh_qps.hip -- quisk_process_samples (quisk.c:2289-2742) for a BANK of `nch` receivers (include/quiskhip.h group 9b).
//
// The reference's receive function handles one receiver per process (its state is function-static, quisk.c:2301-2321).  Here the
// same function runs for many receivers side by side: every receiver has its own tune frequency, Rx filter and squelch level;
// mode, rates, noise-blanker level, AGC release gain, test tone, inversion and kill_audio are the bank's (they are process-wide
// globals in the reference).  One call = one block per receiver, in the reference's order:
//
//   AddTestTone, spectrum inversion            quisk.c:1258-1303,2438-2446        qh_ps::prep_kernel
//   NoiseBlanker                               quisk.c:680-784,2448-2449          qh_nb (nch streams)
//   FFT ring producer (panadapter)             quisk.c:2454-2475                  qh_pan (nch displays), on the same samples
//   tune + quisk_process_decimate + quisk_process_demodulate, bank 0   quisk.c:2477-2530      qh_qrx (nch receivers)
//   cFracDecim to 48 ksps                      quisk.c:622-665,2654-2659          qh_ps::fracdecim_kernel
//   HB45 interpolation to the playback rate    quisk.c:2663-2682                  qh_rat (one polyphase filter)
//   process_agc                                quisk.c:2162-2287,2686-2702        qh_qagc (nch state machines)
//   kill_audio / squelch                       quisk.c:2712-2728                  qh_ps::epilogue_kernel (only when one can act)
//
// Not here, because they are about ONE operator's transceiver and stay in the one-receiver block API (qh_quisk_rx_compat.cpp,
// which runs the same kernels with nch = 1): the key-down replacement and the key-up envelope, the split / sub-receiver second
// channel with Buffer2Chan, measure_freq, the WDSP hand-off.
//
// Long calls are cut into `pieces` time pieces: process_agc is a chain of dependent instructions per receiver (one wavefront
// each, a fraction of the chip), so piece p's AGC runs on a second stream beside the filters of piece p + 1.
#include <cmath>
#include <mutex>
#include <vector>
#include "qh_internal.hpp"
#include "qh_ps_kernels.hpp"

using qh::set_error;
using qh_ps::u64;

namespace {

template <typename T> struct Buf {              // grows on demand; contents are not kept across a growth
    T *p = nullptr;
    size_t cap = 0;
    int need(size_t n)
    {
        if (n <= cap) return QH_OK;
        if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 8 + 64;
        if (hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess) return set_error(QH_ERR_HIP, "qh_qps: hipMalloc of %zu bytes failed", want * sizeof(T));
        cap = want;
        return QH_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

bool is_fm_mode(int mode) { return mode == 5 || mode == 13; }
bool has_ssb_squelch(int mode) { return mode <= 4 || mode == 10; }

}  // namespace

struct qh_qps {
    std::mutex mtx;
    int device = 0, nch = 0, sample_rate = 0, playback_rate = 48000, mode = 3, bandwidth = 2700, ratio = 1;
    hipStream_t stream = nullptr, agc_stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_piece[2] = { nullptr, nullptr }, ev_agc[2] = { nullptr, nullptr };
    bool agc_recorded[2] = { false, false }, agc_started = false;
    int last_agc_par = -1;                       // parity of the AGC event recorded last, over all calls (the AGC's stream runs in order)
    long long layout[6] = { 0, 0, 0, 0, 0, 0 };  // how the call before this one cut the bank's rows and the scratch halves (pipelined calls)
    qh_qrx *rx = nullptr;
    qh_nb *nb = nullptr;
    qh_pan *pan = nullptr;
    qh_rat *up = nullptr;
    qh_qagc *agc = nullptr;
    int nb_level = 0, invert = 0, kill_audio = 0, pieces = 0;
    bool pipelined = false;                      // qh_qps_set_pipelined: a call does not wait for its own AGC (the next call's filters run beside its tail)
    double agc_gain = 80.0;                      // agcReleaseGain, quisk.c:191
    bool tone_on = false;
    u64 tone_phase = 0, tone_step = 0, audio_phase = 0;
    bool squelch_can_act = false;                // a squelch has been switched on: the epilogue pass runs
    double fd_dindex = 1.0;                      // cFracDecim's static (quisk.c:626-629)
    double2 *fd_hist[2] = { nullptr, nullptr };
    int fd_cur = 0;
    int *d_flags = nullptr;
    std::vector<int> h_flags;
    Buf<double2> d_x, d_nb, d_o, d_fd, d_up;
    long long o_stride = 0, fd_stride = 0, up_stride = 0;

    ~qh_qps()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (agc_stream) (void)hipStreamSynchronize(agc_stream);
        if (rx) qh_qrx_destroy(rx);
        if (nb) qh_nb_destroy(nb);
        if (pan) qh_pan_destroy(pan);
        if (up) qh_rat_destroy(up);
        if (agc) qh_qagc_destroy(agc);
        d_x.release(); d_nb.release(); d_o.release(); d_fd.release(); d_up.release();
        for (double2 *&h : fd_hist) if (h) (void)hipFree(h);
        if (d_flags) (void)hipFree(d_flags);
        for (hipEvent_t &e : ev_piece) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t &e : ev_agc) if (e) (void)hipEventDestroy(e);
        if (agc_stream) (void)hipStreamDestroy(agc_stream);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }

    int decim_rate() const { return qh_qrx_decim_rate(rx); }

    // the most playback-rate samples a call of n inputs can return (the exact count is the call's n_out)
    int out_capacity(int n) const
    {
        long long m = qh_qrx_out_count(rx, n);
        const int dr = decim_rate();
        m += 64;
        if (dr != 48000) m = (long long)((double)(m + 2) * 48000.0 / dr) + 8 * 9;      // cFracDecim rounds per piece
        return (int)(m * ratio);
    }

    // one piece of a call: rows of `in`, n samples each, to rows of the bank-rate buffer at o_off; returns the audio's length there
    int filters(const double2 *in, long long in_stride, int n, long long o_off, int *n_bank)
    {
        const double2 *cur = in;
        long long cs = in_stride;
        if (tone_on || invert) {
            const int kind = !tone_on ? -1 : mode == 4 ? 1 : is_fm_mode(mode) ? 2 : 0;
            const u64 da = qh_ps::turns_step(1000.0, (double)sample_rate);
            hipLaunchKernelGGL(qh_ps::prep_kernel, dim3(qh_ps::grid_x(n, 256u), (unsigned)nch), dim3(256), 0, stream, cur, cs, d_x.p, (long long)n, n, kind,
                               tone_phase, tone_step, audio_phase, da, invert);
            if (tone_on) {
                tone_phase += tone_step * (u64)n;
                if (kind >= 1) audio_phase += da * (u64)n;
            }
            cur = d_x.p; cs = n;
        }
        if (nb_level > 0 || nb) {               // (once created the blanker stays in the path: its delay line is part of the stream)
            if (!nb && !(nb = qh_nb_create(device, nch, sample_rate, stream))) return QH_ERR_HIP;
            if (int rc = qh_nb_set_level(nb, nb_level)) return rc;
            if (int rc = qh_nb_process(nb, cur, cs, d_nb.p, n, n)) return rc;
            cur = d_nb.p; cs = n;
        }
        if (pan) if (int rc = qh_pan_feed(pan, reinterpret_cast<const double *>(cur), cs, n)) return rc;
        return qh_qrx_process(rx, reinterpret_cast<const double *>(cur), cs, n, reinterpret_cast<double *>(d_o.p + o_off), o_stride, n_bank);
    }
};

extern "C" {

qh_qps *qh_qps_create(int device, int nch, int sample_rate, int playback_rate, int mode, int bandwidth, const qh_qrx_tables *tables,
                      int fft_size, int data_width, void *stream)
{
    if (nch <= 0 || sample_rate <= 0 || !tables) { set_error(QH_ERR_INVALID, "qh_qps_create: bad arguments"); return nullptr; }
    const int ratio = playback_rate / 48000;
    if (playback_rate <= 0 || playback_rate % 48000 || (ratio != 1 && ratio != 2 && ratio != 4 && ratio != 8)) {
        set_error(QH_ERR_UNSUPPORTED, "Failure in quisk.c in integer interpolation: playback rate %d is not 48000 x 1, 2, 4 or 8 (quisk.c:2664-2681)", playback_rate);
        return nullptr;
    }
    if (mode == 6) { set_error(QH_ERR_UNSUPPORTED, "mode EXT calls the user's quisk_extern_demod per receiver (extdemod.c:13): the one-receiver block API carries it"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error(QH_ERR_NO_DEVICE, "no HIP device %d (libquiskhip has no CPU fallback)", device);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { set_error(QH_ERR_HIP, "hipSetDevice failed"); return nullptr; }
    qh_qps *h = new qh_qps();
    h->device = device; h->nch = nch; h->sample_rate = sample_rate; h->playback_rate = playback_rate; h->mode = mode; h->bandwidth = bandwidth;
    h->ratio = ratio;
    auto fail = [&](const char *what) { if (what) set_error(QH_ERR_HIP, "qh_qps_create: %s failed", what); delete h; return (qh_qps *)nullptr; };
    h->stream = (hipStream_t)stream;
    if (!h->stream) {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail("stream creation");
        h->own_stream = true;
    }
    // (a CU mask per stream -- hipExtStreamCreateWithCUMask, the AGC on CUs of its own -- was tried: the masks change nothing on this
    // stack, the filters run as fast on "64 CUs" as on 192)
    if (hipStreamCreateWithFlags(&h->agc_stream, hipStreamNonBlocking) != hipSuccess) return fail("stream creation");
    for (hipEvent_t &e : h->ev_piece) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail("event creation");
    for (hipEvent_t &e : h->ev_agc) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail("event creation");
    h->rx = qh_qrx_create_ex(device, nch, sample_rate, mode, bandwidth, tables, h->stream);
    if (!h->rx) return fail(nullptr);
    if (qh_qrx_set_mute_deferred(h->rx, 1)) return fail(nullptr);         // the squelch acts behind the AGC (quisk.c:2712-2728)
    if (fft_size > 0 && data_width > 0) {
        h->pan = qh_pan_create(device, nch, fft_size, data_width, (double)sample_rate, h->stream);
        if (!h->pan) return fail(nullptr);
    }
    if (ratio > 1) {
        const std::vector<double> taps = qh_ps::playback_interp_taps(ratio);
        h->up = qh_rat_create(device, nch, taps.data(), (int)taps.size(), ratio, 1, QH_F64, h->stream);
        if (!h->up) return fail(nullptr);
    }
    // Agc1 = {0.7, 0, 0} (quisk.c:2321): max_out 0.7, the playback rate, release time 1 s (agc_release_time, quisk.c:192)
    h->agc = qh_qagc_create(device, nch, playback_rate, 0.7, 1.0, mode == 9 ? 1 : 0, h->agc_stream);
    if (!h->agc) return fail(nullptr);
    if (qh_qagc_set_gain(h->agc, -1, h->agc_gain)) return fail(nullptr);
    for (int i = 0; i < 2; i++) {
        if (hipMalloc((void **)&h->fd_hist[i], (size_t)nch * 3 * sizeof(double2)) != hipSuccess) return fail("hipMalloc");
        if (qh::dev_zero(h->fd_hist[i], (size_t)nch * 3 * sizeof(double2)) != hipSuccess) return fail("hipMemset");
    }
    if (hipMalloc((void **)&h->d_flags, (size_t)nch * 2 * sizeof(int)) != hipSuccess) return fail("hipMalloc");
    if (qh::dev_zero(h->d_flags, (size_t)nch * 2 * sizeof(int)) != hipSuccess) return fail("hipMemset");
    h->h_flags.assign((size_t)nch * 2, 0);
    return h;
}

void qh_qps_destroy(qh_qps *h) { delete h; }

#define QPS_ENTER(h) if (!(h)) return set_error(QH_ERR_INVALID, "null receiver bank"); std::lock_guard<std::mutex> lk((h)->mtx); QH_HIP(hipSetDevice((h)->device))

int qh_qps_set_tune(qh_qps *h, int ch, int rx_tune_freq) { QPS_ENTER(h); return qh_qrx_set_tune(h->rx, ch, rx_tune_freq); }                 // set_tune, quisk.c:4702
int qh_qps_set_tune_all(qh_qps *h, const int *rx_tune_freq) { QPS_ENTER(h); return qh_qrx_set_tune_all(h->rx, rx_tune_freq); }              // the bank's set_tune in one launch
int qh_qps_set_filters(qh_qps *h, int ch, const double *filtI, const double *filtQ, int size)                                                // set_filters, quisk.c:4551
{
    QPS_ENTER(h);
    return qh_qrx_set_filters(h->rx, ch, filtI, filtQ, size);
}
int qh_qps_set_agc(qh_qps *h, double level)                                                                                                  // set_agc, quisk.c:4543
{
    QPS_ENTER(h);
    h->agc_gain = level;
    return qh_qagc_set_gain(h->agc, -1, level);
}
int qh_qps_set_noise_blanker(qh_qps *h, int level) { QPS_ENTER(h); h->nb_level = level < 0 ? 0 : level; return QH_OK; }                     // quisk.c:4605
int qh_qps_set_auto_notch(qh_qps *h, int on, int rit_freq)                                                                                   // quisk.c:4596
{
    QPS_ENTER(h);
    if (h->mode == 9) return QH_OK;         // DGT-IQ never calls dAutoNotch (quisk.c:2112-2127): the flag is taken and has no effect, as in the reference
    return qh_qrx_set_auto_notch(h->rx, on, rit_freq);
}
int qh_qps_invert_spectrum(qh_qps *h, int invert) { QPS_ENTER(h); h->invert = invert ? 1 : 0; return QH_OK; }                               // quisk.c:4535
int qh_qps_set_kill_audio(qh_qps *h, int kill) { QPS_ENTER(h); h->kill_audio = kill ? 1 : 0; return QH_OK; }
int qh_qps_add_tone(qh_qps *h, int freq)                                                                                                     // add_tone, quisk.c:3203
{
    QPS_ENTER(h);
    h->tone_on = freq != 0;
    if (h->tone_on) h->tone_step = qh_ps::turns_step((double)freq, (double)h->sample_rate);
    return QH_OK;
}
int qh_qps_set_squelch(qh_qps *h, int ch, double level)                                                                                      // set_squelch (FM), quisk.c:4721
{
    QPS_ENTER(h);
    if (ch < -1 || ch >= h->nch) return set_error(QH_ERR_INVALID, "qh_qps_set_squelch: receiver %d of %d", ch, h->nch);
    if (!is_fm_mode(h->mode)) return QH_OK;        // the level is looked at by the FM demodulator alone (quisk.c:2076-2085); a bank keeps its mode
    h->squelch_can_act = true;
    return qh_qrx_set_squelch(h->rx, ch, level);
}
int qh_qps_set_ssb_squelch(qh_qps *h, int enabled, int level)                                                                                // quisk.c:4729
{
    QPS_ENTER(h);
    if (!has_ssb_squelch(h->mode)) return QH_OK;   // ssb_squelch is called by the CW, SSB and AM demodulators alone (quisk.c:1925,1970,2020); a bank keeps its mode
    if (enabled) h->squelch_can_act = true;
    return qh_qrx_set_ssb_squelch(h->rx, enabled, level);
}
int qh_qps_set_pieces(qh_qps *h, int pieces) { QPS_ENTER(h); if (pieces < 0 || pieces > 64) return set_error(QH_ERR_INVALID, "0 (automatic) .. 64 pieces"); h->pieces = pieces; return QH_OK; }
int qh_qps_filter_rate(qh_qps *h) { return h ? qh_qrx_filter_rate(h->rx) : 0; }                                                             // get_filter_rate, quisk.c:2787
int qh_qps_decim_rate(qh_qps *h) { return h ? h->decim_rate() : 0; }
int qh_qps_out_capacity(qh_qps *h, int n_in) { return h && n_in > 0 ? h->out_capacity(n_in) : 0; }

// One block per receiver: d_in [nch][in_stride] complex doubles on the device, n each; the playback samples go to d_out
// [nch][out_stride]; *n_out = their count per receiver (the same for all: one clock).  Asynchronous on the bank's stream.
int qh_qps_process(qh_qps *h, const double *d_in, long long in_stride, int n, double *d_out, long long out_stride, int *n_out)
{
    QPS_ENTER(h);
    if (n_out) *n_out = 0;
    if (n <= 0) return QH_OK;                                           // quisk.c:2336-2337
    if (!d_in || !d_out || in_stride < n) return set_error(QH_ERR_INVALID, "qh_qps_process: bad buffers");
    if (out_stride < h->out_capacity(n)) return set_error(QH_ERR_INVALID, "qh_qps_process: out_stride %lld is shorter than qh_qps_out_capacity = %d", out_stride, h->out_capacity(n));
    const int nch = h->nch;
    const double2 *in = reinterpret_cast<const double2 *>(d_in);
    double2 *out = reinterpret_cast<double2 *>(d_out);
    const int dr = h->decim_rate();
    const bool frac = dr != 48000;
    // pieces: the AGC of one beside the filters of the next.  cFracDecim and the interpolator carry their own phases, so any cut
    // gives the same stream; process_agc's FIRST call only initialises and leaves its whole block alone (quisk.c:2173-2190), so the
    // first call of a bank stays one piece.
    int P = h->pieces;
    if (P <= 0) P = n >= (1 << 15) ? 4 : 1;       // (measured at 256 x 2^20: 1 piece 4.12 ms, 4: 3.60, 8: 3.65, 16: 3.57, 32: 4.0)
    if (!h->agc_started) P = 1;
    // The squelches are the CALL's: the FM squelch averages the level over the block it is given and mutes that block (quisk.c:2076-2085,
    // 2716), ssb_squelch counts its one-second timer down by the block length once per call, its first call only makes the plan, and the
    // flag it leaves mutes the whole block (quisk.c:1104-1112,1173-1176,2712-2728).  A call cut into pieces would decide piece by piece.
    if (h->squelch_can_act) P = 1;
    // (a short first piece, so that the AGC -- the long pole: one dependent chain per receiver -- starts early, was measured: no gain)
    const int per = ((n + P - 1) / P + 63) / 64 * 64;
    // Every piece has a FIXED stretch of the bank's output rows, piece * (its bound): with stretches laid end to end by the counts that
    // came out, a piece's start moved by a sample from call to call whenever the piece length is not a multiple of the bank's decimation
    // (240 ksps / 5 ...), and piece 0 of a pipelined call -- which waits for its own parity's AGC only -- could write into the stretch
    // the other parity's AGC of the call before was still reading (two pieces, no scratch behind the bank).
    const long long bank_piece = (long long)qh_qrx_out_count(h->rx, per) + 64;
    const int cap_bank = (int)(P * bank_piece);
    h->o_stride = cap_bank;
    if (int rc = h->d_o.need((size_t)nch * (size_t)cap_bank)) return rc;
    if (h->tone_on || h->invert) if (int rc = h->d_x.need((size_t)nch * (size_t)per)) return rc;
    if (h->nb_level > 0 || h->nb) if (int rc = h->d_nb.need((size_t)nch * (size_t)per)) return rc;
    // scratch behind the bank, two halves used by alternate pieces (the AGC of piece p reads one while piece p + 1 fills the other)
    const long long fd_bound = frac ? (long long)((double)(bank_piece + 2) * 48000.0 / dr) + 8 : bank_piece;
    const long long up_bound = fd_bound * h->ratio + 64;
    h->fd_stride = fd_bound; h->up_stride = up_bound;
    if (frac) if (int rc = h->d_fd.need((size_t)nch * (size_t)fd_bound * 2)) return rc;
    if (h->up) if (int rc = h->d_up.need((size_t)nch * (size_t)up_bound * 2)) return rc;
    const bool scratch = frac || h->up;
    const bool epi = h->kill_audio || h->squelch_can_act;
    long long o_off = 0, out_off = 0;
    int piece = 0, last_agc = -1;
    if (int rc = qh_qrx_squelch_pieces(h->rx, P > 1 ? 1 : 0)) return rc;       // (0: also ends a call that an error cut short)
    // Pipelined calls: a piece waits for the AGC piece that recorded ITS parity's event last, which covers what it overwrites only while
    // this call cuts the bank's rows and the scratch halves as the call before it did.  Another n, piece count or bound (a squelch
    // switched on, set_pieces, a longer block): piece 0 could run over rows the other parity's AGC piece is still reading -- such a
    // call waits for the AGC event recorded LAST, i.e. for all of the call before it (the AGC's stream runs in order).
    {
        const long long now[6] = { (long long)n, (long long)P, (long long)per, (long long)cap_bank, fd_bound, up_bound };
        bool same = true;
        for (int i = 0; i < 6; i++) same = same && now[i] == h->layout[i];
        if (h->pipelined && !same && h->last_agc_par >= 0) QH_HIP(hipStreamWaitEvent(h->stream, h->ev_agc[h->last_agc_par], 0));
        for (int i = 0; i < 6; i++) h->layout[i] = now[i];
    }
    for (int pos = 0; pos < n; piece++) {
        const int cnt = n - pos < per ? n - pos : per, par = piece & 1;
        // this piece's scratch half was read by the AGC two pieces back.  Pipelined calls: the call before this one may still be in its
        // AGC -- whatever this piece overwrites (its scratch half, its stretch of the bank's output rows) was read by the AGC piece that
        // recorded this parity's event last, or by one ahead of it on the AGC's stream
        if (h->agc_recorded[par] && (h->pipelined || (scratch && piece >= 2))) QH_HIP(hipStreamWaitEvent(h->stream, h->ev_agc[par], 0));
        int nb_ = 0;
        if (int rc = h->filters(in + pos, in_stride, cnt, o_off, &nb_)) return rc;
        const double2 *audio = h->d_o.p + o_off;
        long long as = h->o_stride;
        int na = nb_;
        if (frac && na > 0) {                                           // cFracDecim to 48 ksps (quisk.c:2654-2659)
            const double fdecim = dr / 48000.0;
            const double d0 = h->fd_dindex;
            const int M = qh_ps::fracdecim_walk(na, h->fd_dindex, fdecim);
            if (M > fd_bound) return set_error(QH_ERR_HIP, "qh_qps: cFracDecim count %d above its bound %lld", M, fd_bound);
            double2 *dst = h->d_fd.p + (size_t)par * (size_t)nch * (size_t)fd_bound;
            if (M > 0) hipLaunchKernelGGL(qh_ps::fracdecim_kernel, dim3(qh_ps::grid_x(M, 256u), (unsigned)nch), dim3(256), 0, h->stream, audio, as,
                                          (const double2 *)h->fd_hist[h->fd_cur], M, d0, fdecim - 1, dst, h->fd_stride, na);
            hipLaunchKernelGGL(qh_ps::fd_hist_kernel, dim3((unsigned)nch), dim3(64), 0, h->stream, audio, as, na, (const double2 *)h->fd_hist[h->fd_cur],
                               h->fd_hist[h->fd_cur ^ 1]);
            h->fd_cur ^= 1;
            audio = dst; as = h->fd_stride; na = M;
        }
        if (h->up && na > 0) {                                          // to the playback rate (quisk.c:2663-2682)
            if ((long long)qh_rat_out_count(h->up, na) > up_bound) return set_error(QH_ERR_HIP, "qh_qps: interpolator count above its bound");
            double2 *dst = h->d_up.p + (size_t)par * (size_t)nch * (size_t)up_bound;
            int got = 0;
            if (int rc = qh_rat_process(h->up, audio, as, na, dst, h->up_stride, &got)) return rc;
            audio = dst; as = h->up_stride; na = got;
        }
        if (na > 0) {
            // process_agc on the second stream, from the piece's audio into the caller's rows (quisk.c:2686-2702)
            if (out_off + na > out_stride) return set_error(QH_ERR_INVALID, "qh_qps_process: the output rows are too short");
            QH_HIP(hipEventRecord(h->ev_piece[par], h->stream));
            QH_HIP(hipStreamWaitEvent(h->agc_stream, h->ev_piece[par], 0));
            if (int rc = qh_qagc_process2(h->agc, audio, as, out + out_off, out_stride, na)) return rc;
            QH_HIP(hipEventRecord(h->ev_agc[par], h->agc_stream));
            h->agc_recorded[par] = true;
            h->agc_started = true;              // (process_agc's initialising call has happened: only now may a call be cut into pieces)
            last_agc = par; h->last_agc_par = par;
        }
        if (nb_ > bank_piece) return set_error(QH_ERR_HIP, "qh_qps: a piece's bank output %d above its bound %lld", nb_, bank_piece);
        o_off += bank_piece;
        out_off += na;
        pos += cnt;
    }
    if (P > 1) if (int rc = qh_qrx_squelch_pieces(h->rx, 0)) return rc;      // (the reference's one look at the FM squelch's count, behind the last piece)
    // the call ends on the bank's stream (the AGC stream runs in order) -- unless it is pipelined: then its output is complete at
    // qh_qps_synchronize (or behind the next call's AGC), and the next call's filters start beside this call's last AGC piece
    hipStream_t tail = h->stream;
    if (h->pipelined && last_agc >= 0) tail = h->agc_stream;
    else if (last_agc >= 0) QH_HIP(hipStreamWaitEvent(h->stream, h->ev_agc[last_agc], 0));
    const int total = (int)out_off;
    if (epi && total > 0) {                                             // kill_audio / squelch (quisk.c:2712-2728); no key here: no envelope
        const int *f0 = qh_qrx_squelch_flag(h->rx, 0), *f1 = nch > 1 ? qh_qrx_squelch_flag(h->rx, 1) : f0;
        const int step = f0 && f1 ? (int)(f1 - f0) : 0;
        if (tail != h->stream) {        // (the flags are the bank's, written on its stream)
            QH_HIP(hipEventRecord(h->ev_piece[0], h->stream));
            QH_HIP(hipStreamWaitEvent(tail, h->ev_piece[0], 0));
        }
        hipLaunchKernelGGL(qh_ps::epilogue_kernel, dim3(qh_ps::grid_x(total, 256u), (unsigned)nch), dim3(256), 0, tail, (const double2 *)out, out_stride,
                           out, out_stride, total, f0, step, f0, step, h->kill_audio, 1.0, 0.0, 0, h->d_flags);
        if (tail != h->stream) { QH_HIP(hipEventRecord(h->ev_agc[last_agc], tail)); h->last_agc_par = last_agc; }      // (what waits for this parity's AGC also waits for the epilogue)
    }
    if (n_out) *n_out = total;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

int qh_qps_synchronize(qh_qps *h)
{
    QPS_ENTER(h);
    QH_HIP(hipStreamSynchronize(h->stream));
    QH_HIP(hipStreamSynchronize(h->agc_stream));
    return QH_OK;
}

// Pipelined calls, for a caller that streams block after block: a call returns with its AGC still running on the bank's second stream
// and the next call's filters start beside it; a call's output rows are complete after qh_qps_synchronize (not in the order of the
// stream the bank was given).  Same samples either way.
int qh_qps_set_pipelined(qh_qps *h, int on)
{
    QPS_ENTER(h);
    if (!on && h->pipelined) { QH_HIP(hipStreamSynchronize(h->stream)); QH_HIP(hipStreamSynchronize(h->agc_stream)); }
    h->pipelined = on != 0;
    return QH_OK;
}

int qh_qps_process_host(qh_qps *h, const double *h_in, long long in_stride, int n, double *h_out, long long out_stride, int *n_out)
{
    if (!h) return set_error(QH_ERR_INVALID, "null receiver bank");
    if (n_out) *n_out = 0;
    if (n <= 0) return QH_OK;
    if (!h_in || !h_out) return set_error(QH_ERR_INVALID, "null buffer");
    QH_HIP(hipSetDevice(h->device));
    int cap = 0;
    { std::lock_guard<std::mutex> lk(h->mtx); cap = h->out_capacity(n); }       // (a setter on another thread may be rebuilding the bank)
    double2 *din = nullptr, *dout = nullptr;
    QH_HIP(hipMalloc((void **)&din, (size_t)h->nch * (size_t)n * 16));
    if (hipMalloc((void **)&dout, (size_t)h->nch * (size_t)cap * 16) != hipSuccess) { (void)hipFree(din); return set_error(QH_ERR_HIP, "hipMalloc failed"); }
    hipError_t e = hipMemcpy2D(din, (size_t)n * 16, h_in, (size_t)in_stride * 16, (size_t)n * 16, (size_t)h->nch, hipMemcpyHostToDevice);
    int got = 0, rc = QH_OK;
    if (e == hipSuccess) rc = qh_qps_process(h, reinterpret_cast<const double *>(din), n, n, reinterpret_cast<double *>(dout), cap, &got);
    if (e == hipSuccess && rc == QH_OK) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && rc == QH_OK) e = hipStreamSynchronize(h->agc_stream);
    if (e == hipSuccess && rc == QH_OK && got > 0) {
        if (out_stride < got) rc = set_error(QH_ERR_INVALID, "qh_qps_process_host: out_stride %lld < %d samples", out_stride, got);
        else e = hipMemcpy2D(h_out, (size_t)out_stride * 16, dout, (size_t)cap * 16, (size_t)got * 16, (size_t)h->nch, hipMemcpyDeviceToHost);
    }
    (void)hipFree(din); (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess) return set_error(QH_ERR_HIP, "qh_qps_process_host: copy failed");
    if (n_out) *n_out = got;
    return QH_OK;
}

// squelch_real of every receiver as the last call left it (quisk.c:2712-2728); flags[nch]
int qh_qps_squelch_flags(qh_qps *h, int *flags)
{
    QPS_ENTER(h);
    if (!flags) return set_error(QH_ERR_INVALID, "null flags");
    QH_HIP(hipStreamSynchronize(h->agc_stream));           // (a pipelined call's epilogue runs there)
    QH_HIP(hipMemcpyAsync(h->h_flags.data(), h->d_flags, (size_t)h->nch * 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    QH_HIP(hipStreamSynchronize(h->stream));
    for (int c = 0; c < h->nch; c++) flags[c] = h->h_flags[(size_t)2 * c];
    return QH_OK;
}

// get_graph(1, zoom, deltaf) (quisk.c:5142) of every receiver: pixels [nch][data_width] dB, smeter [nch]; *count = FFTs averaged (0: nothing new)
int qh_qps_get_graph(qh_qps *h, double zoom, double deltaf, double *pixels, double *smeter, int *count)
{
    QPS_ENTER(h);
    if (!h->pan) return set_error(QH_ERR_INVALID, "the bank was created without a panadapter");
    return qh_pan_graph(h->pan, zoom, deltaf, pixels, smeter, count);
}

}  // extern "C"
