// qh_quisk_rx_compat.cpp -- the Quisk native block API for ONE receiver (include/quiskhip.h group 9).
//
// quisk.c keeps its receive path behind `int quisk_process_samples(complex double *cSamples, int nSamples)`
// (quisk.h:375, quisk.c:2289): in place, returns the output count, parameters arrive through globals that the
// GUI thread sets with set_tune / set_rx_mode / set_filters / set_agc (quisk.c:4702,4621,4551,4543) and reads with
// get_filter_rate / get_graph (quisk.c:2787,5142).  This layer offers exactly that shape -- one process-wide
// receiver, the same call names with a qh_quisk_ prefix -- on top of the batched GPU bank (qh_qrx.hip), the AGC
// (qh_qagc.hip) and the panadapter (qh_pan.hip).  A maintainer replaces the body of quisk_process_samples with a
// call to qh_quisk_process_samples.  Mode, bandwidth class or rate changes rebuild
// the bank (filter histories restart: a few ms of transient where the reference keeps its static histories).
#include <cmath>
#include <complex>
#include <cstring>
#include <mutex>
#include <vector>
#include "qh_internal.hpp"

namespace {

struct QuiskRx {
    std::mutex mtx;
    int sample_rate = 0, mode = 3 /* USB */, tune = 0, bandwidth = 2700;
    double agc_gain = 80.0;                     // agcReleaseGain, quisk.c:191
    bool agc_on = true;
    std::vector<double> tables[13];
    qh_qrx_tables t{};
    bool have_tables = false;
    std::vector<double> filtI, filtQ;
    qh_qrx *bank = nullptr;
    int bank_mode = -1, bank_bw_class = -1, bank_rate = 0;
    bool params_dirty = true;
    // panadapter (record_app's fft_size / data_width, quisk.c:5946)
    qh_pan *pan = nullptr;
    int fft_size = 0, data_width = 0;
    std::vector<double> out;
    // NoiseBlanker's statics (quisk.c:682-687): outlive mode changes, so they are not the bank's
    // Agc1 (quisk.c:2321) is one static AGC for the playback stream whatever the mode: it belongs here, not to a bank
    qh_qagc *agc = nullptr;
    int agc_rate = 0;
    double agc_gain_set = -1.0;
    qh_nb *nb = nullptr;
    int nb_level = 0;
    int auto_notch = 0, rit_freq = 0, notch_applied = -1;
    std::vector<double> nb_out;
    // ---- the rest of quisk_process_samples' orchestration (quisk.c:2289-2742) ----
    // A second receiver bank on another frequency: split Rx/Tx (the same samples at quisk_tx_tune_freq + rit_freq, the
    // filters of nFilter 0) or the played sub-receiver (its own samples, frequency and mode, the filters of nFilter 1);
    // the two audio streams go to the real and the imaginary output and get an AGC each (Agc1, Agc2).
    int tx_tune = 0, split_rxtx = 0;                 // set_tune's second argument, set_split_rxtx (quisk.c:4702,4694)
    int play_channel = -1, play_method = 0;          // set_multirx_play_channel / _method (quisk.c:4856,4846)
    static constexpr int kMaxSub = 9;                // QUISK_MAX_SUB_RECEIVERS, quisk.h
    int sub_freq[kMaxSub] = { 0 }, sub_mode[kMaxSub] = { 0 };      // set_multirx_freq / set_multirx_mode (quisk.c:4826,4836)
    std::vector<double> sub_samples;                 // the played sub-receiver's block for the coming call (multirx_cSamples[])
    int sub_have = 0;
    std::vector<double> filt2I, filt2Q;              // set_filters(..., nFilter = 1)
    int bandwidth2 = 2700;
    qh_qrx *bank2 = nullptr;
    int bank2_mode = -1, bank2_cls = -1, bank2_tune = 0x7fffffff;
    bool filt2_dirty = true;
    qh_qagc *agc2 = nullptr;
    int agc2_rate = 0;
    double agc2_gain_set = -1.0;
    std::vector<double> out2, chan_a, chan_b;
    int old_split = 0, old_play = -1;
    // key handling (quisk.c:2368-2433): the block is replaced by the sidetone or by silence while the key is down and
    // for TxRxSilenceMsec after it, then the volume comes back over 5 ms
    int key_down = 0, cw_key_down = 0, active_sidetone = 0, is_fdx = 0, kill_audio = 0, invert_spectrum = 0;
    int playback_rate = 48000, txrx_silence_ms = 50;
    double sidetone_volume = 0.0;
    std::complex<double> sidetone_phase{ 1.0, 0.0 }, sidetone_vec{ 0.0, 0.0 };
    double out_counter = 0.0, sidetone_env = 0.0, keyup_env = 1.0;
    int sidetone_on = 0, play_silence = 0;
};

QuiskRx g;

int bw_class(int mode, int bw)     // what of the bandwidth the bank's structure depends on (quisk.c:2089,2143)
{
    if (mode == 7 || mode == 8 || mode == 11 || mode == 12) return bw < 3000 ? 0 : 1;
    if (mode == 9) return bw < 19000 ? 0 : 1;
    return 0;
}

int ensure_bank()
{
    if (!g.sample_rate || !g.have_tables) return qh::set_error(QH_ERR_INVALID, "qh_quisk_open has not been called");
    const int cls = bw_class(g.mode, g.bandwidth);
    if (!g.bank || g.bank_mode != g.mode || g.bank_bw_class != cls || g.bank_rate != g.sample_rate) {
        if (g.bank) { qh_qrx_destroy(g.bank); g.bank = nullptr; }
        g.bank = qh_qrx_create_ex(0, 1, g.sample_rate, g.mode, g.bandwidth, &g.t, nullptr);
        if (!g.bank) return QH_ERR_HIP;
        g.bank_mode = g.mode; g.bank_bw_class = cls; g.bank_rate = g.sample_rate;
        g.params_dirty = true;
        g.notch_applied = -1;
    }
    if (g.params_dirty) {
        if (int rc = qh_qrx_set_tune(g.bank, 0, g.tune)) return rc;
        if (int rc = qh_qrx_set_filters(g.bank, 0, g.filtI.data(), g.filtQ.data(), (int)g.filtI.size())) return rc;
        g.params_dirty = false;
    }
    if (g.notch_applied != g.auto_notch) {          // a set_auto_notch call (or a fresh bank) starts the notch over
        if (g.mode != 9 /* DGT-IQ has no notch */) if (int rc = qh_qrx_set_auto_notch(g.bank, g.auto_notch, g.rit_freq)) return rc;
        g.notch_applied = g.auto_notch;
    }
    return QH_OK;
}

}  // namespace

extern "C" {

// quisk_sound_state.sample_rate + the filters.h tables (data, passed in like to qh_qrx_create_ex) + record_app's
// fft_size and data_width (0, 0: no panadapter).
int qh_quisk_open(int sample_rate, const qh_qrx_tables *tables, int fft_size, int data_width)
{
    static const int len[13] = { 98, 147, 245, 50, 36, 186, 309, 125, 55, 114, 136, 174, 189 };
    if (sample_rate <= 0 || !tables) return qh::set_error(QH_ERR_INVALID, "qh_quisk_open: bad arguments");
    std::lock_guard<std::mutex> lk(g.mtx);
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
    g.sample_rate = sample_rate;
    if (g.bank) { qh_qrx_destroy(g.bank); g.bank = nullptr; }
    if (g.pan) { qh_pan_destroy(g.pan); g.pan = nullptr; }
    if (g.nb) { qh_nb_destroy(g.nb); g.nb = nullptr; }      // "sample_rate != sample_rate: Initialization", quisk.c:697
    if (g.agc) { qh_qagc_destroy(g.agc); g.agc = nullptr; }
    if (g.bank2) { qh_qrx_destroy(g.bank2); g.bank2 = nullptr; }
    if (g.agc2) { qh_qagc_destroy(g.agc2); g.agc2 = nullptr; }
    g.fft_size = fft_size; g.data_width = data_width;
    // the orchestration state starts like the reference's statics and globals at program start
    g.tx_tune = 0; g.split_rxtx = 0; g.play_channel = -1; g.play_method = 0; g.sub_have = 0; g.old_split = 0; g.old_play = -1;
    for (int i = 0; i < QuiskRx::kMaxSub; i++) { g.sub_freq[i] = 0; g.sub_mode[i] = 0; }
    g.filt2I.clear(); g.filt2Q.clear(); g.filt2_dirty = true;
    g.key_down = g.cw_key_down = g.active_sidetone = g.is_fdx = g.kill_audio = g.invert_spectrum = 0;
    g.out_counter = 0.0; g.sidetone_env = 0.0; g.keyup_env = 1.0; g.sidetone_on = 0; g.play_silence = 0;
    if (fft_size > 0 && data_width > 0) {
        g.pan = qh_pan_create(0, 1, fft_size, data_width, (double)sample_rate, nullptr);
        if (!g.pan) return QH_ERR_HIP;
    }
    return ensure_bank();
}

void qh_quisk_close(void)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    if (g.bank) { qh_qrx_destroy(g.bank); g.bank = nullptr; }
    if (g.pan) { qh_pan_destroy(g.pan); g.pan = nullptr; }
    if (g.nb) { qh_nb_destroy(g.nb); g.nb = nullptr; }
    if (g.agc) { qh_qagc_destroy(g.agc); g.agc = nullptr; }
    if (g.bank2) { qh_qrx_destroy(g.bank2); g.bank2 = nullptr; }
    if (g.agc2) { qh_qagc_destroy(g.agc2); g.agc2 = nullptr; }
    g.sample_rate = 0;
}

void qh_quisk_set_tune(int rx_tune_freq)            // set_tune, quisk.c:4702
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.tune = rx_tune_freq; g.params_dirty = true;
}

void qh_quisk_set_rx_mode(int mode)                 // set_rx_mode, quisk.c:4621
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.mode = mode;
}

// set_filters(filterI, filterQ, bandwidth, start_offset, nFilter = 0), quisk.c:4551
int qh_quisk_set_filters(const double *filtI, const double *filtQ, int size, int bandwidth)
{
    if (size < 0 || size >= 10001 || (size > 0 && (!filtI || !filtQ)))
        return qh::set_error(QH_ERR_INVALID, "Filter size must be less than 10001");    // MAX_FILTER_SIZE, quisk.c:4576
    std::lock_guard<std::mutex> lk(g.mtx);
    g.filtI.assign(filtI, filtI + size); g.filtQ.assign(filtQ, filtQ + size);
    g.bandwidth = bandwidth; g.params_dirty = true; g.filt2_dirty = true;       // split Rx/Tx runs bank 1 with these too
    return QH_OK;
}

void qh_quisk_set_agc(double level)                 // set_agc, quisk.c:4543
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.agc_gain = level; g.params_dirty = true;
}

void qh_quisk_set_auto_notch(int on, int rit_freq)   // set_auto_notch, quisk.c:4596: the flag, and dAutoNotch(NULL, ...)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.auto_notch = on ? 1 : 0; g.rit_freq = rit_freq; g.notch_applied = -1;
}

void qh_quisk_set_noise_blanker(int level)          // set_noise_blanker, quisk.c:4605
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.nb_level = level < 0 ? 0 : level;
}

int qh_quisk_get_filter_rate(void)                  // get_filter_rate(-1, 0): the rate the current Rx filter runs at
{
    std::lock_guard<std::mutex> lk(g.mtx);
    if (ensure_bank()) return 0;
    return qh_qrx_filter_rate(g.bank);
}

// ---- the second receiver bank (split Rx/Tx, played sub-receiver) -------------------------------------------------------
static int ensure_bank2(int mode, int bandwidth, const std::vector<double> &fI, const std::vector<double> &fQ, int tune, bool filt_dirty)
{
    const int cls = bw_class(mode, bandwidth);
    if (!g.bank2 || g.bank2_mode != mode || g.bank2_cls != cls) {
        if (g.bank2) { qh_qrx_destroy(g.bank2); g.bank2 = nullptr; }
        g.bank2 = qh_qrx_create_ex(0, 1, g.sample_rate, mode, bandwidth, &g.t, nullptr);
        if (!g.bank2) return QH_ERR_HIP;
        g.bank2_mode = mode; g.bank2_cls = cls; g.bank2_tune = 0x7fffffff;
        filt_dirty = true;
    }
    if (g.bank2_tune != tune) { if (int rc = qh_qrx_set_tune(g.bank2, 0, tune)) return rc; g.bank2_tune = tune; }
    if (filt_dirty) if (int rc = qh_qrx_set_filters(g.bank2, 0, fI.data(), fQ.data(), (int)fI.size())) return rc;
    return QH_OK;
}

static int agc_run(qh_qagc *&agc, int &agc_rate, double &gain_set, int rate, int is_cpx, double *buf, int n)
{
    if (agc && agc_rate != rate) { qh_qagc_destroy(agc); agc = nullptr; }
    if (!agc) {
        agc = qh_qagc_create(0, 1, rate, 0.7, 1.0, 0, nullptr);       // struct AgcState {0.7, 0, 0}, quisk.c:2321
        if (!agc) return QH_ERR_HIP;
        agc_rate = rate; gain_set = -1.0;
    }
    if (int rc = qh_qagc_set_cpx(agc, is_cpx)) return rc;
    if (gain_set != g.agc_gain) { if (int rc = qh_qagc_set_gain(agc, -1, g.agc_gain)) return rc; gain_set = g.agc_gain; }
    return qh_qagc_process_host(agc, buf, n, n);
}

// The key is down (or was a moment ago): the block is not demodulated; sidetone or silence at the playback rate take its
// place (quisk.c:2368-2433).  Returns the number of samples written, or -1 when the block is radio sound.
static int key_block(double *cSamples, int nSamples)
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

// quisk_process_samples (quisk.c:2289): in place; returns the number of output samples at the playback rate
// (the buffer must have room for them: SAMP_BUFFER_SIZE in the reference); nSamples <= 0 is returned unchanged.
int qh_quisk_process_samples(double *cSamples, int nSamples)
{
    if (nSamples <= 0) return nSamples;                                  // quisk.c:2336-2337
    if (!cSamples) { qh::set_error(QH_ERR_INVALID, "null sample buffer"); return 0; }
    std::lock_guard<std::mutex> lk(g.mtx);
    if (ensure_bank()) return 0;
    {
        const int kb = key_block(cSamples, nSamples);
        if (kb >= 0) { g.sub_have = 0; return kb; }
    }
    if (g.invert_spectrum)                                               // quisk.c:2441-2446
        for (int i = 0; i < nSamples; i++) cSamples[2 * i + 1] = -cSamples[2 * i + 1];
    if (g.nb_level > 0 || g.nb) {                                        // NoiseBlanker(cSamples, nSamples), quisk.c:2448-2449
        if (!g.nb && !(g.nb = qh_nb_create(0, 1, g.sample_rate, nullptr))) return 0;
        if (qh_nb_set_level(g.nb, g.nb_level)) return 0;
        g.nb_out.resize((size_t)nSamples * 2);
        if (qh_nb_process_host(g.nb, cSamples, nSamples, g.nb_out.data(), nSamples, nSamples)) return 0;
        std::memcpy(cSamples, g.nb_out.data(), (size_t)nSamples * 2 * sizeof(double));
    }
    if (g.pan && qh_pan_feed_host(g.pan, cSamples, nSamples, nSamples)) return 0;       // the FFT ring producer, quisk.c:2454-2475
    const int cap = qh_qrx_out_count(g.bank, nSamples);
    g.out.resize((size_t)(cap > 0 ? cap : 1) * 2);
    int got = 0;
    if (qh_qrx_process_host(g.bank, cSamples, nSamples, nSamples, g.out.data(), cap > 0 ? cap : 1, &got)) return 0;
    const int rate = qh_qrx_decim_rate(g.bank);
    const bool stereo_mode = g.mode == 9 || g.mode == 6;                 // DGT-IQ, EXT: already stereo (quisk.c:2536,2687)
    // ---- a second channel: the same receiver on the transmit frequency, or the played sub-receiver (quisk.c:2539-2621)
    bool two = false;
    if (!stereo_mode && g.split_rxtx) {
        if (ensure_bank2(g.mode, g.bandwidth, g.filtI, g.filtQ, g.tx_tune + g.rit_freq, g.filt2_dirty || !g.old_split)) return 0;
        two = true;
    } else if (!stereo_mode && g.play_channel >= 0 && g.sub_have == nSamples) {
        const int pc = g.play_channel;
        if (g.filt2I.empty()) { qh::set_error(QH_ERR_INVALID, "played sub-receiver without set_filters(..., nFilter = 1)"); return 0; }
        if (ensure_bank2(g.sub_mode[pc], g.bandwidth2, g.filt2I, g.filt2Q, g.sub_freq[pc], g.filt2_dirty || g.old_play != pc)) return 0;
        two = true;
    }
    g.filt2_dirty = false;
    g.old_split = g.split_rxtx; g.old_play = g.play_channel;
    if (two) {
        const double *src2 = g.split_rxtx ? cSamples : g.sub_samples.data();
        g.out2.resize((size_t)(cap > 0 ? cap : 1) * 2);
        int got2 = 0;
        if (qh_qrx_process_host(g.bank2, src2, nSamples, nSamples, g.out2.data(), cap > 0 ? cap : 1, &got2)) return 0;
        if (got2 < got) got = got2;             // (Buffer2Chan, quisk.c:1577-1611: the banks here return equal counts)
        // which stream is the real (left) output
        int first_is_real = 1, both = 0;        // both: 1 = bank 0 on both channels, 2 = bank 1 on both
        if (g.split_rxtx) {
            switch (g.split_rxtx) {
            default:
            case 1: first_is_real = g.tx_tune < g.tune; break;                 // higher frequency is real
            case 2: first_is_real = g.tx_tune >= g.tune; break;                // lower frequency is real
            case 3: both = 1; break;
            case 4: both = 2; break;
            }
        } else {
            switch (g.play_method) {
            default:
            case 0: both = 2; break;
            case 1: first_is_real = 1; break;
            case 2: first_is_real = 0; break;
            }
        }
        for (int i = 0; i < got; i++) {
            const double d = g.out[2 * (size_t)i], d2 = g.out2[2 * (size_t)i];
            const double re = both == 1 ? d : both == 2 ? d2 : first_is_real ? d : d2;
            const double im = both == 1 ? d : both == 2 ? d2 : first_is_real ? d2 : d;
            g.out[2 * (size_t)i] = re; g.out[2 * (size_t)i + 1] = im;
        }
    }
    g.sub_have = 0;
    // ---- AGC (quisk.c:2686-2702)
    if (g.agc_on && got > 0) {
        if (stereo_mode) {
            if (agc_run(g.agc, g.agc_rate, g.agc_gain_set, rate, 1, g.out.data(), got)) return 0;
        } else if (g.split_rxtx || g.play_channel >= 0) {        // separate AGC for left and right
            g.chan_a.assign((size_t)got * 2, 0.0); g.chan_b.assign((size_t)got * 2, 0.0);
            for (int i = 0; i < got; i++) { g.chan_a[2 * (size_t)i] = g.out[2 * (size_t)i]; g.chan_b[2 * (size_t)i] = g.out[2 * (size_t)i + 1]; }
            if (agc_run(g.agc, g.agc_rate, g.agc_gain_set, rate, 0, g.chan_a.data(), got)) return 0;
            if (agc_run(g.agc2, g.agc2_rate, g.agc2_gain_set, rate, 0, g.chan_b.data(), got)) return 0;
            for (int i = 0; i < got; i++) { g.out[2 * (size_t)i] = g.chan_a[2 * (size_t)i]; g.out[2 * (size_t)i + 1] = g.chan_b[2 * (size_t)i]; }
        } else {
            if (agc_run(g.agc, g.agc_rate, g.agc_gain_set, rate, 0, g.out.data(), got)) return 0;
        }
    }
    if (g.kill_audio) std::memset(g.out.data(), 0, (size_t)got * 2 * sizeof(double));       // quisk.c:2712-2716
    if (g.keyup_env < 1.0) {                                             // raise the volume slowly after the key goes up, quisk.c:2729-2738
        const double di = 1.0 / (g.playback_rate * 5e-3);
        for (int i = 0; i < got; i++) {
            g.keyup_env += di;
            if (g.keyup_env > 1.0) { g.keyup_env = 1.0; break; }
            g.out[2 * (size_t)i] *= g.keyup_env; g.out[2 * (size_t)i + 1] *= g.keyup_env;
        }
    }
    std::memcpy(cSamples, g.out.data(), (size_t)got * 2 * sizeof(double));
    return got;
}

// ---- the setters of the orchestration state (names after the QS calls they mirror)
void qh_quisk_set_tx_tune(int tx_tune_freq) { std::lock_guard<std::mutex> lk(g.mtx); g.tx_tune = tx_tune_freq; }           // set_tune's 2nd argument, quisk.c:4702
void qh_quisk_set_split_rxtx(int split) { std::lock_guard<std::mutex> lk(g.mtx); g.split_rxtx = split; }                  // quisk.c:4694
void qh_quisk_set_multirx_play_channel(int ch) { std::lock_guard<std::mutex> lk(g.mtx); g.play_channel = ch >= QuiskRx::kMaxSub ? -1 : ch; }    // quisk.c:4856
void qh_quisk_set_multirx_play_method(int m) { std::lock_guard<std::mutex> lk(g.mtx); g.play_method = m; }                 // quisk.c:4846
void qh_quisk_set_multirx_freq(int index, int freq) { std::lock_guard<std::mutex> lk(g.mtx); if (index >= 0 && index < QuiskRx::kMaxSub) g.sub_freq[index] = freq; }   // quisk.c:4826
void qh_quisk_set_multirx_mode(int index, int mode) { std::lock_guard<std::mutex> lk(g.mtx); if (index >= 0 && index < QuiskRx::kMaxSub) g.sub_mode[index] = mode; }   // quisk.c:4836
// the played sub-receiver's samples for the coming qh_quisk_process_samples call (the reference's sample source fills
// multirx_cSamples[index] with as many samples as the main receiver gets)
int qh_quisk_multirx_samples(int index, const double *cSamples, int nSamples)
{
    if (!cSamples || nSamples <= 0) return qh::set_error(QH_ERR_INVALID, "qh_quisk_multirx_samples: no samples");
    std::lock_guard<std::mutex> lk(g.mtx);
    if (index != g.play_channel) return QH_OK;      // only the played sub-receiver is demodulated on this path
    g.sub_samples.assign(cSamples, cSamples + 2 * (size_t)nSamples);
    g.sub_have = nSamples;
    return QH_OK;
}
// set_filters(filterI, filterQ, bandwidth, start_offset, nFilter = 1): the played sub-receiver's filter (quisk.c:4551)
int qh_quisk_set_filters2(const double *filtI, const double *filtQ, int size, int bandwidth)
{
    if (size <= 0 || size >= 10001 || !filtI || !filtQ) return qh::set_error(QH_ERR_INVALID, "Filter size must be less than 10001");
    std::lock_guard<std::mutex> lk(g.mtx);
    g.filt2I.assign(filtI, filtI + size); g.filt2Q.assign(filtQ, filtQ + size);
    g.bandwidth2 = bandwidth; g.filt2_dirty = true;
    return QH_OK;
}
// quisk_is_key_down() / QUISK_CWKEY_DOWN / quisk_active_sidetone / quisk_isFDX as the caller sees them now; the sidetone
// of set_sidetone (quisk.c:4710: volume, |rit_freq| as its pitch) at the playback rate of open_sound
void qh_quisk_set_key_state(int key_down, int cw_key_down, int active_sidetone, int is_fdx)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.key_down = key_down; g.cw_key_down = cw_key_down; g.active_sidetone = active_sidetone; g.is_fdx = is_fdx;
}
void qh_quisk_set_sidetone(double volume, int rit_freq, int playback_rate, int txrx_silence_msec)
{
    std::lock_guard<std::mutex> lk(g.mtx);
    g.sidetone_volume = volume; g.rit_freq = rit_freq; g.notch_applied = -1;
    if (playback_rate > 0) g.playback_rate = playback_rate;
    if (txrx_silence_msec >= 0) g.txrx_silence_ms = txrx_silence_msec;
    g.sidetone_phase = std::exp(std::complex<double>(0.0, 2.0 * 3.14159265358979323846 * std::abs(rit_freq) / g.playback_rate));
}
void qh_quisk_set_kill_audio(int kill) { std::lock_guard<std::mutex> lk(g.mtx); g.kill_audio = kill; }
void qh_quisk_invert_spectrum(int invert) { std::lock_guard<std::mutex> lk(g.mtx); g.invert_spectrum = invert; }           // quisk.c:4535

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
