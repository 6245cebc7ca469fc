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
    g.fft_size = fft_size; g.data_width = data_width;
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
    g.bandwidth = bandwidth; g.params_dirty = true;
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

// quisk_process_samples (quisk.c:2289): in place; returns the number of output samples at the playback rate
// (the buffer must have room for them: SAMP_BUFFER_SIZE in the reference); nSamples <= 0 is returned unchanged.
int qh_quisk_process_samples(double *cSamples, int nSamples)
{
    if (nSamples <= 0) return nSamples;                                  // quisk.c:2336-2337
    if (!cSamples) { qh::set_error(QH_ERR_INVALID, "null sample buffer"); return 0; }
    std::lock_guard<std::mutex> lk(g.mtx);
    if (ensure_bank()) return 0;
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
    if (g.agc_on && got > 0) {                                           // process_agc(&Agc1, ...), quisk.c:2686-2702
        const int rate = qh_qrx_decim_rate(g.bank);
        if (g.agc && g.agc_rate != rate) { qh_qagc_destroy(g.agc); g.agc = nullptr; }
        if (!g.agc) {
            g.agc = qh_qagc_create(0, 1, rate, 0.7, 1.0, 0, nullptr);
            if (!g.agc) return 0;
            g.agc_rate = rate;
            g.agc_gain_set = -1.0;
        }
        if (qh_qagc_set_cpx(g.agc, g.mode == 9 /* DGT-IQ */)) return 0;
        if (g.agc_gain_set != g.agc_gain) { if (qh_qagc_set_gain(g.agc, -1, g.agc_gain)) return 0; g.agc_gain_set = g.agc_gain; }
        if (qh_qagc_process_host(g.agc, g.out.data(), got, got)) return 0;
    }
    std::memcpy(cSamples, g.out.data(), (size_t)got * 2 * sizeof(double));
    return got;
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
