/* quisk_rx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the receive path of quisk_process_samples (quisk.c:2289-2742) for one receiver (bank 0):
 *   NCO tune (quisk.c:2477-2488) -> quisk_process_decimate (quisk.c:1673-1846: the SDR-IQ rates, the PlanDecimation
 *   branch and its 6/5 x 4/5 rational stage) -> quisk_process_demodulate (quisk.c:1848-2160: CWL/CWU/LSB/USB/AM/FM/
 *   DGT-U/DGT-L/DGT-IQ/DGT-FM; IMD, FDV-U/L up to the audio, i.e. without the codec) with cRxFilterOut / dRxFilterOut
 *   (quisk.c:1182-1256) -> mono to both channels (quisk.c:2622-2627).
 * process_agc and the squelches and the noise blanker (rank 3) are restated too and off
 * until switched on; auto-notch, test tone and key-down handling are not restated (off by default in the
 * reference).  The stages call the filter.c restatement (quisk_oracle.c), which is
 * pinned bit-exactly to the reference build; the control flow above them is PARITY UNPINNED (quisk.c needs
 * <fftw3.h>, quisk.c:6, and cannot be built here).
 */
#ifndef QUISK_RX_ORACLE_H
#define QUISK_RX_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

enum { QO_CWL = 0, QO_CWU, QO_LSB, QO_USB, QO_AM, QO_FM, QO_EXT, QO_DGT_U, QO_DGT_L, QO_DGT_IQ, QO_IMD, QO_FDV_U,
       QO_FDV_L, QO_DGT_FM };                                   /* rx_mode_type, quisk.h:55-70 */

typedef struct {                /* the filters.h tables the path uses */
    const double *f48dec24;     /* quiskFilt48dec24Coefs[98]      */
    const double *f144d3;       /* quiskFilt144D3Coefs[147]       */
    const double *f240d5;       /* quiskFilt240D5CoefsSharp[245]  */
    const double *audio24p4;    /* quiskAudio24p4Coefs[50]        */
    const double *audio24p6;    /* quiskAudio24p6Coefs[36]        */
    const double *lp48;         /* quiskLpFilt48Coefs[186]        */
    const double *fmhp;         /* quiskAudioFmHpCoefs[309]       */
    const double *f300d5;       /* quiskFilt300D5Coefs[125]       (6/5 stage, quisk.c:1836)      */
    const double *sdriq53;      /* quiskFilt53D1Coefs[55]         (SDR-IQ rates, quisk.c:1706-1710) */
    const double *sdriq111;     /* quiskFilt111D2Coefs[114]       */
    const double *sdriq133;     /* quiskFilt133D2Coefs[136]       */
    const double *sdriq167;     /* quiskFilt167D3Coefs[174]       */
    const double *sdriq185;     /* quiskFilt185D3Coefs[189]       */
} qo_rx_tables;

typedef struct qo_rx qo_rx;
qo_rx *qo_rx_create(int sample_rate, const qo_rx_tables *t);
void qo_rx_set_ssb_squelch(qo_rx *r, int enabled, int level);   /* set_ssb_squelch, quisk.c:4729; CW / SSB / AM */
void qo_rx_set_squelch(qo_rx *r, double level);                 /* set_squelch (FM), quisk.c:4721; default -999 = never */
void qo_rx_set_bandwidth(qo_rx *r, int bw);                     /* filter_bandwidth[0], third argument of set_filters */
void qo_rx_free(qo_rx *r);
void qo_rx_set_tune(qo_rx *r, int rx_tune_freq);                /* set_tune, quisk.c:4702 */
void qo_rx_set_mode(qo_rx *r, int mode);                        /* set_rx_mode, quisk.c:4621 */
void qo_rx_set_filters(qo_rx *r, const double *filtI, const double *filtQ, int size);   /* set_filters, quisk.c:4551 */
/* in place on n interleaved complex samples; returns the number of 48 ksps output samples (may exceed n: the
 * buffer must hold max(n, returned)).  */
int qo_rx_process(qo_rx *r, double *cSamples, int n);
/* process_agc (quisk.c:2162-2287) as quisk_process_samples applies it to the playback stream (quisk.c:2686-2702):
 * off by default here so that the stages can be checked without it; release_gain is set_agc's argument
 * (agcReleaseGain, quisk.c:191,4543). */
void qo_rx_set_agc(qo_rx *r, int on, double release_gain);
/* The AGC alone: state for one stream. */
typedef struct qo_agc qo_agc;
qo_agc *qo_agc_create(int sample_rate, double max_out, double release_time);
void qo_agc_free(qo_agc *a);
void qo_agc_process(qo_agc *a, double *csamples, int count, int is_cpx, double release_gain);
/* NoiseBlanker (quisk.c:680-784) alone, one stream; level 0 = off, 1..3 = limit 6.0 / 4.0 / 2.5; in place.  The
 * output lags the input by qo_nb_delay() = 3 * (int)(sample_rate * 500e-6 + 0.5) samples. */
typedef struct qo_nb qo_nb;
qo_nb *qo_nb_create(int sample_rate);
void qo_nb_free(qo_nb *b);
void qo_nb_set_level(qo_nb *b, int level);
int qo_nb_delay(const qo_nb *b);
void qo_nb_process(qo_nb *b, double *csamples, int count);
/* dAutoNotch (quisk.c:786-963) alone, one real audio stream, in place; `sidetone` = rit_freq in the CW modes else 0,
 * `rate` = quisk_filter_srate.  qo_notch_set is set_auto_notch (quisk.c:4596): stores the flag and re-initialises. */
typedef struct qo_notch qo_notch;
qo_notch *qo_notch_create(void);
void qo_notch_free(qo_notch *a);
void qo_notch_set(qo_notch *a, int on);
void qo_notch_init(qo_notch *a);
void qo_notch_process(qo_notch *a, double *dsamples, int count, int sidetone, int rate);
void qo_rx_set_auto_notch(qo_rx *r, int on, int rit_freq);      /* set_auto_notch + set_sidetone's rit_freq */
void qo_rx_set_noise_blanker(qo_rx *r, int level);              /* set_noise_blanker, quisk.c:4605; runs before the tune */
int qo_rx_decim_srate(const qo_rx *r);
int qo_rx_filter_srate(const qo_rx *r);

/* ---- quisk_process_samples as a whole (quisk.c:2289-2742): key-down replacement, AddTestTone, inversion, NoiseBlanker,
 * FFT ring, banks 0-2 (main receiver; split Rx/Tx or played sub-receiver with Buffer2Chan; sub-receiver 1's digital
 * output), measure_freq, cFracDecim, the wdspFexchange0 hand-off, HB45 interpolation to the playback rate, the AGCs,
 * kill_audio / squelch, the key-up envelope.  One struct instead of the reference's statics and globals. */
struct qo_graph;
struct wo_shim;
typedef struct qo_ps qo_ps;
qo_ps *qo_ps_create(int sample_rate, int playback_rate, const qo_rx_tables *t);
void qo_ps_free(qo_ps *p);
void qo_ps_set_tune(qo_ps *p, int rx_tune_freq, int tx_tune_freq);
void qo_ps_set_mode(qo_ps *p, int mode);
void qo_ps_set_filters(qo_ps *p, const double *fI, const double *fQ, int size, int bandwidth, int nFilter);
void qo_ps_set_agc(qo_ps *p, double level);
void qo_ps_set_split_rxtx(qo_ps *p, int split);
void qo_ps_set_multirx_play_channel(qo_ps *p, int ch);
void qo_ps_set_multirx_play_method(qo_ps *p, int method);
void qo_ps_set_multirx_freq(qo_ps *p, int index, int freq);
void qo_ps_set_multirx_mode(qo_ps *p, int index, int mode);
void qo_ps_set_multirx_count(qo_ps *p, int n);
void qo_ps_set_sub_rx1_output(qo_ps *p, int on);
void qo_ps_multirx_samples(qo_ps *p, int index, const double *x, int n);
void qo_ps_set_key_state(qo_ps *p, int key_down, int cw_key_down, int active_sidetone, int is_fdx);
void qo_ps_set_sidetone(qo_ps *p, double volume, int rit_freq, int txrx_silence_ms);
void qo_ps_set_kill_audio(qo_ps *p, int kill);
void qo_ps_invert_spectrum(qo_ps *p, int invert);
void qo_ps_set_noise_blanker(qo_ps *p, int level);
void qo_ps_set_auto_notch(qo_ps *p, int on);
void qo_ps_set_squelch(qo_ps *p, double level);
void qo_ps_set_ssb_squelch(qo_ps *p, int enabled, int level);
void qo_ps_add_tone(qo_ps *p, int freq);                       /* add_tone, quisk.c:3203 */
double qo_ps_measure_frequency(qo_ps *p, int mode);            /* measure_frequency, quisk.c:3181 */
void qo_ps_set_graph(qo_ps *p, struct qo_graph *g);
void qo_ps_set_wdsp(qo_ps *p, struct wo_shim *s, void (*fn)(void *ctx, double *in, double *out, int *error), void *ctx);
int qo_ps_sub_rx1_audio(qo_ps *p, double *out, int cap);
void qo_ps_restart_bank(qo_ps *p, int bank);                   /* test hook: what a GPU bank rebuild forgets */
int qo_ps_overrun(const qo_ps *p);                             /* 1 once a call has handed Buffer2Chan more than BUF2CHAN_SIZE samples (the reference overruns its arrays there) */
int qo_ps_squelch_flags(const qo_ps *p);                       /* bit 0: squelch_real, bit 1: squelch_imag of the last call */
/* in place; the buffer must hold max(n, output count) samples; returns the count at the playback rate */
int qo_ps_process(qo_ps *p, double *cSamples, int n);

/* get_filter, quisk.c:5481-5568: out[data_width] dB, negative frequencies first */
void qo_get_filter(const double *filtI, const double *filtQ, int sizeFilter, int data_width, int fft_size, double *out);

#ifdef __cplusplus
}
#endif
#endif
