/* quisk_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, double precision, reentrant) of Quisk's native receive
 * DSP primitives (filter.c) and of the parts of quisk.c that sit on the hot path
 * (NCO tune, Rx filter, SSB/AM/FM detectors, panadapter).  Citations are
 * file:line relative to /root/reference.
 *
 * Pinning:
 *  - the filter.c primitives (qo_c* / qo_d* below) ARE pinned: oracle/Makefile's
 *    `ref` target compiles the reference's own filter.c into oracle/_ref/ and
 *    tests/test_oracle_filter.py requires bit-identical outputs on seeded inputs;
 *    tests/golden/filter_*.npz hold vectors generated from that build.
 *  - the quisk.c restatements (qo_tune, qo_rxfilter, qo_graph ...) are PARITY UNPINNED
 *    by reference execution: quisk.c includes <fftw3.h> (quisk.c:6) which this image
 *    lacks, so it cannot be built here under the rules.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef QUISK_ORACLE_H
#define QUISK_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define QO_SAMP_BUFFER_SIZE 66000           /* quisk.h:15 */

/* FIR with real taps over complex or real data; state = circular history + decimation phase
 * (struct quisk_cFilter / quisk_dFilter, filter.h:1-21) */
typedef struct {
    const double *taps;     /* not owned */
    double *ctaps;          /* complex taps made by qo_fir_tune (owned), interleaved */
    int ntaps;
    int phase;              /* decim_index */
    int pos;                /* index of the slot the next sample is written to */
    double *hist;           /* ntaps complex (or real) samples, interleaved */
    int is_complex;
} qo_fir;

void qo_fir_init(qo_fir *f, const double *taps, int ntaps, int is_complex);   /* filter.c:9-34 */
void qo_fir_free(qo_fir *f);
void qo_fir_tune(qo_fir *f, double freq, int ssb_upper);                       /* filter.c:58-81 */

int qo_cDecimate(double *x, int count, qo_fir *f, int decim);                  /* filter.c:203-229 */
int qo_cCDecimate(double *x, int count, qo_fir *f, int decim);                 /* filter.c:231-257 */
int qo_dDecimate(double *x, int count, qo_fir *f, int decim);                  /* filter.c:259-285 */
int qo_cInterpolate(double *x, int count, qo_fir *f, int interp);              /* filter.c:131-165 */
int qo_dInterpolate(double *x, int count, qo_fir *f, int interp);              /* filter.c:167-201 */
int qo_cInterpDecim(double *x, int count, qo_fir *f, int interp, int decim);   /* filter.c:287-324 */
int qo_dFilter(double *x, int count, qo_fir *f);                               /* filter.c:347-370 */
double qo_dD_out(double sample, qo_fir *f);                                    /* filter.c:326-345 */
void qo_dC_out(double sample, qo_fir *f, double *out_re_im);                   /* filter.c:83-104 */

/* 45-tap half-band (struct quisk_cHB45Filter, filter.h:23-37) */
typedef struct {
    int toggle;
    double samples[2 * 22];
    double center[2 * 11];
} qo_hb45;

void qo_hb45_init(qo_hb45 *f);
int qo_cDecim2HB45(double *x, int count, qo_hb45 *f);                          /* filter.c:377-417 */
int qo_cInterp2HB45(double *x, int count, qo_hb45 *f);                         /* filter.c:455-488 */
int qo_dInterp2HB45(double *x, int count, qo_hb45 *f);                         /* filter.c:420-453 */

extern const double qo_hb45_coef[12];                                          /* filter.c:382-385 */

#ifdef __cplusplus
}
#endif

/* ---------------------------------------------------------------- panadapter (quisk.c, PARITY UNPINNED) */
/* Restates the FFT ring producer in quisk_process_samples (quisk.c:2454-2475, every full block is
 * processed; the reference drops blocks when its 4-deep ring is full), record_app's Hanning window
 * (quisk.c:6003-6009) and get_graph job 1 (quisk.c:5142-5331; scan_blocks = 0, no remote head). */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct qo_graph qo_graph;
qo_graph *qo_graph_create(int fft_size, int data_width, double fft_sample_rate);
void qo_graph_free(qo_graph *g);
/* passband for the RMS S-meter: first bin from (rx_tune_freq + filter_start_offset), width filter_bandwidth */
void qo_graph_set_smeter_band(qo_graph *g, double f_start, double bandwidth);
/* append n complex samples; windows/transforms/averages every completed block.  Returns blocks completed. */
int qo_graph_feed(qo_graph *g, const double *x, int n);
/* the `1/graph_refresh has elapsed` branch: pixel row (data_width doubles, dB, clamped to [-200, 0]) and the
 * S-meter in dB; resets the average.  Returns the number of FFTs that were averaged (0: nothing to return). */
int qo_graph_get(qo_graph *g, double zoom, double deltaf, double *pixels, double *smeter_db);
/* get_bandscope (quisk.c:4957-5011) with init_bandscope's window (quisk.c:2876-2893) and copy2pixels (quisk.c:4932-4955):
 * blocks of `size` real samples (already scaled by 1 / bandscopeScale) -> Hanning, r2c, |X| averaged over the blocks;
 * qo_bscope_get is the `1 / graph_refresh has elapsed` branch: graph_width pixels in dB (floor -200) for the view
 * (zoom, deltaf) of 0 .. clock / 2, and the largest |sample| since the last get (hermes_adc_level).  Returns the number
 * of blocks averaged (0: nothing to return). */
typedef struct qo_bscope qo_bscope;
qo_bscope *qo_bscope_create(int size, int graph_width);
void qo_bscope_free(qo_bscope *b);
void qo_bscope_block(qo_bscope *b, const double *samples);
int qo_bscope_get(qo_bscope *b, int clock, double zoom, double deltaf, double *pixels, double *adc_level);
void qo_copy2pixels(double *pixels, int n_pixels, const double *fft, int fft_size, double zoom, double deltaf, double rate);
/* watfall_OnGraphData (quisk.c:5372-5421): one waterfall row, `width` RGB pixels, from `size` dB values. */
void qo_watfall_row(const double *db, int size, int width, const unsigned char *red, const unsigned char *green,
                    const unsigned char *blue, int y_zero, int y_scale, double gain, unsigned char *rgb);
#ifdef __cplusplus
}
#endif

#endif
