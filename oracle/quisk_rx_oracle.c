/* quisk_rx_oracle.c -- TEST INFRASTRUCTURE ONLY.  See quisk_rx_oracle.h. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fft_oracle.h"
#include "quisk_oracle.h"
#include "quisk_rx_oracle.h"

#define MAX_FILTER_SIZE 10001       /* quisk.h */
#define FM_FILTER_DEMPH 300.0       /* quisk.c:40 */

#define AGC_DELAY 15                /* quisk.c:47 */
#define CLIP32 2147483647.0         /* quisk.h:13 */

struct qo_agc {                     /* struct AgcState, quisk.c:68-84 */
    double max_out;
    int sample_rate, buf_size, index_read, index_start, is_clipping;
    double themax, gain, delta, target_gain, time_release, release_time;
    double *c_samp;
};

qo_agc *qo_agc_create(int sample_rate, double max_out, double release_time)
{
    qo_agc *a = (qo_agc *)calloc(1, sizeof(*a));
    a->max_out = max_out; a->sample_rate = sample_rate; a->release_time = release_time;
    return a;
}

void qo_agc_free(qo_agc *a) { if (a) { free(a->c_samp); free(a); } }

void qo_agc_process(qo_agc *d, double *cs, int count, int is_cpx, double agcReleaseGain)      /* quisk.c:2162-2287 */
{
    int i;
    double out_magn, buf_magn, dtmp, clip_gain, sre, sim;
    if (!d->buf_size) {             /* the first call only initialises (quisk.c:2173-2190) */
        d->buf_size = d->sample_rate * AGC_DELAY / 1000;
        d->index_read = 0; d->index_start = 0; d->is_clipping = 0;
        d->themax = 1.0; d->gain = 100; d->delta = 0; d->target_gain = 100;
        d->time_release = 1.0 - exp(-1.0 / d->sample_rate / d->release_time);
        d->c_samp = (double *)calloc((size_t)d->buf_size * 2, sizeof(double));
        return;
    }
    for (i = 0; i < count; i++) {
        sre = cs[2 * i]; sim = cs[2 * i + 1];
        cs[2 * i] = d->c_samp[2 * d->index_read] * d->gain;                 /* FIFO output */
        cs[2 * i + 1] = d->c_samp[2 * d->index_read + 1] * d->gain;
        out_magn = is_cpx ? hypot(cs[2 * i], cs[2 * i + 1]) : fabs(cs[2 * i]);
        if (out_magn > CLIP32) { cs[2 * i] /= out_magn; cs[2 * i + 1] /= out_magn; }
        d->c_samp[2 * d->index_read] = sre; d->c_samp[2 * d->index_read + 1] = sim;
        buf_magn = is_cpx ? hypot(sre, sim) : fabs(sre);
        if (d->is_clipping == 0) {
            if (buf_magn * d->gain > d->max_out * CLIP32) {
                d->target_gain = d->max_out * CLIP32 / buf_magn;
                d->delta = (d->gain - d->target_gain) / d->buf_size;
                d->is_clipping = 1;
                d->themax = buf_magn;
                d->gain -= d->delta;
            } else if (d->index_read == d->index_start) {
                clip_gain = d->max_out * CLIP32 / d->themax;
                d->target_gain = agcReleaseGain > clip_gain ? clip_gain : agcReleaseGain;
                d->themax = buf_magn;
                d->gain = d->gain * (1.0 - d->time_release) + d->target_gain * d->time_release;
            } else {
                if (d->themax < buf_magn) d->themax = buf_magn;
                d->gain = d->gain * (1.0 - d->time_release) + d->target_gain * d->time_release;
            }
        } else {
            if (buf_magn > d->themax) {
                d->themax = buf_magn;
                d->target_gain = d->max_out * CLIP32 / buf_magn;
                dtmp = (d->gain - d->target_gain) / d->buf_size;
                if (dtmp > d->delta) d->delta = dtmp;
            }
            d->gain -= d->delta;
            if (d->gain <= d->target_gain) {
                d->is_clipping = 0;
                d->gain = d->target_gain;
                d->themax = buf_magn;
                d->index_start = d->index_read;
            }
        }
        if (++d->index_read >= d->buf_size) d->index_read = 0;
    }
}

struct qo_rx {
    int sample_rate, decim2, decim3, decim5, decim_srate, filter_srate, mode, tune, bandwidth;
    qo_rx_tables t;
    double tv_re, tv_im;            /* rxTuneVector, quisk.c:2308 */
    /* quisk_process_decimate storage, quisk.c:1678-1698 */
    qo_hb45 hb[5];
    qo_fir d3[3], d5[3], d48to24, f300d5, d5s, sdriq53, sdriq111, sdriq133, sdriq167, sdriq185;
    /* quisk_process_demodulate storage, quisk.c:1855-1875 */
    qo_hb45 dHB4, dHB5, dHB6, dHB7;
    qo_fir dm48to24, audio24p4, audio12p2, audio24p6, audio48p3, fmhp;
    double fm1_re, fm1_im, dc_remove, FM_a_0, FM_a_1, FM_b_1, FM_x_1, FM_y_1;
    /* cRxFilterOut / dRxFilterOut storage, quisk.c:1190-1193,1225-1229 */
    int sizeFilter, indexC, indexD;
    double *filtI, *filtQ, *bufI, *bufQ, *bufC;
    double *dsamples;
    int dcap;
    double rf_sum, squelch, squelch_level;      /* MeasureSquelch[0], quisk.c:255-263; squelch_level quisk.c:193 */
    int rf_count, squelch_active;
    int ssb_squelch_enabled, ssb_squelch_level; /* set_ssb_squelch, quisk.c:4729 */
    int sq_inited, sq_index, sq_open;           /* ssb_squelch's static plan flag and MS->index, MS->sq_open */
    double sq_in[512], sq_delay[512];           /* MS->in_fft; d_delay's buffer (quisk.c:1057-1084) */
    int sq_delay_index;
    qo_nb *nb;                      /* NoiseBlanker's statics, quisk.c:682-687 */
    qo_notch *notch;                /* dAutoNotch's statics, quisk.c:798-813 */
    int rit_freq;                   /* quisk.c:202 */
    qo_agc *agc;                    /* Agc1 = {0.7, 0, 0}, quisk.c:2321 */
    int agc_on;
    double agc_gain;
};

/* ---- NoiseBlanker, quisk.c:680-784 --------------------------------------------------
 * A delay line of 3 * hwindow samples (hwindow = 500 us) with a running sum of the magnitudes in it.  A sample
 * larger than `limit` times the mean is a pulse: the hwindow samples before it are tapered to zero, samples are
 * zeroed until the pulses stop, then the gain ramps back up over hwindow samples. */
struct qo_nb {
    int level, sample_rate, save_size, hwindow_size, state, index, win_index;
    double save_sum, *cSaved, *dSaved;
};

qo_nb *qo_nb_create(int sample_rate)
{
    qo_nb *b = (qo_nb *)calloc(1, sizeof(*b));
    b->sample_rate = sample_rate;
    b->hwindow_size = (int)(sample_rate * 500.E-6 + 0.5);       /* QUISK_NB_HWINDOW_SECS, quisk.c:679,702 */
    b->save_size = b->hwindow_size * 3;
    b->dSaved = (double *)calloc((size_t)b->save_size, sizeof(double));
    b->cSaved = (double *)calloc((size_t)b->save_size * 2, sizeof(double));
    return b;
}

void qo_nb_free(qo_nb *b) { if (b) { free(b->cSaved); free(b->dSaved); free(b); } }
void qo_nb_set_level(qo_nb *b, int level) { b->level = level; }     /* set_noise_blanker, quisk.c:4605 */
int qo_nb_delay(const qo_nb *b) { return b->save_size; }

void qo_nb_process(qo_nb *b, double *cs, int n)
{
    int i, j, k, is_pulse;
    double mag, limit, sre, sim, f;
    if (b->level <= 0) return;                                   /* quisk.c:695 */
    limit = b->level == 2 ? 4.0 : b->level == 3 ? 2.5 : 6.0;     /* quisk.c:716-728 */
    for (i = 0; i < n; i++) {
        sre = cs[2 * i]; sim = cs[2 * i + 1];                    /* newest sample in, oldest out */
        cs[2 * i] = b->cSaved[2 * b->index]; cs[2 * i + 1] = b->cSaved[2 * b->index + 1];
        b->cSaved[2 * b->index] = sre; b->cSaved[2 * b->index + 1] = sim;
        mag = hypot(sre, sim);                                   /* cabs */
        b->save_sum -= b->dSaved[b->index];
        b->dSaved[b->index] = mag;
        b->save_sum += mag;
        is_pulse = mag <= b->save_sum / b->save_size * limit ? 0 : 1;
        if (b->state == 0) {
            if (is_pulse) {                                      /* taper the samples before the pulse */
                b->state = 1;
                k = b->index;
                for (j = 0; j < b->hwindow_size; j++) {
                    f = (double)j / b->hwindow_size;
                    b->cSaved[2 * k] *= f; b->cSaved[2 * k + 1] *= f;
                    if (--k < 0) k = b->save_size - 1;
                }
            } else if (b->win_index) {                           /* pulses have stopped: ramp up to 1.0 */
                f = (double)b->win_index / b->hwindow_size;
                b->cSaved[2 * b->index] *= f; b->cSaved[2 * b->index + 1] *= f;
                if (++b->win_index >= b->hwindow_size) b->win_index = 0;
            }
        } else {                                                 /* in a pulse: zero until it stops */
            b->cSaved[2 * b->index] = 0; b->cSaved[2 * b->index + 1] = 0;
            if (!is_pulse) { b->state = 0; b->win_index = 1; }
        }
        if (++b->index >= b->save_size) b->index = 0;
    }
}

/* ---- dAutoNotch, quisk.c:786-963 -----------------------------------------------------
 * Overlap-save on 2048-sample blocks of the real audio (510 old + 1538 new samples): r2c FFT, half/half averaged
 * magnitude spectrum, the two strongest bins (not near the CW sidetone, not near each other) with a hysteresis
 * count each; when the pair changes, a 511-tap notch filter is designed by frequency sampling (c2r of a 0/1
 * spectrum, centred, Hanning window, r2c), the block's spectrum is multiplied by it and transformed back.  The
 * macros are the reference's (note NOTCH_DATA_SIZE / 20 = 102 and the un-parenthesised DESIGN_SIZE). */
#define NOTCH_DATA_SIZE 2048
#define NOTCH_FILTER_DESIGN_SIZE 512            /* NOTCH_DATA_SIZE / 4 */
#define NOTCH_FILTER_SIZE 511                   /* (NOTCH_FILTER_DESIGN_SIZE - 1) */
#define NOTCH_FILTER_FFT_SIZE 256               /* (NOTCH_FILTER_SIZE / 2 + 1) */
#define NOTCH_DATA_START_SIZE 510               /* (NOTCH_FILTER_SIZE - 1) */
#define NOTCH_DATA_OUTPUT_SIZE 1538             /* (NOTCH_DATA_SIZE - NOTCH_DATA_START_SIZE) */
#define NOTCH_FFT_SIZE 1025                     /* (NOTCH_DATA_SIZE / 2 + 1) */
struct qo_notch {
    int on, old1, count1, old2, count2, index, fltrSig;
    double data_in[NOTCH_DATA_SIZE], data_out[NOTCH_DATA_SIZE];
    double notch_fft[2 * NOTCH_FFT_SIZE], fltr_fft[2 * NOTCH_FFT_SIZE];     /* complex double */
    double fft_window[NOTCH_DATA_SIZE], fltr_in[NOTCH_DATA_SIZE], fltr_out[NOTCH_FILTER_DESIGN_SIZE];
    double average_fft[NOTCH_FFT_SIZE];
};

/* fftw_plan_dft_r2c_1d: bins 0 .. n/2 of the forward transform of n real samples */
static void notch_r2c(const double *in, double *out, int n)
{
    double *buf = (double *)malloc((size_t)n * 2 * sizeof(double));
    int i;
    for (i = 0; i < n; i++) { buf[2 * i] = in[i]; buf[2 * i + 1] = 0.0; }
    fo_fft(buf, n, -1);
    memcpy(out, buf, (size_t)(n / 2 + 1) * 2 * sizeof(double));
    free(buf);
}

/* fftw_plan_dft_c2r_1d: unnormalised inverse of the Hermitian extension of bins 0 .. n/2 (imaginary parts of bins 0
 * and n/2 do not enter) */
static void notch_c2r(const double *in, double *out, int n)
{
    double *buf = (double *)malloc((size_t)n * 2 * sizeof(double));
    int i;
    buf[0] = in[0]; buf[1] = 0.0;
    buf[n] = in[n]; buf[n + 1] = 0.0;
    for (i = 1; i < n / 2; i++) {
        buf[2 * i] = in[2 * i]; buf[2 * i + 1] = in[2 * i + 1];
        buf[2 * (n - i)] = in[2 * i]; buf[2 * (n - i) + 1] = -in[2 * i + 1];
    }
    fo_fft(buf, n, +1);
    for (i = 0; i < n; i++) out[i] = buf[2 * i];
    free(buf);
}

void qo_notch_init(qo_notch *a)                                  /* dAutoNotch(NULL, 0, 0, 0), quisk.c:826-834 */
{
    a->index = NOTCH_DATA_START_SIZE;
    a->fltrSig = -1;
    a->old1 = a->old2 = 0;
    a->count1 = a->count2 = -4;
    memset(a->data_out, 0, sizeof(a->data_out));
    memset(a->data_in, 0, sizeof(a->data_in));
    memset(a->average_fft, 0, sizeof(a->average_fft));
}

qo_notch *qo_notch_create(void)
{
    int i;
    qo_notch *a = (qo_notch *)calloc(1, sizeof(*a));
    for (i = 0; i < NOTCH_FILTER_SIZE; i++)                      /* Hanning, quisk.c:822-823; the rest stays 0 */
        a->fft_window[i] = 0.50 - 0.50 * cos(2. * M_PI * i / (NOTCH_FILTER_SIZE));
    qo_notch_init(a);
    return a;
}

void qo_notch_free(qo_notch *a) { free(a); }
void qo_notch_set(qo_notch *a, int on) { a->on = on; qo_notch_init(a); }        /* set_auto_notch, quisk.c:4596-4603 */

void qo_notch_process(qo_notch *a, double *dsamples, int nSamples, int sidetone, int rate)
{
    int i, j, k, i1, i2, inp, signal, delta_sig, delta_i1, half_width;
    double d, d1, d2, avg;
    if (!a->on) return;                                          /* quisk.c:835-836 */
    for (inp = 0; inp < nSamples; inp++) {
        a->data_in[a->index] = dsamples[inp];
        dsamples[inp] = a->data_out[a->index];
        if (++a->index >= NOTCH_DATA_SIZE) {
            a->index = NOTCH_DATA_START_SIZE;
            notch_r2c(a->data_in, a->notch_fft, NOTCH_DATA_SIZE);
            delta_sig = (300 * 2 * NOTCH_FFT_SIZE + rate / 2) / rate;
            delta_i1 = (400 * 2 * NOTCH_FFT_SIZE + rate / 2) / rate;
            signal = sidetone != 0 ? (abs(sidetone) * 2 * NOTCH_FFT_SIZE + rate / 2) / rate : -999;
            avg = 1;
            d1 = 0; i1 = 0;
            for (i = 0; i < NOTCH_FFT_SIZE; i++) {
                d = hypot(a->notch_fft[2 * i], a->notch_fft[2 * i + 1]);
                avg += d;
                a->average_fft[i] = 0.5 * a->average_fft[i] + 0.5 * d;
                if (abs(i - signal) > delta_sig && a->average_fft[i] > d1) { d1 = a->average_fft[i]; i1 = i; }
            }
            if (abs(i1 - a->old1) < 3) a->count1++; else a->count1--;
            if (a->count1 > 4) a->count1 = 4; else if (a->count1 < -1) a->count1 = -1;
            if (a->count1 < 0) a->old1 = i1;
            avg /= NOTCH_FFT_SIZE;
            d2 = 0; i2 = 0;
            for (i = 0; i < NOTCH_FFT_SIZE; i++)
                if (abs(i - signal) > delta_sig && abs(i - i1) > delta_i1 && a->average_fft[i] > d2) { d2 = a->average_fft[i]; i2 = i; }
            if (abs(i2 - a->old2) < 3) a->count2++; else a->count2--;
            if (a->count2 > 4) a->count2 = 4; else if (a->count2 < -2) a->count2 = -2;
            if (a->count2 < 0) a->old2 = i2;
            if (a->count1 > 0 && a->count2 > 0) k = i1 + 10000 * i2;
            else if (a->count1 > 0) k = i1;
            else k = 0;
            if (a->fltrSig != k) {                               /* make the filter if it is different */
                a->fltrSig = k;
                half_width = (100 * 2 * NOTCH_FILTER_FFT_SIZE + rate / 2) / rate;
                if (half_width < 3) half_width = 3;
                for (i = 0; i < NOTCH_FILTER_FFT_SIZE; i++) { a->fltr_fft[2 * i] = 1.0; a->fltr_fft[2 * i + 1] = 0.0; }
                k = (i1 + 2) / 4;
                if (a->count1 > 0)
                    for (i = -half_width; i <= half_width; i++) {
                        j = k + i;
                        if (j >= 0 && j < NOTCH_FILTER_FFT_SIZE) { a->fltr_fft[2 * j] = 0.0; a->fltr_fft[2 * j + 1] = 0.0; }
                    }
                k = (i2 + 2) / 4;
                if (a->count1 > 0 && a->count2 > 0)
                    for (i = -half_width; i <= half_width; i++) {
                        j = k + i;
                        if (j >= 0 && j < NOTCH_FILTER_FFT_SIZE) { a->fltr_fft[2 * j] = 0.0; a->fltr_fft[2 * j + 1] = 0.0; }
                    }
                /* fltrRev reads bins 0 .. 256; bin 256 still holds what the previous fltrFwd left there */
                notch_c2r(a->fltr_fft, a->fltr_out, NOTCH_FILTER_DESIGN_SIZE);
                memmove(a->fltr_out + NOTCH_FILTER_DESIGN_SIZE / 2 - 1, a->fltr_out, sizeof(double) * (NOTCH_FILTER_SIZE / 2 - 1));
                for (i = NOTCH_FILTER_DESIGN_SIZE / 2 - 2, j = NOTCH_FILTER_DESIGN_SIZE / 2; i >= 0; i--, j++)
                    a->fltr_out[i] = a->fltr_out[j];
                for (i = 0; i < NOTCH_FILTER_SIZE; i++)
                    a->fltr_in[i] = a->fltr_out[i] * a->fft_window[i] / NOTCH_DATA_SIZE / 4;   /* "/ NOTCH_FILTER_DESIGN_SIZE" as the macro expands */
                for (i = NOTCH_FILTER_SIZE; i < NOTCH_DATA_SIZE; i++) a->fltr_in[i] = 0.0;
                notch_r2c(a->fltr_in, a->fltr_fft, NOTCH_DATA_SIZE);
            }
            for (i = 0; i < NOTCH_FFT_SIZE; i++) {              /* apply the filter */
                double xr = a->notch_fft[2 * i], xi = a->notch_fft[2 * i + 1], fr = a->fltr_fft[2 * i], fi = a->fltr_fft[2 * i + 1];
                a->notch_fft[2 * i] = xr * fr - xi * fi;
                a->notch_fft[2 * i + 1] = xr * fi + xi * fr;
            }
            notch_c2r(a->notch_fft, a->data_out, NOTCH_DATA_SIZE);
            memmove(a->data_in, a->data_in + NOTCH_DATA_OUTPUT_SIZE, NOTCH_DATA_START_SIZE * sizeof(double));
            for (i = NOTCH_DATA_START_SIZE; i < NOTCH_DATA_SIZE; i++)
                a->data_out[i] /= NOTCH_DATA_SIZE / 20;          /* "Empirical": integer 102 */
        }
    }
    (void)avg;
}

int qo_rx_decim_srate(const qo_rx *r) { return r->decim_srate; }
int qo_rx_filter_srate(const qo_rx *r) { return r->filter_srate; }

static int plan_decimation(int rate, int *p2, int *p3, int *p5)     /* quisk.c:1633-1671 */
{
    int i, best = rate, try_, i2, i3, i5, d2 = 0, d3 = 0, d5 = 0;
    for (i2 = 0; i2 <= 6; i2++)
        for (i3 = 0; i3 <= 3; i3++)
            for (i5 = 0; i5 <= 3; i5++) {
                try_ = rate;
                for (i = 0; i < i2; i++) try_ /= 2;
                for (i = 0; i < i3; i++) try_ /= 3;
                for (i = 0; i < i5; i++) try_ /= 5;
                if (try_ >= 48000 && try_ < best) { d2 = i2; d3 = i3; d5 = i5; best = try_; }
            }
    if (best >= 50000) best = best * 24 / 25;
    *p2 = d2; *p3 = d3; *p5 = d5;
    return best;
}

qo_rx *qo_rx_create(int sample_rate, const qo_rx_tables *t)
{
    int i, d2, d3, d5;
    double www, nnn;
    qo_rx *r;
    plan_decimation(sample_rate, &d2, &d3, &d5);
    r = (qo_rx *)calloc(1, sizeof(*r));
    r->sample_rate = sample_rate; r->decim2 = d2; r->decim3 = d3; r->decim5 = d5;
    r->t = *t;
    r->tv_re = 1.0; r->tv_im = 0.0;
    r->mode = QO_USB;
    for (i = 0; i < 5; i++) qo_hb45_init(&r->hb[i]);
    for (i = 0; i < 3; i++) { qo_fir_init(&r->d3[i], t->f144d3, 147, 1); qo_fir_init(&r->d5[i], t->f240d5, 245, 1); }
    qo_fir_init(&r->d48to24, t->f48dec24, 98, 1);
    qo_fir_init(&r->f300d5, t->f300d5, 125, 1);                 /* quisk.c:1720 */
    qo_fir_init(&r->d5s, t->f240d5, 245, 1);                    /* filtDecim5S, quisk.c:1717 */
    qo_fir_init(&r->sdriq53, t->sdriq53, 55, 1); qo_fir_init(&r->sdriq111, t->sdriq111, 114, 1);
    qo_fir_init(&r->sdriq133, t->sdriq133, 136, 1); qo_fir_init(&r->sdriq167, t->sdriq167, 174, 1);
    qo_fir_init(&r->sdriq185, t->sdriq185, 189, 1);
    r->bandwidth = 2700;
    r->squelch_level = -999.0;
    qo_hb45_init(&r->dHB4); qo_hb45_init(&r->dHB5); qo_hb45_init(&r->dHB6); qo_hb45_init(&r->dHB7);
    qo_fir_init(&r->dm48to24, t->f48dec24, 98, 1);
    qo_fir_init(&r->audio24p4, t->audio24p4, 50, 0);
    qo_fir_init(&r->audio12p2, t->audio24p4, 50, 0);            /* quisk.c:1884: same table */
    qo_fir_init(&r->audio24p6, t->audio24p6, 36, 0);
    qo_fir_init(&r->audio48p3, t->lp48, 186, 0);
    qo_fir_init(&r->fmhp, t->fmhp, 309, 0);
    r->fm1_re = 10; r->fm1_im = 0;                              /* quisk.c:1893 */
    www = tan(M_PI * FM_FILTER_DEMPH / 48000);                  /* quisk.c:1894-1898 */
    nnn = 1.0 / (1.0 + www);
    r->FM_a_0 = www * nnn; r->FM_a_1 = r->FM_a_0; r->FM_b_1 = nnn * (www - 1.0);
    r->filtI = (double *)calloc(MAX_FILTER_SIZE, sizeof(double));
    r->filtQ = (double *)calloc(MAX_FILTER_SIZE, sizeof(double));
    r->bufI = (double *)calloc(MAX_FILTER_SIZE, sizeof(double));
    r->bufQ = (double *)calloc(MAX_FILTER_SIZE, sizeof(double));
    r->bufC = (double *)calloc(2 * MAX_FILTER_SIZE, sizeof(double));
    return r;
}

void qo_rx_free(qo_rx *r)
{
    int i;
    if (!r) return;
    for (i = 0; i < 3; i++) { qo_fir_free(&r->d3[i]); qo_fir_free(&r->d5[i]); }
    qo_fir_free(&r->f300d5); qo_fir_free(&r->d5s); qo_fir_free(&r->sdriq53); qo_fir_free(&r->sdriq111);
    qo_fir_free(&r->sdriq133); qo_fir_free(&r->sdriq167); qo_fir_free(&r->sdriq185);
    qo_fir_free(&r->d48to24); qo_fir_free(&r->dm48to24); qo_fir_free(&r->audio24p4); qo_fir_free(&r->audio12p2);
    qo_fir_free(&r->audio24p6); qo_fir_free(&r->audio48p3); qo_fir_free(&r->fmhp);
    qo_agc_free(r->agc);
    qo_nb_free(r->nb);
    qo_notch_free(r->notch);
    free(r->filtI); free(r->filtQ); free(r->bufI); free(r->bufQ); free(r->bufC); free(r->dsamples);
    free(r);
}

void qo_rx_set_tune(qo_rx *r, int f) { r->tune = f; }
void qo_rx_set_mode(qo_rx *r, int mode) { r->mode = mode; }
void qo_rx_set_bandwidth(qo_rx *r, int bw) { r->bandwidth = bw; }
void qo_rx_set_squelch(qo_rx *r, double level) { r->squelch_level = level; }     /* set_squelch, quisk.c:4721-4727 */
void qo_rx_set_ssb_squelch(qo_rx *r, int enabled, int level) { r->ssb_squelch_enabled = enabled; r->ssb_squelch_level = level; }
void qo_rx_set_agc(qo_rx *r, int on, double release_gain) { r->agc_on = on; r->agc_gain = release_gain; }

void qo_rx_set_auto_notch(qo_rx *r, int on, int rit_freq)
{
    if (!r->notch) r->notch = qo_notch_create();
    r->rit_freq = rit_freq;
    qo_notch_set(r->notch, on);
}

void qo_rx_set_noise_blanker(qo_rx *r, int level)
{
    if (!r->nb) r->nb = qo_nb_create(r->sample_rate);
    qo_nb_set_level(r->nb, level);
}

void qo_rx_set_filters(qo_rx *r, const double *fI, const double *fQ, int size)
{
    memcpy(r->filtI, fI, (size_t)size * sizeof(double));
    memcpy(r->filtQ, fQ, (size_t)size * sizeof(double));
    r->sizeFilter = size;
}

static void cRxFilterOut(qo_rx *r, double re, double im, double *ore, double *oim)     /* quisk.c:1218-1256 */
{
    int j, k;
    double accI = 0, accQ = 0;
    if (!r->sizeFilter) { *ore = re; *oim = im; return; }
    if (r->indexC >= r->sizeFilter) r->indexC = 0;
    r->bufI[r->indexC] = re;
    r->bufQ[r->indexC] = im;
    j = r->indexC;
    for (k = 0; k < r->sizeFilter; k++) {
        accI += r->bufI[j] * r->filtI[k];
        accQ += r->bufQ[j] * r->filtQ[k];
        if (++j >= r->sizeFilter) j = 0;
    }
    r->indexC++;
    *ore = accI; *oim = accQ;
}

static void dRxFilterOut(qo_rx *r, double re, double im, double *ore, double *oim)     /* quisk.c:1182-1216 */
{
    int j, k;
    double ar = 0, ai = 0;
    if (!r->sizeFilter) { *ore = re; *oim = im; return; }
    if (r->indexD >= r->sizeFilter) r->indexD = 0;
    r->bufC[2 * r->indexD] = re;
    r->bufC[2 * r->indexD + 1] = im;
    j = r->indexD;
    for (k = 0; k < r->sizeFilter; k++) {
        ar += r->bufC[2 * j] * r->filtI[k];
        ai += r->bufC[2 * j + 1] * r->filtI[k];
        if (++j >= r->sizeFilter) j = 0;
    }
    r->indexD++;
    *ore = ar; *oim = ai;
}

#define SQUELCH_FFT_SIZE 512        /* quisk.c:53 */
#define CLIP16 32767.0              /* quisk.h:14 */

static void ssb_squelch(qo_rx *r, const double *ds, int n, int samp_rate)      /* quisk.c:1086-1180 */
{
    int i, bw, bw1, bw2, inp;
    double d, arith_avg, geom_avg, ratio;
    static double fft_window[SQUELCH_FFT_SIZE];
    double buf[2 * SQUELCH_FFT_SIZE];
    if (!r->sq_inited) {            /* "if (!plan) { ...; return; }": the first call only sets up */
        r->sq_inited = 1;
        return;
    }
    for (i = 0; i < SQUELCH_FFT_SIZE; i++) fft_window[i] = 0.50 - 0.50 * cos(2. * M_PI * i / SQUELCH_FFT_SIZE);
    for (inp = 0; inp < n; inp++) {
        r->sq_in[r->sq_index++] = ds[inp];
        if (r->sq_index >= SQUELCH_FFT_SIZE) {
            r->sq_index = 0;
            for (i = 0; i < SQUELCH_FFT_SIZE; i++) { buf[2 * i] = r->sq_in[i] * fft_window[i]; buf[2 * i + 1] = 0.0; }
            fo_fft(buf, SQUELCH_FFT_SIZE, -1);                  /* fftw_execute_dft_r2c: bins 0 .. N/2 */
            bw = r->bandwidth;
            if (bw > 3000) bw = 3000;
            bw1 = 300 * SQUELCH_FFT_SIZE / samp_rate;
            bw2 = (bw + 300) * SQUELCH_FFT_SIZE / samp_rate;
            if (bw2 > SQUELCH_FFT_SIZE / 2 + 1) bw2 = SQUELCH_FFT_SIZE / 2 + 1;        /* out_fft holds N/2 + 1 bins */
            arith_avg = 0.0; geom_avg = 0.0;
            for (i = bw1; i < bw2; i++) {
                double cr = buf[2 * i] / CLIP16, ci = buf[2 * i + 1] / CLIP16;
                d = cr * cr + ci * ci;
                if (d > 1E-4) { arith_avg += d; geom_avg += log(d); }
            }
            if (arith_avg > 1E-4) {
                bw = bw2 - bw1;
                arith_avg = log(arith_avg / bw);
                geom_avg /= bw;
                ratio = arith_avg - geom_avg;
            } else {
                ratio = 1.0;
            }
            if (ratio > r->ssb_squelch_level * 0.005) r->sq_open = samp_rate;       /* one second timer */
        }
    }
    r->sq_open -= n;
    if (r->sq_open < 0) r->sq_open = 0;
    r->squelch_active = r->sq_open == 0;
}

static void d_delay(qo_rx *r, double *ds, int n)                /* quisk.c:1057-1084, samp_delay = SQUELCH_FFT_SIZE */
{
    int i;
    for (i = 0; i < n; i++) {
        double sample = r->sq_delay[r->sq_delay_index];
        r->sq_delay[r->sq_delay_index] = ds[i];
        ds[i] = sample;
        if (++r->sq_delay_index >= SQUELCH_FFT_SIZE) r->sq_delay_index = 0;
    }
}

static int process_decimate(qo_rx *r, double *x, int n)        /* quisk.c:1729-1843 */
{
    int i2 = r->decim2, i3 = r->decim3, i5 = r->decim5, k = 0;
    switch ((r->sample_rate + 100) / 1000) {
    case 41:
        r->decim_srate = 48000;
        break;
    case 53:
        r->decim_srate = r->sample_rate;
        n = qo_cDecimate(x, n, &r->sdriq53, 1);
        break;
    case 111:
        r->decim_srate = r->sample_rate / 2;
        n = qo_cDecimate(x, n, &r->sdriq111, 2);
        break;
    case 133:
        r->decim_srate = r->sample_rate / 2;
        n = qo_cDecimate(x, n, &r->sdriq133, 2);
        break;
    case 185:
        r->decim_srate = r->sample_rate / 3;
        n = qo_cDecimate(x, n, &r->sdriq185, 3);
        break;
    case 370:
        r->decim_srate = r->sample_rate / 6;
        n = qo_cDecim2HB45(x, n, &r->hb[1]);
        n = qo_cDecimate(x, n, &r->sdriq185, 3);
        break;
    case 740:
        r->decim_srate = r->sample_rate / 12;
        n = qo_cDecim2HB45(x, n, &r->hb[1]);
        n = qo_cDecim2HB45(x, n, &r->hb[2]);
        n = qo_cDecimate(x, n, &r->sdriq185, 3);
        break;
    case 1333:
        r->decim_srate = r->sample_rate / 24;
        n = qo_cDecim2HB45(x, n, &r->hb[0]);
        n = qo_cDecim2HB45(x, n, &r->hb[1]);
        n = qo_cDecim2HB45(x, n, &r->hb[2]);
        n = qo_cDecimate(x, n, &r->sdriq167, 3);
        break;
    default:
        r->decim_srate = r->sample_rate;
        while (i2 > 1 && k < 5) { n = qo_cDecim2HB45(x, n, &r->hb[k++]); r->decim_srate /= 2; i2--; }
        k = 0;
        while (i3 > 0) { n = qo_cDecimate(x, n, &r->d3[k++], 3); r->decim_srate /= 3; i3--; }
        k = 0;
        while (i5 > 0) { n = qo_cDecimate(x, n, &r->d5[k++], 5); r->decim_srate /= 5; i5--; }
        if (i2 > 0) { n = qo_cDecimate(x, n, &r->d48to24, 2); r->decim_srate /= 2; i2--; }
        if (r->decim_srate >= 50000) {                          /* quisk.c:1834-1838 */
            r->decim_srate = r->decim_srate * 24 / 25;
            n = qo_cInterpDecim(x, n, &r->f300d5, 6, 5);
            n = qo_cInterpDecim(x, n, &r->d5s, 4, 5);
        }
        break;
    }
    return n;
}

static int process_demodulate(qo_rx *r, double *x, double *ds, int n)   /* quisk.c:1906-2068 */
{
    int i;
    double re, im, d, di;
    r->squelch_active = 0;                                      /* quisk.c:1908 */
    switch (r->mode) {
    case QO_CWL: case QO_CWU:
        r->filter_srate = r->decim_srate / 8;
        n = qo_cDecim2HB45(x, n, &r->dHB5);
        n = qo_cDecim2HB45(x, n, &r->dHB4);
        n = qo_cDecimate(x, n, &r->dm48to24, 2);
        for (i = 0; i < n; i++) {
            cRxFilterOut(r, x[2 * i], x[2 * i + 1], &re, &im);
            ds[i] = r->mode == QO_CWL ? re + im : re - im;
        }
        if (r->notch) qo_notch_process(r->notch, ds, n, r->rit_freq, r->filter_srate);                 /* quisk.c:1923-1924 */
        if (r->ssb_squelch_enabled) { ssb_squelch(r, ds, n, r->filter_srate); d_delay(r, ds, n); }     /* quisk.c:1925-1928 */
        n = qo_dInterpolate(ds, n, &r->audio12p2, 2);
        n = qo_dInterp2HB45(ds, n, &r->dHB6);
        n = qo_dInterp2HB45(ds, n, &r->dHB7);
        break;
    case QO_LSB: case QO_USB: default:
        r->filter_srate = r->decim_srate / 4;
        n = qo_cDecim2HB45(x, n, &r->dHB5);
        n = qo_cDecimate(x, n, &r->dm48to24, 2);
        for (i = 0; i < n; i++) {
            cRxFilterOut(r, x[2 * i], x[2 * i + 1], &re, &im);
            ds[i] = r->mode == QO_LSB ? re + im : re - im;
        }
        if (r->notch) qo_notch_process(r->notch, ds, n, 0, r->filter_srate);                           /* quisk.c:1968-1969 */
        if (r->ssb_squelch_enabled) { ssb_squelch(r, ds, n, r->filter_srate); d_delay(r, ds, n); }     /* quisk.c:1970-1973 */
        n = qo_dInterpolate(ds, n, &r->audio24p4, 2);
        n = qo_dInterp2HB45(ds, n, &r->dHB7);
        break;
    case QO_AM:
        r->filter_srate = r->decim_srate / 2;
        n = qo_cDecimate(x, n, &r->dm48to24, 2);
        for (i = 0; i < n; i++) {
            dRxFilterOut(r, x[2 * i], x[2 * i + 1], &re, &im);
            di = hypot(re, im);                                 /* cabs */
            d = di + r->dc_remove * 0.99;
            di = d - r->dc_remove;
            r->dc_remove = d;
            ds[i] = di;
        }
        n = qo_dFilter(ds, n, &r->audio24p6);
        if (r->notch) qo_notch_process(r->notch, ds, n, 0, r->filter_srate);                           /* quisk.c:2018-2019 */
        if (r->ssb_squelch_enabled) { ssb_squelch(r, ds, n, r->filter_srate); d_delay(r, ds, n); }     /* quisk.c:2020-2023 */
        n = qo_dInterp2HB45(ds, n, &r->dHB7);
        break;
    case QO_FM: case QO_DGT_FM:
        r->filter_srate = r->decim_srate;
        for (i = 0; i < n; i++) {
            double pr, pi;
            dRxFilterOut(r, x[2 * i], x[2 * i + 1], &re, &im);
            r->rf_sum += hypot(re, im);                         /* MeasureSquelch[bank].rf_sum += cabs(cx), quisk.c:2032 */
            r->rf_count += 1;
            pr = re * r->fm1_re + im * r->fm1_im;               /* cx * conj(fm_1) */
            pi = im * r->fm1_re - re * r->fm1_im;
            di = atan2(pi, pr);                                 /* carg */
            r->fm1_re = re; r->fm1_im = im;
            ds[i] = di;
        }
        for (i = 0; i < n; i++) {
            ds[i] *= 20e5;
            di = ds[i];
            ds[i] = r->FM_y_1 = di * r->FM_a_0 + r->FM_x_1 * r->FM_a_1 - r->FM_y_1 * r->FM_b_1;
            r->FM_x_1 = di;
        }
        n = qo_dDecimate(ds, n, &r->audio48p3, 4);
        n = qo_dFilter(ds, n, &r->fmhp);
        n = qo_dInterp2HB45(ds, n, &r->dHB6);
        n = qo_dInterp2HB45(ds, n, &r->dHB7);
        if (r->notch) qo_notch_process(r->notch, ds, n, 0, r->filter_srate);                           /* quisk.c:2069-2070 */
        if (r->rf_count >= 2400) {                              /* quisk.c:2076-2084 */
            r->squelch = r->rf_sum / r->rf_count / CLIP32;
            r->squelch = r->squelch > 1.E-10 ? 20 * log10(r->squelch) : -200.0;
            r->rf_sum = 0; r->rf_count = 0;
        }
        r->squelch_active = r->squelch < r->squelch_level;      /* quisk.c:2085 */
        break;
    case QO_DGT_U: case QO_FDV_U: case QO_DGT_L: case QO_FDV_L:         /* quisk.c:2087-2140 */
        if (r->bandwidth < 3000) {                                       /* DGT_NARROW_FREQ, quisk.c:52 */
            r->filter_srate = r->decim_srate / 8;
            n = qo_cDecim2HB45(x, n, &r->dHB5);
            n = qo_cDecim2HB45(x, n, &r->dHB4);
            n = qo_cDecimate(x, n, &r->dm48to24, 2);
        } else {
            r->filter_srate = r->decim_srate;
        }
        for (i = 0; i < n; i++) {
            cRxFilterOut(r, x[2 * i], x[2 * i + 1], &re, &im);
            ds[i] = (r->mode == QO_DGT_L || r->mode == QO_FDV_L) ? re + im : re - im;
        }
        if (r->notch) qo_notch_process(r->notch, ds, n, 0, r->filter_srate);                           /* quisk.c:2106-2107,2133-2134 */
        if (r->bandwidth < 3000) {
            n = qo_dInterpolate(ds, n, &r->audio12p2, 2);
            n = qo_dInterp2HB45(ds, n, &r->dHB6);
            n = qo_dInterp2HB45(ds, n, &r->dHB7);
        }
        break;
    case QO_DGT_IQ:                                                      /* quisk.c:2141-2153 */
        r->filter_srate = r->decim_srate;
        if (r->bandwidth < 19000)
            for (i = 0; i < n; i++) dRxFilterOut(r, x[2 * i], x[2 * i + 1], &x[2 * i], &x[2 * i + 1]);
        break;
    }
    return n;
}

int qo_rx_process(qo_rx *r, double *x, int n)
{
    int i;
    if (n <= 0) return n;
    if (n * 2 > r->dcap) { r->dcap = n * 2 + 64; free(r->dsamples); r->dsamples = (double *)malloc((size_t)r->dcap * sizeof(double)); }
    if (r->nb) qo_nb_process(r->nb, x, n);                      /* quisk.c:2448-2449 */
    if (r->tune != 0) {                                         /* quisk.c:2477-2488 */
        double a = -2.0 * M_PI * r->tune / r->sample_rate;      /* cexp((I * -2.0 * M_PI * tune) / sample_rate) */
        double pr = cos(a), pi = sin(a), t;
        for (i = 0; i < n; i++) {
            t = x[2 * i] * r->tv_re - x[2 * i + 1] * r->tv_im;
            x[2 * i + 1] = x[2 * i] * r->tv_im + x[2 * i + 1] * r->tv_re;
            x[2 * i] = t;
            t = r->tv_re * pr - r->tv_im * pi;
            r->tv_im = r->tv_re * pi + r->tv_im * pr;
            r->tv_re = t;
        }
    }
    n = process_decimate(r, x, n);
    n = process_demodulate(r, x, r->dsamples, n);
    if (r->mode != QO_DGT_IQ)                                   /* "This mode is already stereo", quisk.c:2534 */
        for (i = 0; i < n; i++) { x[2 * i] = r->dsamples[i]; x[2 * i + 1] = r->dsamples[i]; }     /* quisk.c:2622-2627 */
    if (r->agc_on) {                                            /* quisk.c:2686-2702; playback rate = decim_srate here */
        if (!r->agc) r->agc = qo_agc_create(r->decim_srate, 0.7, 1.0);
        qo_agc_process(r->agc, x, n, r->mode == QO_DGT_IQ, r->agc_gain);
    }
    if (r->squelch_active)                                      /* squelch_real && squelch_imag, quisk.c:2623,2716-2719 */
        for (i = 0; i < n; i++) { x[2 * i] = 0; x[2 * i + 1] = 0; }
    return n;
}
