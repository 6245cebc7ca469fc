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
    int bandwidth0;                             /* filter_bandwidth[0]: what ssb_squelch reads in every bank (quisk.c:1120) */
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
    r->bandwidth = 2700; r->bandwidth0 = 2700;
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
void qo_rx_set_bandwidth(qo_rx *r, int bw) { r->bandwidth = bw; r->bandwidth0 = bw; }      /* a receiver on its own is bank 0 with filter set 0 */
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
            bw = r->bandwidth0;                                 /* "bw = filter_bandwidth[0]", whatever the bank or nFilter: quisk.c:1120 */
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

/* ==== quisk_process_samples as a whole, quisk.c:2289-2742 ============================================================
 * The orchestration around the banks: key-down replacement (sidetone / silence, quisk.c:2368-2433), AddTestTone
 * (quisk.c:1258-1303), spectrum inversion, NoiseBlanker, the FFT ring, the tune, quisk_process_decimate /
 * measure_freq (quisk.c:5579-5649) / quisk_process_demodulate on bank 0, the second channel (split Rx/Tx or the played
 * sub-receiver) on bank 1 with Buffer2Chan (quisk.c:1577-1611), the digital output of sub-receiver 1 on bank 2, and the
 * tail: cFracDecim (quisk.c:622-665) -> wdspFexchange0 -> HB45 interpolation to the playback rate (quisk.c:2654-2682)
 * -> process_agc -> kill_audio / squelch -> key-up envelope.  The reference keeps all of this in function statics and
 * globals; here it is one struct.  Bank storage (decimator and demodulator filters, c/dRxFilterOut rings) is per bank,
 * the filter taps per nFilter and `sizeFilter` one global for all of them, as in the reference (quisk.c:127-129,4591).
 * PARITY UNPINNED (quisk.c needs <fftw3.h>); every filter it calls is the pinned quisk_oracle.c. */
#include "wdsp_oracle.h"

#define QO_MAX_SUB 9                /* QUISK_MAX_SUB_RECEIVERS, quisk.h */
#define BUF2CHAN_SIZE 12000         /* quisk.c:1576 */
#define BIG_VOLUME 2.2e9            /* quisk.h:11 */
#define MF_FFT_SIZE 12000           /* measure_freq's fft_size, quisk.c:5586 */

struct qo_ps {
    int sample_rate, playback_rate;
    qo_rx *bank[3];
    qo_rx_tables t;
    /* globals set by the GUI thread */
    int rx_mode, rx_tune, tx_tune, rit_freq, split_rxtx, play_channel, play_method, multirx_count;
    int sub_freq[QO_MAX_SUB], sub_mode[QO_MAX_SUB];
    double *sub_samples[QO_MAX_SUB];            /* multirx_cSamples[] for the coming call (NULL: none) */
    int sub_cap[QO_MAX_SUB];
    double *filtI[3], *filtQ[3];                /* cFilterI / cFilterQ [nFilter] */
    int filter_bandwidth[3], sizeFilter;
    int key_down, cw_key_down, active_sidetone, is_fdx, kill_audio, invert, nb_level, sub_rx1_driver;
    int txrx_silence_ms;
    double sidetone_volume, agc_release_gain;
    double tt_phase_re, tt_phase_im;            /* testtonePhase (0 = off) */
    int measure_freq_mode;
    double measured_frequency;
    /* statics of quisk_process_samples, quisk.c:2301-2321 */
    int old_split, old_play;
    double rxTV[2], txTV[2], aux1TV[2], aux2TV[2], sidetoneV[2], sidetonePhase[2];
    double dOutCounter, sidetoneEnvelope, keyupEnvelope;
    int sidetoneIsOn, playSilence;
    qo_hb45 HalfBand7, HalfBand8, HalfBand9;
    qo_agc *Agc1, *Agc2, *Agc3;
    qo_nb *nb;
    /* AddTestTone statics */
    double ttV[2], audioV[2];
    /* Buffer2Chan statics */
    int nbuf1, nbuf2;
    int b2c_overrun;                /* a call handed Buffer2Chan more than its arrays hold: the reference writes past them there (quisk.c:1589-1595) */
    double buf1[BUF2CHAN_SIZE], buf2[BUF2CHAN_SIZE];
    /* cFracDecim statics */
    double fd_dindex, fd_c0[2], fd_c1[2], fd_c2[2];
    /* measure_freq statics */
    int mf_index, mf_count;
    double *mf_samples, *mf_window, *mf_average;
    qo_hb45 mfHB1, mfHB2, mfHB3;
    /* hooks */
    qo_graph *graph;
    wo_shim *shim; wo_fexchange0_fn wdsp_fn; void *wdsp_ctx;
    /* work */
    double *dsamples, *dsamples2, *orig, *bufc, *sub1_out;
    int cap, sub1_n;
    int squelch_real, squelch_imag;             /* as left by the last call */
};

qo_ps *qo_ps_create(int sample_rate, int playback_rate, const qo_rx_tables *t)
{
    int i;
    qo_ps *p = (qo_ps *)calloc(1, sizeof(*p));
    p->sample_rate = sample_rate; p->playback_rate = playback_rate; p->t = *t;
    for (i = 0; i < 3; i++) {
        p->bank[i] = qo_rx_create(sample_rate, t);
        p->filtI[i] = (double *)calloc(MAX_FILTER_SIZE, sizeof(double));
        p->filtQ[i] = (double *)calloc(MAX_FILTER_SIZE, sizeof(double));
        p->filter_bandwidth[i] = 0;
    }
    p->rx_mode = QO_USB; p->play_channel = -1; p->old_play = 0;     /* "static int old_multirx_play_channel = 0", quisk.c:2303 */
    p->txrx_silence_ms = 50; p->agc_release_gain = 80.0;            /* agcReleaseGain, quisk.c:191 */
    p->rxTV[0] = p->txTV[0] = p->aux1TV[0] = p->aux2TV[0] = 1.0;
    p->sidetoneV[0] = BIG_VOLUME; p->sidetonePhase[0] = 1.0;
    p->keyupEnvelope = 1.0;
    qo_hb45_init(&p->HalfBand7); qo_hb45_init(&p->HalfBand8); qo_hb45_init(&p->HalfBand9);
    p->Agc1 = qo_agc_create(playback_rate, 0.7, 1.0);               /* {0.7, 0, 0}: sample_rate 0 -> playback_rate, quisk.c:2174-2175 */
    p->Agc2 = qo_agc_create(playback_rate, 0.7, 1.0);
    p->Agc3 = qo_agc_create(playback_rate, 0.7, 1.0);
    p->ttV[0] = 21474836.47; p->audioV[0] = 1.0;                    /* quisk.c:1263-1264 */
    p->fd_dindex = 1;
    qo_hb45_init(&p->mfHB1); qo_hb45_init(&p->mfHB2); qo_hb45_init(&p->mfHB3);
    return p;
}

void qo_ps_free(qo_ps *p)
{
    int i;
    if (!p) return;
    for (i = 0; i < 3; i++) { qo_rx_free(p->bank[i]); free(p->filtI[i]); free(p->filtQ[i]); }
    for (i = 0; i < QO_MAX_SUB; i++) free(p->sub_samples[i]);
    qo_agc_free(p->Agc1); qo_agc_free(p->Agc2); qo_agc_free(p->Agc3);
    qo_nb_free(p->nb);
    free(p->mf_samples); free(p->mf_window); free(p->mf_average);
    free(p->dsamples); free(p->dsamples2); free(p->orig); free(p->bufc); free(p->sub1_out);
    free(p);
}

void qo_ps_set_tune(qo_ps *p, int rx, int tx) { p->rx_tune = rx; p->tx_tune = tx; }                /* set_tune, quisk.c:4702 */
void qo_ps_set_mode(qo_ps *p, int mode) { p->rx_mode = mode; }                                      /* set_rx_mode, quisk.c:4621 */
void qo_ps_set_filters(qo_ps *p, const double *fI, const double *fQ, int size, int bw, int nFilter) /* set_filters, quisk.c:4551 */
{
    p->filter_bandwidth[nFilter] = bw;
    memcpy(p->filtI[nFilter], fI, (size_t)size * sizeof(double));
    memcpy(p->filtQ[nFilter], fQ, (size_t)size * sizeof(double));
    p->sizeFilter = size;
}
void qo_ps_set_agc(qo_ps *p, double level) { p->agc_release_gain = level; }                         /* set_agc, quisk.c:4543 */
void qo_ps_set_split_rxtx(qo_ps *p, int s) { p->split_rxtx = s; }
void qo_ps_set_multirx_play_channel(qo_ps *p, int ch) { p->play_channel = ch >= QO_MAX_SUB ? -1 : ch; }     /* quisk.c:4856 */
void qo_ps_set_multirx_play_method(qo_ps *p, int m) { p->play_method = m; }
void qo_ps_set_multirx_freq(qo_ps *p, int i, int f) { if (i >= 0 && i < QO_MAX_SUB) p->sub_freq[i] = f; }
void qo_ps_set_multirx_mode(qo_ps *p, int i, int m) { if (i >= 0 && i < QO_MAX_SUB) p->sub_mode[i] = m; }
void qo_ps_set_multirx_count(qo_ps *p, int n) { p->multirx_count = n; }                             /* quisk_multirx_count */
void qo_ps_set_sub_rx1_output(qo_ps *p, int on) { p->sub_rx1_driver = on; }                         /* quiskPlaybackDevices[QUISK_INDEX_SUB_RX1]->driver */
void qo_ps_multirx_samples(qo_ps *p, int i, const double *x, int n)                                 /* multirx_cSamples[i] of the coming call */
{
    if (i < 0 || i >= QO_MAX_SUB) return;
    if (n > p->sub_cap[i]) { free(p->sub_samples[i]); p->sub_samples[i] = (double *)malloc((size_t)n * 2 * sizeof(double)); p->sub_cap[i] = n; }
    memcpy(p->sub_samples[i], x, (size_t)n * 2 * sizeof(double));
}
void qo_ps_set_key_state(qo_ps *p, int key_down, int cw_key_down, int active_sidetone, int is_fdx)
{
    p->key_down = key_down; p->cw_key_down = cw_key_down; p->active_sidetone = active_sidetone; p->is_fdx = is_fdx;
}
void qo_ps_set_sidetone(qo_ps *p, double volume, int rit_freq, int txrx_silence_ms)                 /* set_sidetone, quisk.c:4710-4719 */
{
    double a = 2.0 * M_PI * abs(rit_freq) / p->playback_rate;
    p->sidetone_volume = volume; p->rit_freq = rit_freq;
    p->sidetonePhase[0] = cos(a); p->sidetonePhase[1] = sin(a);
    if (txrx_silence_ms >= 0) p->txrx_silence_ms = txrx_silence_ms;
    if (p->bank[0]->notch && (p->rx_mode == QO_CWL || p->rx_mode == QO_CWU)) qo_notch_init(p->bank[0]->notch);      /* quisk.c:4716-4717 */
    p->bank[0]->rit_freq = rit_freq;
}
void qo_ps_set_kill_audio(qo_ps *p, int k) { p->kill_audio = k; }
void qo_ps_invert_spectrum(qo_ps *p, int inv) { p->invert = inv; }
void qo_ps_set_noise_blanker(qo_ps *p, int level) { p->nb_level = level; }
void qo_ps_set_auto_notch(qo_ps *p, int on) { qo_rx_set_auto_notch(p->bank[0], on, p->rit_freq); }  /* set_auto_notch, quisk.c:4596 */
void qo_ps_set_squelch(qo_ps *p, double level) { int i; for (i = 0; i < 3; i++) qo_rx_set_squelch(p->bank[i], level); }
void qo_ps_set_ssb_squelch(qo_ps *p, int enabled, int level) { int i; for (i = 0; i < 3; i++) qo_rx_set_ssb_squelch(p->bank[i], enabled, level); }
void qo_ps_add_tone(qo_ps *p, int freq)                                                             /* add_tone, quisk.c:3203-3216 */
{
    if (freq && p->sample_rate) { double a = 2.0 * M_PI * freq / p->sample_rate; p->tt_phase_re = cos(a); p->tt_phase_im = sin(a); }
    else { p->tt_phase_re = 0; p->tt_phase_im = 0; }
}
double qo_ps_measure_frequency(qo_ps *p, int mode)                                                  /* measure_frequency, quisk.c:3181-3191 */
{
    if (mode >= 0) p->measure_freq_mode = mode;
    return p->measured_frequency;
}
void qo_ps_set_graph(qo_ps *p, qo_graph *g) { p->graph = g; }
void qo_ps_set_wdsp(qo_ps *p, wo_shim *s, wo_fexchange0_fn fn, void *ctx) { p->shim = s; p->wdsp_fn = fn; p->wdsp_ctx = ctx; }
int qo_ps_sub_rx1_audio(qo_ps *p, double *out, int cap)            /* what play_sound_interface got for QUISK_INDEX_SUB_RX1 in the last call */
{
    int n = p->sub1_n < cap ? p->sub1_n : cap;
    if (n > 0) memcpy(out, p->sub1_out, (size_t)n * 2 * sizeof(double));
    return p->sub1_n;
}
/* TEST HOOK, not in the reference: forget bank `b`'s filter storage (decimators, demodulator, Rx filter rings), as the GPU
 * library does when a mode change makes it rebuild the bank; the tune vectors, AGCs and everything else carry on. */
void qo_ps_restart_bank(qo_ps *p, int b)
{
    int inited;
    if (b < 0 || b > 2) return;
    inited = p->bank[b]->sq_inited;
    qo_rx_free(p->bank[b]);
    p->bank[b] = qo_rx_create(p->sample_rate, &p->t);
    p->bank[b]->sq_inited = inited;
    p->bank[b]->rit_freq = p->rit_freq;
}
int qo_ps_squelch_flags(const qo_ps *p) { return p->squelch_real | (p->squelch_imag << 1); }
int qo_ps_overrun(const qo_ps *p) { return p->b2c_overrun; }

static void cmul_inplace(double *a, const double *b)   /* a *= b */
{
    double t = a[0] * b[0] - a[1] * b[1];
    a[1] = a[0] * b[1] + a[1] * b[0];
    a[0] = t;
}

static void ps_tune(double *x, int n, double *vec, double freq, int rate)       /* quisk.c:2483-2488 and its three copies */
{
    double a = -2.0 * M_PI * freq / rate, ph[2];
    int i;
    ph[0] = cos(a); ph[1] = sin(a);
    for (i = 0; i < n; i++) { cmul_inplace(x + 2 * i, vec); cmul_inplace(vec, ph); }
}

static void AddTestTone(qo_ps *p, double *x, int n)                             /* quisk.c:1258-1303 */
{
    int i;
    double tp[2], ap[2], a, v[2], e[2];
    tp[0] = p->tt_phase_re; tp[1] = p->tt_phase_im;
    a = 2.0 * M_PI * 1000 / p->sample_rate; ap[0] = cos(a); ap[1] = sin(a);
    switch (p->rx_mode) {
    default:
        for (i = 0; i < n; i++) { x[2 * i] += p->ttV[0]; x[2 * i + 1] += p->ttV[1]; cmul_inplace(p->ttV, tp); }
        break;
    case QO_AM:
        for (i = 0; i < n; i++) {
            double g = 1.0 + p->audioV[0];
            x[2 * i] += p->ttV[0] * g; x[2 * i + 1] += p->ttV[1] * g;
            cmul_inplace(p->ttV, tp); cmul_inplace(p->audioV, ap);
        }
        break;
    case QO_FM: case QO_DGT_FM:
        for (i = 0; i < n; i++) {
            e[0] = cos(p->audioV[0]); e[1] = sin(p->audioV[0]);                  /* cexp(I * creal(audioVector)) */
            v[0] = p->ttV[0]; v[1] = p->ttV[1];
            cmul_inplace(v, e);
            x[2 * i] += v[0]; x[2 * i + 1] += v[1];
            cmul_inplace(p->ttV, tp); cmul_inplace(p->audioV, ap);
        }
        break;
    }
}

static int Buffer2Chan(qo_ps *p, double *samp1, int count1, double *samp2, int count2)      /* quisk.c:1577-1611 */
{
    int nout;
    if (samp1 == NULL) { p->nbuf1 = p->nbuf2 = 0; return 0; }
    if (p->nbuf1 == 0 && p->nbuf2 == 0 && count1 == count2) return count1;
    if (count1 + p->nbuf1 >= BUF2CHAN_SIZE || count2 + p->nbuf2 >= BUF2CHAN_SIZE) p->nbuf1 = p->nbuf2 = 0;
    if (count1 > BUF2CHAN_SIZE || count2 > BUF2CHAN_SIZE) {
        /* the reference copies on regardless and runs over its static arrays (undefined behaviour; Quisk's own blocks are far shorter):
         * the restatement stops here and says so (qo_ps_overrun), callers keep their blocks below 12 000 audio samples */
        p->b2c_overrun = 1;
        return count1 < count2 ? count1 : count2;
    }
    memcpy(p->buf1 + p->nbuf1, samp1, (size_t)count1 * sizeof(double)); p->nbuf1 += count1;
    memcpy(p->buf2 + p->nbuf2, samp2, (size_t)count2 * sizeof(double)); p->nbuf2 += count2;
    nout = p->nbuf1 <= p->nbuf2 ? p->nbuf1 : p->nbuf2;
    memcpy(samp1, p->buf1, (size_t)nout * sizeof(double)); p->nbuf1 -= nout;
    memmove(p->buf1, p->buf1 + nout, (size_t)p->nbuf1 * sizeof(double));
    memcpy(samp2, p->buf2, (size_t)nout * sizeof(double)); p->nbuf2 -= nout;
    memmove(p->buf2, p->buf2 + nout, (size_t)p->nbuf2 * sizeof(double));
    return nout;
}

static int cFracDecim(qo_ps *p, double *x, int n, double fdecim)               /* quisk.c:622-665 */
{
    int i, nout = 0;
    double xm0, xm1, xm2, xm3, c3[2];
    for (i = 0; i < n; i++) {
        c3[0] = x[2 * i]; c3[1] = x[2 * i + 1];
        if (p->fd_dindex < 2) {
            int k;
            xm0 = p->fd_dindex - 0; xm1 = p->fd_dindex - 1; xm2 = p->fd_dindex - 2; xm3 = p->fd_dindex - 3;
            for (k = 0; k < 2; k++)
                x[2 * nout + k] = xm1 * xm2 * xm3 * p->fd_c0[k] / -6.0 + xm0 * xm2 * xm3 * p->fd_c1[k] / 2.0 +
                                  xm0 * xm1 * xm3 * p->fd_c2[k] / -2.0 + xm0 * xm1 * xm2 * c3[k] / 6.0;
            nout++;
            p->fd_dindex += fdecim - 1;
        } else {
            p->fd_dindex -= 1;
        }
        p->fd_c0[0] = p->fd_c1[0]; p->fd_c0[1] = p->fd_c1[1];
        p->fd_c1[0] = p->fd_c2[0]; p->fd_c1[1] = p->fd_c2[1];
        p->fd_c2[0] = c3[0]; p->fd_c2[1] = c3[1];
    }
    return nout;
}

static void measure_freq(qo_ps *p, const double *x, int n, int srate)           /* quisk.c:5579-5649 */
{
    int i, k, center, ipeak;
    double dmax, c3, freq, *buf, *avg;
    const int N = MF_FFT_SIZE;
    if (!p->mf_samples) {                                                        /* measure_freq(NULL, 0, 0) from record_app */
        p->mf_samples = (double *)calloc((size_t)N * 2, sizeof(double));
        p->mf_window = (double *)malloc((size_t)(N + 1) * sizeof(double));
        p->mf_average = (double *)calloc((size_t)N, sizeof(double));
        for (i = 0; i < N; i++) p->mf_window[i] = 0.50 - 0.50 * cos(2. * M_PI * i / (N - 1));
    }
    avg = p->mf_average;
    buf = (double *)malloc((size_t)(n > 0 ? n : 1) * 2 * sizeof(double));
    memcpy(buf, x, (size_t)n * 2 * sizeof(double));                             /* do not destroy cSamples */
    n = qo_cDecim2HB45(buf, n, &p->mfHB1);
    n = qo_cDecim2HB45(buf, n, &p->mfHB2);
    n = qo_cDecim2HB45(buf, n, &p->mfHB3);
    srate /= 8;
    for (i = 0; i < n && p->mf_index < N; i++, p->mf_index++) {
        p->mf_samples[2 * p->mf_index] = buf[2 * i]; p->mf_samples[2 * p->mf_index + 1] = buf[2 * i + 1];
    }
    free(buf);
    if (p->mf_index < N) return;                                                /* wait for a full array (the rest of the block is dropped) */
    for (i = 0; i < N; i++) { p->mf_samples[2 * i] *= p->mf_window[i]; p->mf_samples[2 * i + 1] *= p->mf_window[i]; }
    fo_fft(p->mf_samples, N, -1);
    p->mf_index = 0;
    p->mf_count++;
    k = 0;
    for (i = N / 2; i < N; i++) avg[k++] += hypot(p->mf_samples[2 * i], p->mf_samples[2 * i + 1]);
    for (i = 0; i < N / 2; i++) avg[k++] += hypot(p->mf_samples[2 * i], p->mf_samples[2 * i + 1]);
    if (p->mf_count < p->measure_freq_mode / 2) return;
    p->mf_count = 0;
    dmax = 1.e-20; ipeak = 0;
    center = N / 2 - p->rit_freq * N / srate;
    k = 500;
    k = k * N / srate;
    for (i = center - k; i <= center + k; i++)
        if (avg[i] > dmax) { dmax = avg[i]; ipeak = i; }
    c3 = 1.36 * (avg[ipeak + 1] - avg[ipeak - 1]) / (avg[ipeak - 1] + avg[ipeak] + avg[ipeak + 1]);
    freq = srate * (2 * (ipeak + c3) - N) / 2 / N;
    freq += p->rx_tune;
    p->measured_frequency = freq;
    memset(avg, 0, sizeof(double) * N);
}

/* the bank's filter set for this call: cFilterI/Q[nFilter] with the one global sizeFilter; the rings stay the bank's.
 * ssb_squelch's `plan` is ONE function static for all banks (quisk.c:1093,1104): the first call of any bank creates it. */
static void ps_load_filter(qo_ps *p, int bank, int nFilter, int mode)
{
    qo_rx *r = p->bank[bank];
    r->sq_inited = p->bank[0]->sq_inited | p->bank[1]->sq_inited | p->bank[2]->sq_inited;
    memcpy(r->filtI, p->filtI[nFilter], (size_t)p->sizeFilter * sizeof(double));
    memcpy(r->filtQ, p->filtQ[nFilter], (size_t)p->sizeFilter * sizeof(double));
    r->sizeFilter = p->sizeFilter;
    r->bandwidth = p->filter_bandwidth[nFilter];
    r->bandwidth0 = p->filter_bandwidth[0];
    r->mode = mode;
}

static int ps_take(qo_ps *p, int n)                                             /* quisk.c:2372-2375 */
{
    int nout;
    p->dOutCounter += (double)n * p->playback_rate / p->sample_rate;
    nout = (int)p->dOutCounter;
    p->dOutCounter -= nout;
    return nout;
}

int qo_ps_process(qo_ps *p, double *x, int n)
{
    int i, nout, orig_n, n2, decim_srate, stereo;
    double d, di, env_step = 1. / (p->playback_rate * 5e-3);
    double *ds, *ds2;
    p->sub1_n = 0;
    if (n <= 0) return n;                                                       /* quisk.c:2336-2337 */
    if (n > p->cap) {
        p->cap = n * 2;
        free(p->dsamples); free(p->dsamples2); free(p->orig); free(p->bufc); free(p->sub1_out);
        p->dsamples = (double *)malloc((size_t)p->cap * 8 * sizeof(double));    /* room for the interpolated count */
        p->dsamples2 = (double *)malloc((size_t)p->cap * 8 * sizeof(double));
        p->orig = (double *)malloc((size_t)p->cap * 16 * sizeof(double));
        p->bufc = (double *)malloc((size_t)p->cap * 2 * sizeof(double));
        p->sub1_out = (double *)malloc((size_t)p->cap * 2 * sizeof(double));
    }
    ds = p->dsamples; ds2 = p->dsamples2;
    orig_n = n;
    if (p->split_rxtx) {                                                        /* quisk.c:2361-2365 */
        memcpy(p->orig, x, (size_t)n * 2 * sizeof(double));
        if (!p->old_split) Buffer2Chan(p, NULL, 0, NULL, 0);
    }
    if (p->play_channel != p->old_play) Buffer2Chan(p, NULL, 0, NULL, 0);
    p->old_split = p->split_rxtx;
    p->old_play = p->play_channel;
    if (p->key_down && !p->is_fdx) {                                            /* quisk.c:2371-2400 */
        nout = ps_take(p, n);
        p->playSilence = (int)(p->playback_rate * 1E-3 * p->txrx_silence_ms);
        p->keyupEnvelope = 0;
        if (p->active_sidetone == 2 && p->cw_key_down) {
            if (!p->sidetoneIsOn) { p->sidetoneIsOn = 1; p->sidetoneEnvelope = 0; p->sidetoneV[0] = BIG_VOLUME; p->sidetoneV[1] = 0; }
            for (i = 0; i < nout; i++) {
                if (p->sidetoneEnvelope < 1.0) { p->sidetoneEnvelope += env_step; if (p->sidetoneEnvelope > 1.0) p->sidetoneEnvelope = 1.0; }
                d = p->sidetoneV[0] * p->sidetone_volume * p->sidetoneEnvelope;
                x[2 * i] = d; x[2 * i + 1] = d;
                cmul_inplace(p->sidetoneV, p->sidetonePhase);
            }
        } else {
            for (i = 0; i < nout; i++) { x[2 * i] = 0; x[2 * i + 1] = 0; }
        }
        return nout;
    }
    if (p->sidetoneIsOn) {                                                      /* quisk.c:2402-2421 */
        nout = ps_take(p, n);
        for (i = 0; i < nout; i++) {
            p->sidetoneEnvelope -= env_step;
            if (p->sidetoneEnvelope < 0) { p->sidetoneIsOn = 0; p->sidetoneEnvelope = 0; break; }
            d = p->sidetoneV[0] * p->sidetone_volume * p->sidetoneEnvelope;
            x[2 * i] = d; x[2 * i + 1] = d;
            cmul_inplace(p->sidetoneV, p->sidetonePhase);
        }
        for (; i < nout; i++) { x[2 * i] = 0; x[2 * i + 1] = 0; p->playSilence--; }
        return nout;
    }
    if (p->playSilence > 0) {                                                   /* quisk.c:2422-2433 */
        nout = ps_take(p, n);
        for (i = 0; i < nout; i++) { x[2 * i] = 0; x[2 * i + 1] = 0; }
        p->playSilence -= nout;
        return nout;
    }
    if (p->tt_phase_re != 0 || p->tt_phase_im != 0) AddTestTone(p, x, n);      /* quisk.c:2438-2439 */
    if (p->invert) for (i = 0; i < n; i++) x[2 * i + 1] = -x[2 * i + 1];        /* quisk.c:2441-2446 */
    if (!p->key_down) {                                                         /* quisk.c:2448-2449 */
        if (!p->nb) p->nb = qo_nb_create(p->sample_rate);
        qo_nb_set_level(p->nb, p->nb_level);
        qo_nb_process(p->nb, x, n);
    }
    if (p->graph) qo_graph_feed(p->graph, x, n);                                /* quisk.c:2454-2475 */
    if (p->rx_tune != 0) ps_tune(x, n, p->rxTV, p->rx_tune, p->sample_rate);    /* quisk.c:2477-2488 */
    ps_load_filter(p, 0, 0, p->rx_mode);
    n = process_decimate(p->bank[0], x, n);                                     /* quisk.c:2518 */
    decim_srate = p->bank[0]->decim_srate;
    if (p->measure_freq_mode) measure_freq(p, x, n, decim_srate);               /* quisk.c:2527-2528 */
    n = process_demodulate(p->bank[0], x, ds, n);                               /* quisk.c:2530 */
    p->squelch_real = p->squelch_imag = 0;
    stereo = p->rx_mode == QO_DGT_IQ;
    if (stereo) {
        ;                                                                       /* this mode is already stereo */
    } else if (p->split_rxtx) {                                                 /* quisk.c:2537-2590 */
        int s0 = p->bank[0]->squelch_active, s1, first_real;
        ps_tune(p->orig, orig_n, p->txTV, p->tx_tune + p->rit_freq, p->sample_rate);
        ps_load_filter(p, 1, 0, p->rx_mode);
        n2 = process_decimate(p->bank[1], p->orig, orig_n);
        n2 = process_demodulate(p->bank[1], p->orig, ds2, n2);
        s1 = p->bank[1]->squelch_active;
        n = Buffer2Chan(p, ds, n, ds2, n2);
        switch (p->split_rxtx) {
        default:
        case 1: first_real = p->tx_tune < p->rx_tune; goto two;
        case 2: first_real = p->tx_tune >= p->rx_tune;
        two:
            if (first_real) { p->squelch_real = s0; p->squelch_imag = s1; for (i = 0; i < n; i++) { x[2 * i] = ds[i]; x[2 * i + 1] = ds2[i]; } }
            else { p->squelch_real = s1; p->squelch_imag = s0; for (i = 0; i < n; i++) { x[2 * i] = ds2[i]; x[2 * i + 1] = ds[i]; } }
            break;
        case 3: p->squelch_real = p->squelch_imag = s0; for (i = 0; i < n; i++) { x[2 * i] = ds[i]; x[2 * i + 1] = ds[i]; } break;
        case 4: p->squelch_real = p->squelch_imag = s1; for (i = 0; i < n; i++) { x[2 * i] = ds2[i]; x[2 * i + 1] = ds2[i]; } break;
        }
    } else if (p->play_channel >= 0 && p->sub_samples[p->play_channel]) {       /* quisk.c:2591-2621 */
        int pc = p->play_channel, s0 = p->bank[0]->squelch_active, s1;
        memcpy(p->bufc, p->sub_samples[pc], (size_t)orig_n * 2 * sizeof(double));
        ps_tune(p->bufc, orig_n, p->aux1TV, p->sub_freq[pc], p->sample_rate);
        ps_load_filter(p, 1, 1, p->sub_mode[pc]);
        n2 = process_decimate(p->bank[1], p->bufc, orig_n);
        n2 = process_demodulate(p->bank[1], p->bufc, ds2, n2);
        s1 = p->bank[1]->squelch_active;
        n = Buffer2Chan(p, ds, n, ds2, n2);
        switch (p->play_method) {
        default:
        case 0: p->squelch_real = p->squelch_imag = s1; for (i = 0; i < n; i++) { x[2 * i] = ds2[i]; x[2 * i + 1] = ds2[i]; } break;
        case 1: p->squelch_real = s0; p->squelch_imag = s1; for (i = 0; i < n; i++) { x[2 * i] = ds[i]; x[2 * i + 1] = ds2[i]; } break;
        case 2: p->squelch_real = s1; p->squelch_imag = s0; for (i = 0; i < n; i++) { x[2 * i] = ds2[i]; x[2 * i + 1] = ds[i]; } break;
        }
    } else {                                                                    /* quisk.c:2622-2628 */
        p->squelch_real = p->squelch_imag = p->bank[0]->squelch_active;
        for (i = 0; i < n; i++) { x[2 * i] = ds[i]; x[2 * i + 1] = ds[i]; }
    }
    {                                                                           /* sub-receiver 1 on a digital output device, quisk.c:2630-2651 */
        int m = p->sub_mode[0];
        if (p->multirx_count > 0 && (m == QO_DGT_U || m == QO_DGT_L || m == QO_DGT_IQ || m == QO_DGT_FM) && p->sub_rx1_driver &&
            p->sub_samples[0]) {
            double *s = p->sub_samples[0];
            ps_tune(s, orig_n, p->aux2TV, p->sub_freq[0], p->sample_rate);
            ps_load_filter(p, 2, 2, m);
            n2 = process_decimate(p->bank[2], s, orig_n);
            n2 = process_demodulate(p->bank[2], s, ds2, n2);
            if (m == QO_DGT_IQ) {
                qo_agc_process(p->Agc3, s, n2, 1, p->agc_release_gain);
            } else {
                for (i = 0; i < n2; i++) { s[2 * i] = ds2[i]; s[2 * i + 1] = ds2[i]; }
                qo_agc_process(p->Agc3, s, n2, 0, p->agc_release_gain);
            }
            memcpy(p->sub1_out, s, (size_t)n2 * 2 * sizeof(double));            /* play_sound_interface(..., n, multirx_cSamples[0], ...) */
            p->sub1_n = n2;
        }
    }
    for (i = 0; i < QO_MAX_SUB; i++) if (p->sub_cap[i]) { free(p->sub_samples[i]); p->sub_samples[i] = NULL; p->sub_cap[i] = 0; }
    if (decim_srate != 48000)                                                   /* quisk.c:2654-2659 */
        n = cFracDecim(p, x, n, decim_srate / 48000.0);
    if (p->shim && p->wdsp_fn) n = wo_shim_fexchange0(p->shim, p->wdsp_fn, p->wdsp_ctx, x, n);      /* quisk.c:2660-2661 */
    switch (p->playback_rate / 48000) {                                         /* quisk.c:2663-2682 */
    case 1: break;
    case 2: n = qo_cInterp2HB45(x, n, &p->HalfBand7); break;
    case 4: n = qo_cInterp2HB45(x, n, &p->HalfBand7); n = qo_cInterp2HB45(x, n, &p->HalfBand8); break;
    case 8: n = qo_cInterp2HB45(x, n, &p->HalfBand7); n = qo_cInterp2HB45(x, n, &p->HalfBand8); n = qo_cInterp2HB45(x, n, &p->HalfBand9); break;
    default: break;
    }
    if (stereo) {                                                               /* quisk.c:2686-2701 */
        qo_agc_process(p->Agc1, x, n, 1, p->agc_release_gain);
    } else if (p->split_rxtx || p->play_channel >= 0) {
        for (i = 0; i < n; i++) { p->orig[2 * i] = x[2 * i + 1]; p->orig[2 * i + 1] = 0; x[2 * i + 1] = 0; }
        qo_agc_process(p->Agc1, x, n, 0, p->agc_release_gain);
        qo_agc_process(p->Agc2, p->orig, n, 0, p->agc_release_gain);
        for (i = 0; i < n; i++) x[2 * i + 1] = p->orig[2 * i];
    } else {
        qo_agc_process(p->Agc1, x, n, 0, p->agc_release_gain);
    }
    if (p->kill_audio) {                                                        /* quisk.c:2712-2728 */
        p->squelch_real = p->squelch_imag = 1;
        for (i = 0; i < n; i++) { x[2 * i] = 0; x[2 * i + 1] = 0; }
    } else if (p->squelch_real && p->squelch_imag) {
        for (i = 0; i < n; i++) { x[2 * i] = 0; x[2 * i + 1] = 0; }
    } else if (p->squelch_imag) {
        for (i = 0; i < n; i++) x[2 * i + 1] = 0;
    } else if (p->squelch_real) {
        for (i = 0; i < n; i++) x[2 * i] = 0;
    }
    if (p->keyupEnvelope < 1.0) {                                               /* quisk.c:2729-2738 */
        di = env_step;
        for (i = 0; i < n; i++) {
            p->keyupEnvelope += di;
            if (p->keyupEnvelope > 1.0) { p->keyupEnvelope = 1.0; break; }
            x[2 * i] *= p->keyupEnvelope; x[2 * i + 1] *= p->keyupEnvelope;
        }
    }
    return n;
}


/* get_filter (quisk.c:5481-5568): the response of the Rx filter cFilterI/Q[0] as the "RX Filter" screen draws it -- a multitone
 * through the cRxFilterOut loop (its own copy, :5516-5530), the FIRST data_width entries of fft_window (record_app's window of
 * fft_size points, quisk.c:6003-6009), a data_width-point transform, 20 log10 with a floor of -140 dB, negative frequencies first. */
void qo_get_filter(const double *filtI, const double *filtQ, int sizeFilter, int data_width, int fft_size, double *out)
{
    const int total = data_width + sizeFilter;
    double *average = (double *)malloc(sizeof(double) * (size_t)total);
    double *bufI = (double *)calloc((size_t)(sizeFilter > 0 ? sizeFilter : 1), sizeof(double));
    double *bufQ = (double *)calloc((size_t)(sizeFilter > 0 ? sizeFilter : 1), sizeof(double));
    double *samples = (double *)calloc((size_t)data_width * 2, sizeof(double));
    int i, j, k, n, freq, time;
    for (i = 0; i < total; i++) average[i] = 0.5;
    for (freq = 1; freq < data_width / 2.0 - 10.0; freq++) {
        const double delta = 2 * M_PI / data_width * freq;
        double phase = 0;
        for (time = 0; time < total; time++) {
            average[time] += cos(phase);
            phase += delta;
            if (phase > 2 * M_PI) phase -= 2 * M_PI;
        }
    }
    n = 0;
    for (time = 0; time < total; time++) {
        const double d2 = average[time];
        double accI = 0, accQ = 0;
        bufI[n] = d2; bufQ[n] = d2;
        j = n;
        for (k = 0; k < sizeFilter; k++) {
            accI += bufI[j] * filtI[k];
            accQ += bufQ[j] * filtQ[k];
            if (++j >= sizeFilter) j = 0;
        }
        if (++n >= sizeFilter) n = 0;
        if (time >= sizeFilter) { samples[2 * (time - sizeFilter)] = accI; samples[2 * (time - sizeFilter) + 1] = accQ; }
    }
    for (i = 0; i < data_width; i++) {
        const double w = 0.5 + 0.5 * cos(2. * M_PI * (i - fft_size / 2) / fft_size);      /* fft_window[i], quisk.c:6008 */
        samples[2 * i] *= w; samples[2 * i + 1] *= w;
    }
    fo_fft(samples, data_width, -1);
    for (k = 0; k < data_width; k++) {
        double a = hypot(samples[2 * k], samples[2 * k + 1]) * (1. / data_width);
        average[k] = a <= 1e-7 ? -7 : log10(a);
    }
    i = 0;
    for (k = data_width / 2; k < data_width; k++, i++) out[i] = 20.0 * average[k];
    for (k = 0; k < data_width / 2; k++, i++) out[i] = 20.0 * average[k];
    free(samples); free(bufQ); free(bufI); free(average);
}
