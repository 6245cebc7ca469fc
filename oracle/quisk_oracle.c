/* quisk_oracle.c -- TEST INFRASTRUCTURE ONLY.  See quisk_oracle.h.
 *
 * The arithmetic order of every multiply-accumulate follows the reference so that
 * the comparison with oracle/_ref/libquisk_filter_ref.so can be bit-exact:
 * tap k multiplies the sample k positions back, k ascending, accumulator starts at 0,
 * complex x real products are (re*c, im*c).  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "quisk_oracle.h"

#define QO_OUT_CAP (QO_SAMP_BUFFER_SIZE * 8 / 10)    /* filter.c:158,194,315 */

void qo_fir_init(qo_fir *f, const double *taps, int ntaps, int is_complex)
{
    f->taps = taps;
    f->ctaps = NULL;
    f->ntaps = ntaps;
    f->phase = 0;
    f->pos = 0;
    f->is_complex = is_complex;
    f->hist = (double *)calloc((size_t)ntaps * (is_complex ? 2 : 1), sizeof(double));
}

void qo_fir_free(qo_fir *f)
{
    free(f->hist); free(f->ctaps);
    f->hist = NULL; f->ctaps = NULL;
}

void qo_fir_tune(qo_fir *f, double freq, int ssb_upper)
{
    int i;
    double D, w;
    if (!f->ctaps) f->ctaps = (double *)malloc((size_t)f->ntaps * 2 * sizeof(double));
    w = 2.0 * M_PI * freq;                          /* tune = I * 2.0 * M_PI * freq */
    D = (f->ntaps - 1.0) / 2.0;
    for (i = 0; i < f->ntaps; i++) {
        double a = w * (i - D);
        double re = cos(a) * f->taps[i];
        double im = sin(a) * f->taps[i];
        if (ssb_upper) { f->ctaps[2 * i] = re; f->ctaps[2 * i + 1] = im; }
        else           { f->ctaps[2 * i] = im; f->ctaps[2 * i + 1] = re; }
    }
}

/* one complex output: sum_k h[k0 + k*kstep] * hist[pos - k], nk terms */
static inline void cmac_real(const qo_fir *f, int k0, int kstep, int nk, double *re, double *im)
{
    int k, idx = f->pos;
    double ar = 0.0, ai = 0.0;
    const double *h = f->taps + k0;
    for (k = 0; k < nk; k++, h += kstep) {
        ar += f->hist[2 * idx] * *h;
        ai += f->hist[2 * idx + 1] * *h;
        if (--idx < 0) idx = f->ntaps - 1;
    }
    *re = ar; *im = ai;
}

static inline double dmac_real(const qo_fir *f, int k0, int kstep, int nk)
{
    int k, idx = f->pos;
    double a = 0.0;
    const double *h = f->taps + k0;
    for (k = 0; k < nk; k++, h += kstep) {
        a += f->hist[idx] * *h;
        if (--idx < 0) idx = f->ntaps - 1;
    }
    return a;
}

static inline void advance(qo_fir *f) { if (++f->pos >= f->ntaps) f->pos = 0; }

int qo_cDecimate(double *x, int count, qo_fir *f, int decim)
{
    int i, nout = 0;
    for (i = 0; i < count; i++) {
        f->hist[2 * f->pos] = x[2 * i];
        f->hist[2 * f->pos + 1] = x[2 * i + 1];
        if (++f->phase >= decim) {
            double re, im;
            f->phase = 0;
            cmac_real(f, 0, 1, f->ntaps, &re, &im);
            x[2 * nout] = re; x[2 * nout + 1] = im;
            nout++;
        }
        advance(f);
    }
    return nout;
}

int qo_cCDecimate(double *x, int count, qo_fir *f, int decim)
{
    int i, k, nout = 0;
    for (i = 0; i < count; i++) {
        f->hist[2 * f->pos] = x[2 * i];
        f->hist[2 * f->pos + 1] = x[2 * i + 1];
        if (++f->phase >= decim) {
            int idx = f->pos;
            double ar = 0.0, ai = 0.0;
            f->phase = 0;
            for (k = 0; k < f->ntaps; k++) {
                double xr = f->hist[2 * idx], xi = f->hist[2 * idx + 1];
                double cr = f->ctaps[2 * k], ci = f->ctaps[2 * k + 1];
                ar += xr * cr - xi * ci;
                ai += xr * ci + xi * cr;
                if (--idx < 0) idx = f->ntaps - 1;
            }
            x[2 * nout] = ar; x[2 * nout + 1] = ai;
            nout++;
        }
        advance(f);
    }
    return nout;
}

int qo_dDecimate(double *x, int count, qo_fir *f, int decim)
{
    int i, nout = 0;
    for (i = 0; i < count; i++) {
        f->hist[f->pos] = x[i];
        if (++f->phase >= decim) {
            f->phase = 0;
            x[nout++] = dmac_real(f, 0, 1, f->ntaps);
        }
        advance(f);
    }
    return nout;
}

int qo_cInterpolate(double *x, int count, qo_fir *f, int interp)
{
    int i, j, nout = 0;
    double *tmp = (double *)malloc((size_t)(count > 0 ? count : 1) * 2 * sizeof(double));
    memcpy(tmp, x, (size_t)count * 2 * sizeof(double));
    for (i = 0; i < count; i++) {
        f->hist[2 * f->pos] = tmp[2 * i];
        f->hist[2 * f->pos + 1] = tmp[2 * i + 1];
        for (j = 0; j < interp; j++) {
            double re, im;
            cmac_real(f, j, interp, f->ntaps / interp, &re, &im);
            if (nout < QO_OUT_CAP) {
                x[2 * nout] = re * interp; x[2 * nout + 1] = im * interp;
                nout++;
            }
        }
        advance(f);
    }
    free(tmp);
    return nout;
}

int qo_dInterpolate(double *x, int count, qo_fir *f, int interp)
{
    int i, j, nout = 0;
    double *tmp = (double *)malloc((size_t)(count > 0 ? count : 1) * sizeof(double));
    memcpy(tmp, x, (size_t)count * sizeof(double));
    for (i = 0; i < count; i++) {
        f->hist[f->pos] = tmp[i];
        for (j = 0; j < interp; j++) {
            double v = dmac_real(f, j, interp, f->ntaps / interp);
            if (nout < QO_OUT_CAP) x[nout++] = v * interp;
        }
        advance(f);
    }
    free(tmp);
    return nout;
}

int qo_cInterpDecim(double *x, int count, qo_fir *f, int interp, int decim)
{
    int i, nout = 0;
    double *tmp = (double *)malloc((size_t)(count > 0 ? count : 1) * 2 * sizeof(double));
    memcpy(tmp, x, (size_t)count * 2 * sizeof(double));
    for (i = 0; i < count; i++) {
        f->hist[2 * f->pos] = tmp[2 * i];
        f->hist[2 * f->pos + 1] = tmp[2 * i + 1];
        while (f->phase < interp) {
            double re, im;
            cmac_real(f, f->phase, interp, f->ntaps / interp, &re, &im);
            if (nout < QO_OUT_CAP) {
                x[2 * nout] = re * interp; x[2 * nout + 1] = im * interp;
                nout++;
            }
            f->phase += decim;
        }
        advance(f);
        f->phase -= interp;
    }
    free(tmp);
    return nout;
}

int qo_dFilter(double *x, int count, qo_fir *f)
{
    int i;
    for (i = 0; i < count; i++) {
        f->hist[f->pos] = x[i];
        x[i] = dmac_real(f, 0, 1, f->ntaps);
        advance(f);
    }
    return count;
}

double qo_dD_out(double sample, qo_fir *f)
{
    double v;
    f->hist[f->pos] = sample;
    v = dmac_real(f, 0, 1, f->ntaps);
    advance(f);
    return v;
}

void qo_dC_out(double sample, qo_fir *f, double *out)
{
    int k, idx;
    double ar = 0.0, ai = 0.0;
    f->hist[f->pos] = sample;
    idx = f->pos;
    for (k = 0; k < f->ntaps; k++) {
        ar += f->hist[idx] * f->ctaps[2 * k];
        ai += f->hist[idx] * f->ctaps[2 * k + 1];
        if (--idx < 0) idx = f->ntaps - 1;
    }
    advance(f);
    out[0] = ar; out[1] = ai;
}

/* ---------------------------------------------------------------- half-band, 45 taps */
/* Rate 96, cutoff 16-24-32, 120 dB; coef[0] and [44] of the full filter are zero. */
const double qo_hb45_coef[12] = {
    0.000018566625444266, -0.000118469698701817, 0.000457318798253456,
    -0.001347840471412094, 0.003321838571445455, -0.007198422696929033,
    0.014211106939802483, -0.026424776824073383, 0.048414810444971007,
    -0.096214669073304823, 0.314881034738348550, 0.500000000000000000 };

void qo_hb45_init(qo_hb45 *f) { memset(f, 0, sizeof(*f)); }

int qo_cDecim2HB45(double *x, int count, qo_hb45 *f)
{
    int i, k, nout = 0;
    for (i = 0; i < count; i++) {
        if (f->toggle == 0) {
            f->toggle = 1;
            memmove(f->center + 2, f->center, sizeof(double) * 2 * 10);
            f->center[0] = x[2 * i]; f->center[1] = x[2 * i + 1];
        } else {
            double ar, ai;
            f->toggle = 0;
            memmove(f->samples + 2, f->samples, sizeof(double) * 2 * 21);
            f->samples[0] = x[2 * i]; f->samples[1] = x[2 * i + 1];
            ar = (f->samples[0] + f->samples[2 * 21]) * qo_hb45_coef[0];
            ai = (f->samples[1] + f->samples[2 * 21 + 1]) * qo_hb45_coef[0];
            for (k = 1; k < 11; k++) {
                ar += (f->samples[2 * k] + f->samples[2 * (21 - k)]) * qo_hb45_coef[k];
                ai += (f->samples[2 * k + 1] + f->samples[2 * (21 - k) + 1]) * qo_hb45_coef[k];
            }
            ar += f->center[2 * 10] * qo_hb45_coef[11];
            ai += f->center[2 * 10 + 1] * qo_hb45_coef[11];
            x[2 * nout] = ar; x[2 * nout + 1] = ai;
            nout++;
        }
    }
    return nout;
}

int qo_cInterp2HB45(double *x, int count, qo_hb45 *f)
{
    int i, k, nout = 0;
    const int ncoef = 12, nsamp = 22;
    double *tmp = (double *)malloc((size_t)(count > 0 ? count : 1) * 2 * sizeof(double));
    memcpy(tmp, x, (size_t)count * 2 * sizeof(double));
    for (i = 0; i < count; i++) {
        double ar = 0.0, ai = 0.0;
        memmove(f->samples + 2, f->samples, (size_t)(nsamp - 1) * 2 * sizeof(double));
        f->samples[0] = tmp[2 * i]; f->samples[1] = tmp[2 * i + 1];
        if (nout > QO_OUT_CAP) continue;
        x[2 * nout] = f->samples[2 * (ncoef - 1)] * qo_hb45_coef[ncoef - 1] * 2;
        x[2 * nout + 1] = f->samples[2 * (ncoef - 1) + 1] * qo_hb45_coef[ncoef - 1] * 2;
        nout++;
        for (k = 0; k < nsamp / 2; k++) {
            ar += (f->samples[2 * k] + f->samples[2 * (nsamp - 1 - k)]) * qo_hb45_coef[k];
            ai += (f->samples[2 * k + 1] + f->samples[2 * (nsamp - 1 - k) + 1]) * qo_hb45_coef[k];
        }
        x[2 * nout] = ar * 2; x[2 * nout + 1] = ai * 2;
        nout++;
    }
    free(tmp);
    return nout;
}

int qo_dInterp2HB45(double *x, int count, qo_hb45 *f)
{
    /* real data: uses samples[0..21] as 22 doubles (struct quisk_dHB45Filter, filter.h:31-37) */
    int i, k, nout = 0;
    const int ncoef = 12, nsamp = 22;
    double *tmp = (double *)malloc((size_t)(count > 0 ? count : 1) * sizeof(double));
    memcpy(tmp, x, (size_t)count * sizeof(double));
    for (i = 0; i < count; i++) {
        double a = 0.0;
        memmove(f->samples + 1, f->samples, (size_t)(nsamp - 1) * sizeof(double));
        f->samples[0] = tmp[i];
        if (nout > QO_OUT_CAP) continue;
        x[nout++] = f->samples[ncoef - 1] * qo_hb45_coef[ncoef - 1] * 2;
        for (k = 0; k < nsamp / 2; k++)
            a += (f->samples[k] + f->samples[nsamp - 1 - k]) * qo_hb45_coef[k];
        x[nout++] = a * 2;
    }
    free(tmp);
    return nout;
}

/* ---------------------------------------------------------------- panadapter */
#include "fft_oracle.h"

struct qo_graph {
    int fft_size, data_width, index, count_fft;
    double rate, f_start, bandwidth, meter;
    double *samples;    /* one FFT buffer (fft_data.samples) */
    double *window, *avg;
};

qo_graph *qo_graph_create(int fft_size, int data_width, double fft_sample_rate)
{
    int i, j;
    qo_graph *g = (qo_graph *)calloc(1, sizeof(*g));
    g->fft_size = fft_size; g->data_width = data_width; g->rate = fft_sample_rate;
    g->samples = (double *)calloc((size_t)fft_size * 2, sizeof(double));
    g->window = (double *)malloc((size_t)fft_size * sizeof(double));
    g->avg = (double *)calloc((size_t)fft_size, sizeof(double));
    for (i = 0, j = -fft_size / 2; i < fft_size; i++, j++)          /* quisk.c:6003-6009, Hanning */
        g->window[i] = 0.5 + 0.5 * cos(2. * M_PI * j / fft_size);
    return g;
}

void qo_graph_free(qo_graph *g)
{
    if (!g) return;
    free(g->samples); free(g->window); free(g->avg); free(g);
}

void qo_graph_set_smeter_band(qo_graph *g, double f_start, double bandwidth)
{
    g->f_start = f_start; g->bandwidth = bandwidth;
}

static void graph_block(qo_graph *g)       /* quisk.c:5211-5278 */
{
    int i, j, k, n, N = g->fft_size;
    double d2, cr, ci;
    for (i = 0; i < N; i++) { g->samples[2 * i] *= g->window[i]; g->samples[2 * i + 1] *= g->window[i]; }
    fo_fft(g->samples, N, -1);
    d2 = g->bandwidth * N / g->rate;
    i = (int)(g->f_start * N / g->rate + 0.5);
    n = (int)(floor(d2) + 0.01);
    if (i > -N / 2 && i + n + 1 < N / 2) {
        for (j = 0; j < n; i++, j++) {
            k = i < 0 ? N + i : i;
            cr = g->samples[2 * k]; ci = g->samples[2 * k + 1];
            g->meter = g->meter + (cr * cr + ci * ci);
        }
        k = i < 0 ? N + i : i;
        cr = g->samples[2 * k]; ci = g->samples[2 * k + 1];
        g->meter = g->meter + (cr * cr + ci * ci) * (d2 - n);
    }
    g->count_fft++;
    k = 0;
    for (i = N / 2; i < N; i++) g->avg[k++] += hypot(g->samples[2 * i], g->samples[2 * i + 1]);   /* cabs */
    for (i = 0; i < N / 2; i++) g->avg[k++] += hypot(g->samples[2 * i], g->samples[2 * i + 1]);
}

int qo_graph_feed(qo_graph *g, const double *x, int n)      /* quisk.c:2454-2475 */
{
    int i, done = 0;
    for (i = 0; i < n; i++) {
        g->samples[2 * g->index] = x[2 * i];
        g->samples[2 * g->index + 1] = x[2 * i + 1];
        if (++g->index >= g->fft_size) {
            graph_block(g);
            g->index = 0;
            done++;
        }
    }
    return done;
}

int qo_graph_get(qo_graph *g, double zoom, double deltaf, double *pixels, double *smeter_db)   /* quisk.c:5279-5327 */
{
    int i, j, k, n, N = g->fft_size, count = g->count_fft;
    double d2, scale, smeter_scale, Smeter;
    if (count <= 0) return 0;
    scale = log10(count) + log10(N) + 31.0 * log10(2.0);
    scale *= 20.0;
    n = (int)(zoom * (double)N / g->data_width + 0.5);
    if (n < 1) n = 1;
    for (i = 0; i < g->data_width; i++) {       /* in place on fft_avg, like the reference */
        k = (int)(N * (deltaf / g->rate + zoom * ((double)i / g->data_width - 0.5) + 0.5) + 0.1);
        d2 = 0.0;
        for (j = 0; j < n; j++, k++)
            if (k >= 0 && k < N) d2 += g->avg[k];
        g->avg[i] = d2;
    }
    smeter_scale = 1.0 / 2147483647.0 / N;
    Smeter = g->meter * smeter_scale * smeter_scale / count;
    g->meter = 0;
    if (Smeter > 1E-16) Smeter = 10.0 * log10(Smeter);
    else Smeter = -160.0;
    Smeter += 4.25969;
    if (smeter_db) *smeter_db = Smeter;
    for (i = 0; i < g->data_width; i++) {
        d2 = 20.0 * log10(g->avg[i]) - scale;
        if (d2 < -200) d2 = -200;
        else if (d2 > 0) d2 = 0;
        pixels[i] = d2;
    }
    for (i = 0; i < N; i++) g->avg[i] = 0;
    g->count_fft = 0;
    return count;
}

void qo_watfall_row(const double *db, int size, int width, const unsigned char *red, const unsigned char *green,
                    const unsigned char *blue, int y_zero, int y_scale, double gain, unsigned char *rgb)     /* quisk.c:5372-5421 */
{
    int i, l;
    double yz = 40.0 + y_zero * 0.69;       /* -yz is the color center in dB */
    if (size > width) size = width;
    for (i = 0; i < size; i++) {
        l = (int)((db[i] - gain + yz) * (y_scale + 10) * 0.10 + 128);
        if (l < 0) l = 0; else if (l > 255) l = 255;
        *rgb++ = red[l]; *rgb++ = green[l]; *rgb++ = blue[l];
    }
    for (; i < width; i++) { *rgb++ = 0; *rgb++ = 0; *rgb++ = 0; }
}

/* ---- bandscope: copy2pixels quisk.c:4932-4955, init_bandscope :2876-2893, get_bandscope :4957-5011 ---- */
void qo_copy2pixels(double *pixels, int n_pixels, const double *fft, int fft_size, double zoom, double deltaf, double rate)
{
    int i, j, j1, j2;
    double f1, d1, d2, sample;
    f1 = deltaf + rate / 2.0 * (1.0 - zoom);        /* frequency at left of graph */
    for (i = 0; i < n_pixels; i++) {
        d1 = fft_size / rate * (f1 + (double)i / n_pixels * zoom * rate);
        d2 = fft_size / rate * (f1 + (double)(i + 1) / n_pixels * zoom * rate);
        j1 = (int)floor(d1);
        j2 = (int)floor(d2);
        if (j1 == j2) {
            sample = (d2 - d1) * fft[j1];
        } else {
            sample = (j1 + 1 - d1) * fft[j1];
            for (j = j1 + 1; j < j2; j++) sample += fft[j];
            sample += (d2 - j2) * fft[j2];
        }
        pixels[i] = sample;
    }
}

struct qo_bscope {
    int size, graph_width, fft_count;
    double the_max, *window, *average, *buf;
};

qo_bscope *qo_bscope_create(int size, int graph_width)
{
    int i, j;
    qo_bscope *b = (qo_bscope *)calloc(1, sizeof(*b));
    b->size = size; b->graph_width = graph_width;
    b->window = (double *)malloc((size_t)size * sizeof(double));
    b->average = (double *)calloc((size_t)size / 2 + 2, sizeof(double));
    b->buf = (double *)malloc((size_t)size * 2 * sizeof(double));
    for (i = 0, j = -size / 2; i < size; i++, j++) b->window[i] = 0.5 + 0.5 * cos(2. * M_PI * j / size);     /* Hanning */
    return b;
}

void qo_bscope_free(qo_bscope *b) { if (b) { free(b->window); free(b->average); free(b->buf); free(b); } }

void qo_bscope_block(qo_bscope *b, const double *samples)       /* bandscopeState == 99, quisk.c:4970-4983 */
{
    int i, L = b->size / 2 + 1;
    for (i = 0; i < b->size; i++) {
        double d1 = fabs(samples[i]);
        if (d1 > b->the_max) b->the_max = d1;
        b->buf[2 * i] = samples[i] * b->window[i];
        b->buf[2 * i + 1] = 0.0;
    }
    fo_fft(b->buf, b->size, -1);                                 /* r2c: bins 0 .. size / 2 */
    for (i = 0; i < L; i++) b->average[i] += hypot(b->buf[2 * i], b->buf[2 * i + 1]);
    b->fft_count++;
}

int qo_bscope_get(qo_bscope *b, int clock, double zoom, double deltaf, double *pixels, double *adc_level)   /* quisk.c:4984-5006 */
{
    int i, L = b->size / 2 + 1, n = b->fft_count;
    double frac, scale, rate, sample;
    if (n <= 0) return 0;
    b->average[L] = 0.0;                                         /* in case we run off the end */
    frac = (double)L / b->graph_width;
    scale = 1.0 / frac / b->fft_count / b->size;
    rate = clock / 2.0;
    qo_copy2pixels(pixels, b->graph_width, b->average, L, zoom, deltaf, rate);
    for (i = 0; i < b->graph_width; i++) {
        sample = pixels[i] * scale;
        pixels[i] = sample <= 1E-10 ? -200.0 : 20.0 * log10(sample);
    }
    b->fft_count = 0;
    if (adc_level) *adc_level = b->the_max;                      /* hermes_adc_level */
    b->the_max = 0;
    for (i = 0; i < L; i++) b->average[i] = 0;
    return n;
}
