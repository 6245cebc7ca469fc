/* wdsp_oracle.c -- TEST INFRASTRUCTURE ONLY.  See wdsp_oracle.h for scope and the
 * "parity unpinned" statement.  Every block cites the reference lines it restates
 * (paths relative to /root/reference).  State that the reference keeps in per-channel
 * globals (rxa[], ch[]) lives in one heap struct so that many channels can run
 * side by side and on several host threads.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fft_oracle.h"
#include "wdsp_oracle.h"
#include "emnr_oracle.h"
#include "snba_oracle.h"
#include "wcpagc_oracle.h"

#define WO_PI    3.1415926535897932   /* wdsp/comm.h:146 */
#define WO_TWOPI 6.2831853071795864   /* wdsp/comm.h:147 */
#define WO_DSP_MULT 2                 /* wdsp/comm.h:118 */

static void *zalloc(size_t n) { void *p = calloc(1, n ? n : 1); return p; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------ fir_bandpass */
/* wdsp/fir.c:187-254.  rtype 0: N real taps; rtype 1: N complex taps (re, -im*sin). */
double *wo_fir_bandpass(int N, double f_low, double f_high, double samplerate, int wintype, int rtype, double scale)
{
    double *c_impulse = (double *)zalloc((size_t)N * 2 * sizeof(double));
    double ft = (f_high - f_low) / (2.0 * samplerate);
    double ft_rad = WO_TWOPI * ft;
    double w_osc = WO_PI * (f_high + f_low) / samplerate;
    int i, j;
    double m = 0.5 * (double)(N - 1);
    double delta = WO_PI / m;
    double cosphi, posi, posj, sinc, window = 0.0, coef;

    if (N & 1) {
        switch (rtype) {
        case 0: c_impulse[N >> 1] = scale * 2.0 * ft; break;
        case 1: c_impulse[N - 1] = scale * 2.0 * ft; c_impulse[N] = 0.0; break;
        }
    }
    for (i = (N + 1) / 2, j = N / 2 - 1; i < N; i++, j--) {
        posi = (double)i - m;
        posj = (double)j - m;
        sinc = sin(ft_rad * posi) / (WO_PI * posi);
        cosphi = cos(delta * i);
        switch (wintype) {
        case 0: /* Blackman-Harris 4-term, fir.c:220-225 */
            window = +0.21747 + cosphi * (-0.45325 + cosphi * (+0.28256 + cosphi * (-0.04672)));
            break;
        case 1: /* Blackman-Harris 7-term, fir.c:227-236 */
            window = +6.3964424114390378e-02
                   + cosphi * (-2.3993864599352804e-01
                   + cosphi * (+3.5015956323820469e-01
                   + cosphi * (-2.4774111897080783e-01
                   + cosphi * (+8.5438256055858031e-02
                   + cosphi * (-1.2320203369293225e-02
                   + cosphi * (+4.3778825791773474e-04))))));
            break;
        }
        coef = scale * sinc * window;
        switch (rtype) {
        case 0:
            c_impulse[i] = +coef * cos(posi * w_osc);
            c_impulse[j] = +coef * cos(posj * w_osc);
            break;
        case 1:
            c_impulse[2 * i + 0] = +coef * cos(posi * w_osc);
            c_impulse[2 * i + 1] = -coef * sin(posi * w_osc);
            c_impulse[2 * j + 0] = +coef * cos(posj * w_osc);
            c_impulse[2 * j + 1] = -coef * sin(posj * w_osc);
            break;
        }
    }
    return c_impulse;
}

/* get_fsamp_window(), wdsp/fir.c:44-81 */
static double *wo_fsamp_window(int N, int wintype)
{
    int i;
    double arg0, arg1;
    double *window = (double *)zalloc((size_t)N * sizeof(double));
    switch (wintype) {
    case 0:
        arg0 = 2.0 * WO_PI / ((double)N - 1.0);
        for (i = 0; i < N; i++) {
            arg1 = cos(arg0 * (double)i);
            window[i] = +0.21747 + arg1 * (-0.45325 + arg1 * (+0.28256 + arg1 * (-0.04672)));
        }
        break;
    case 1:
        arg0 = 2.0 * WO_PI / ((double)N - 1.0);
        for (i = 0; i < N; ++i) {
            arg1 = cos(arg0 * (double)i);
            window[i] = +6.3964424114390378e-02
                      + arg1 * (-2.3993864599352804e-01
                      + arg1 * (+3.5015956323820469e-01
                      + arg1 * (-2.4774111897080783e-01
                      + arg1 * (+8.5438256055858031e-02
                      + arg1 * (-1.2320203369293225e-02
                      + arg1 * (+4.3778825791773474e-04))))));
        }
        break;
    default:
        for (i = 0; i < N; i++) window[i] = 1.0;
    }
    return window;
}

/* fir_fsamp(), wdsp/fir.c:129-185 (frequency-sampling design, rtype 1 only is used on this path) */
static double *wo_fir_fsamp(int N, const double *A, int rtype, double scale, int wintype)
{
    int n, i, j, k;
    double sum;
    double *window;
    double *c_impulse = (double *)zalloc((size_t)N * 2 * sizeof(double));
    if (N & 1) {
        int M = (N - 1) / 2;
        for (n = 0; n < M + 1; n++) {
            sum = 0.0;
            for (k = 1; k < M + 1; k++)
                sum += 2.0 * A[k] * cos(WO_TWOPI * (n - M) * k / N);
            c_impulse[2 * n + 0] = (1.0 / N) * (A[0] + sum);
            c_impulse[2 * n + 1] = 0.0;
        }
        for (n = M + 1, j = 1; n < N; n++, j++) {
            c_impulse[2 * n + 0] = c_impulse[2 * (M - j) + 0];
            c_impulse[2 * n + 1] = 0.0;
        }
    } else {
        double M = (double)(N - 1) / 2.0;
        for (n = 0; n < N / 2; n++) {
            sum = 0.0;
            for (k = 1; k < N / 2; k++)
                sum += 2.0 * A[k] * cos(WO_TWOPI * (n - M) * k / N);
            c_impulse[2 * n + 0] = (1.0 / N) * (A[0] + sum);
            c_impulse[2 * n + 1] = 0.0;
        }
        for (n = N / 2, j = 1; n < N; n++, j++) {
            c_impulse[2 * n + 0] = c_impulse[2 * (N / 2 - j) + 0];
            c_impulse[2 * n + 1] = 0.0;
        }
    }
    window = wo_fsamp_window(N, wintype);
    switch (rtype) {
    case 0:
        for (i = 0; i < N; i++) c_impulse[i] = scale * c_impulse[2 * i] * window[i];
        break;
    case 1:
        for (i = 0; i < N; i++) {
            c_impulse[2 * i + 0] *= scale * window[i];
            c_impulse[2 * i + 1] = 0.0;
        }
        break;
    }
    free(window);
    return c_impulse;
}

/* fc_impulse(), wdsp/fcurve.c:29-145.  Only the even-nc branch reaches fir_fsamp();
 * odd nc would need fir_fsamp_odd() (an N-point, N odd, backward FFT: fir.c:83-127),
 * which RXA never requests (nc is a power of two, RXA.c:209,211). */
static double *wo_fc_impulse(int nc, double f0, double f1, double g0, double g1, int curve,
                             double samplerate, double scale, int ctfmode, int wintype)
{
    double *A = (double *)zalloc((size_t)(nc / 2 + 1) * sizeof(double));
    int i;
    double fn, f;
    double *impulse;
    int mid = nc / 2;
    double g0_lin = pow(10.0, g0 / 20.0);
    (void)g1;
    if (nc & 1) { free(A); return NULL; }
    for (i = 0; i < mid; i++) {
        fn = ((double)i + 0.5) / (double)mid;
        f = fn * samplerate / 2.0;
        switch (curve) {
        case 0: A[i] = (f0 > 0.0) ? scale * (g0_lin * f / f0) : 0.0; break;
        case 1: A[i] = (f > 0.0) ? scale * (g0_lin * f0 / f) : 0.0; break;
        }
    }
    if (ctfmode == 0) {
        int k, low, high;
        double lowmag, highmag, flow4, fhigh4;
        low = (int)(2.0 * f0 / samplerate * mid - 0.5);
        high = (int)(2.0 * f1 / samplerate * mid - 0.5);
        lowmag = A[low];
        highmag = A[high];
        flow4 = pow((double)low / (double)mid, 4.0);
        fhigh4 = pow((double)high / (double)mid, 4.0);
        k = low;
        while (--k >= 0) {
            f = (double)k / (double)mid;
            lowmag *= (f * f * f * f) / flow4;
            if (lowmag < 1.0e-100) lowmag = 1.0e-100;
            A[k] = lowmag;
        }
        k = high;
        while (++k < mid) {
            f = (double)k / (double)mid;
            highmag *= fhigh4 / (f * f * f * f);
            if (highmag < 1.0e-100) highmag = 1.0e-100;
            A[k] = highmag;
        }
    }
    impulse = wo_fir_fsamp(nc, A, 1, 1.0, wintype);
    free(A);
    return impulse;
}

/* ------------------------------------------------------------------ fircore */
/* wdsp/firmin.h:138-161, firmin.c:290-430 */
struct wo_fircore {
    int size, nc, nfor, buffidx, idxmask, mp;
    double *impulse;  /* nc complex: what the caller gave (a->impulse) */
    double *imp;      /* nc complex: what the masks are made of (a->imp) */
    double *fftin;    /* 2*size complex */
    double **fftout;  /* nfor x 2*size complex */
    double **fmask;   /* nfor x 2*size complex (single mask set: setters here run between blocks) */
    double *accum;    /* 2*size complex */
    double *maskgen;
};

static void fircore_plan(wo_fircore *a)        /* plan_fircore, firmin.c:290-320 */
{
    int i;
    a->nfor = a->nc / a->size;
    a->buffidx = 0;
    a->idxmask = a->nfor - 1;
    a->fftin = (double *)zalloc((size_t)2 * a->size * 2 * sizeof(double));
    a->fftout = (double **)zalloc((size_t)a->nfor * sizeof(double *));
    a->fmask = (double **)zalloc((size_t)a->nfor * sizeof(double *));
    a->maskgen = (double *)zalloc((size_t)2 * a->size * 2 * sizeof(double));
    for (i = 0; i < a->nfor; i++) {
        a->fftout[i] = (double *)zalloc((size_t)2 * a->size * 2 * sizeof(double));
        a->fmask[i] = (double *)zalloc((size_t)2 * a->size * 2 * sizeof(double));
    }
    a->accum = (double *)zalloc((size_t)2 * a->size * 2 * sizeof(double));
}

static void fircore_deplan(wo_fircore *a)      /* deplan_fircore, firmin.c:364-388 */
{
    int i;
    for (i = 0; i < a->nfor; i++) { free(a->fftout[i]); free(a->fmask[i]); }
    free(a->fftout); free(a->fmask); free(a->fftin); free(a->maskgen); free(a->accum);
}

/* analytic, wdsp/fir.c:292-317 (in place) */
static void wo_analytic(int N, double *x)
{
    int i;
    double inv_N = 1.0 / (double)N, two_inv_N = 2.0 * inv_N;
    fo_fft(x, N, -1);
    x[0] *= inv_N; x[1] *= inv_N;
    for (i = 1; i < N / 2; i++) { x[2 * i] *= two_inv_N; x[2 * i + 1] *= two_inv_N; }
    x[N] *= inv_N; x[N + 1] *= inv_N;
    memset(&x[N + 2], 0, (size_t)(N - 2) * sizeof(double));
    fo_fft(x, N, +1);
}

/* mp_imp, wdsp/fir.c:319-368 */
void wo_mp_imp(int N, const double *fir, double *mpfir, int pfactor, int polarity)
{
    int i, size = N * pfactor;
    double inv_PN = 1.0 / (double)size;
    double *firfreq = (double *)zalloc((size_t)size * 2 * sizeof(double));
    double *mag = (double *)zalloc((size_t)size * sizeof(double));
    double *ana = (double *)zalloc((size_t)size * 2 * sizeof(double));
    double *newfreq = (double *)zalloc((size_t)size * 2 * sizeof(double));
    memcpy(firfreq, fir, (size_t)N * 2 * sizeof(double));
    fo_fft(firfreq, size, -1);
    for (i = 0; i < size; i++) {
        mag[i] = sqrt(firfreq[2 * i] * firfreq[2 * i] + firfreq[2 * i + 1] * firfreq[2 * i + 1]) * inv_PN;
        ana[2 * i] = mag[i] > 0.0 ? log(mag[i]) : log(1.0e-300);
    }
    wo_analytic(size, ana);
    for (i = 0; i < size; i++) {
        newfreq[2 * i] = mag[i] * cos(ana[2 * i + 1]);
        newfreq[2 * i + 1] = (polarity ? 1.0 : -1.0) * mag[i] * sin(ana[2 * i + 1]);
    }
    fo_fft(newfreq, size, +1);
    memcpy(mpfir, polarity ? &newfreq[2 * (pfactor - 1) * N] : newfreq, (size_t)N * 2 * sizeof(double));
    free(firfreq); free(mag); free(ana); free(newfreq);
}

static void fircore_calc(wo_fircore *a)        /* calc_fircore, firmin.c:322-346 */
{
    int i;
    if (a->mp) wo_mp_imp(a->nc, a->impulse, a->imp, 16, 0);
    else memcpy(a->imp, a->impulse, (size_t)a->nc * 2 * sizeof(double));
    for (i = 0; i < a->nfor; i++) {
        /* impulse partition right-justified in the 2*size buffer (firmin.c:335) */
        memset(a->maskgen, 0, (size_t)2 * a->size * 2 * sizeof(double));
        memcpy(&a->maskgen[2 * a->size], &a->imp[2 * a->size * i], (size_t)a->size * 2 * sizeof(double));
        fo_fft_oop(a->maskgen, a->fmask[i], 2 * a->size, -1);
    }
}

wo_fircore *wo_fircore_create(int size, int nc, const double *impulse)  /* create_fircore, firmin.c:348-362 */
{
    wo_fircore *a = (wo_fircore *)zalloc(sizeof(*a));
    a->size = size;
    a->nc = nc;
    fircore_plan(a);
    a->imp = (double *)zalloc((size_t)nc * 2 * sizeof(double));
    a->impulse = (double *)zalloc((size_t)nc * 2 * sizeof(double));
    memcpy(a->impulse, impulse, (size_t)nc * 2 * sizeof(double));
    fircore_calc(a);
    return a;
}

void wo_fircore_destroy(wo_fircore *a)
{
    if (!a) return;
    fircore_deplan(a);
    free(a->imp); free(a->impulse);
    free(a);
}

static void fircore_flush(wo_fircore *a)        /* flush_fircore, firmin.c:399-407 */
{
    int i;
    memset(a->fftin, 0, (size_t)2 * a->size * 2 * sizeof(double));
    for (i = 0; i < a->nfor; i++) memset(a->fftout[i], 0, (size_t)2 * a->size * 2 * sizeof(double));
    a->buffidx = 0;
}

static void fircore_set_impulse(wo_fircore *a, const double *impulse)  /* setImpulse_fircore, firmin.c:448-452 */
{
    memcpy(a->impulse, impulse, (size_t)a->nc * 2 * sizeof(double));
    fircore_calc(a);
}

static void fircore_set_mp(wo_fircore *a, int mp)      /* setMp_fircore, firmin.c:469-473 */
{
    a->mp = mp;
    fircore_calc(a);
}

static void fircore_set_nc(wo_fircore *a, int nc, const double *impulse) /* setNc_fircore, firmin.c:454-466 */
{
    fircore_deplan(a);
    free(a->imp); free(a->impulse);
    a->nc = nc;
    fircore_plan(a);
    a->imp = (double *)zalloc((size_t)nc * 2 * sizeof(double));
    a->impulse = (double *)zalloc((size_t)nc * 2 * sizeof(double));
    memcpy(a->impulse, impulse, (size_t)nc * 2 * sizeof(double));
    fircore_calc(a);
}

void wo_fircore_exec(wo_fircore *a, const double *in, double *out)   /* xfircore, firmin.c:409-430 */
{
    int i, j, k;
    int n2 = 2 * a->size;
    memcpy(&a->fftin[2 * a->size], in, (size_t)a->size * 2 * sizeof(double));
    fo_fft_oop(a->fftin, a->fftout[a->buffidx], n2, -1);
    k = a->buffidx;
    memset(a->accum, 0, (size_t)n2 * 2 * sizeof(double));
    for (j = 0; j < a->nfor; j++) {
        const double *fo = a->fftout[k];
        const double *fm = a->fmask[j];
        for (i = 0; i < n2; i++) {
            a->accum[2 * i + 0] += fo[2 * i + 0] * fm[2 * i + 0] - fo[2 * i + 1] * fm[2 * i + 1];
            a->accum[2 * i + 1] += fo[2 * i + 0] * fm[2 * i + 1] + fo[2 * i + 1] * fm[2 * i + 0];
        }
        k = (k + a->idxmask) & a->idxmask;
    }
    a->buffidx = (a->buffidx + 1) & a->idxmask;
    fo_fft(a->accum, n2, +1);                       /* crev: backward, unnormalised */
    memcpy(out, a->accum, (size_t)a->size * 2 * sizeof(double));   /* left half is the result */
    memcpy(a->fftin, &a->fftin[2 * a->size], (size_t)a->size * 2 * sizeof(double));
}

/* ------------------------------------------------------------------ resample */
/* wdsp/resample.h, resample.c:35-157 */
struct wo_resample {
    int run, in_rate, out_rate, L, M, ncoef, cpp, ringsize, idx_in, phnum;
    double *h, *ring;
};

double *wo_calc_resample_taps(int in_rate, int out_rate, double fc, int ncoef_in, double gain,
                              int *pL, int *pM, int *pncoef, int *pcpp)   /* calc_resample, resample.c:35-72 */
{
    int x, y, z, i, j, k, min_rate, L, M, ncoef, cpp;
    double full_rate, fc_norm_high, fc_norm_low;
    double *impulse, *h;
    x = in_rate; y = out_rate;
    while (y != 0) { z = y; y = x % y; x = z; }
    L = out_rate / x;
    M = in_rate / x;
    min_rate = (in_rate < out_rate) ? in_rate : out_rate;
    if (fc == 0.0) fc = 0.45 * (double)min_rate;
    full_rate = (double)(in_rate * L);
    fc_norm_high = fc / full_rate;
    fc_norm_low = -fc_norm_high;                    /* fc_low = -1.0 < 0, resample.c:55-56,97 */
    ncoef = ncoef_in;
    if (ncoef == 0) ncoef = (int)(140.0 * full_rate / min_rate);
    ncoef = (ncoef / L + 1) * L;
    cpp = ncoef / L;
    h = (double *)zalloc((size_t)ncoef * sizeof(double));
    impulse = wo_fir_bandpass(ncoef, fc_norm_low, fc_norm_high, 1.0, 1, 0, gain * (double)L);
    i = 0;
    for (j = 0; j < L; j++)
        for (k = 0; k < ncoef; k += L)
            h[i++] = impulse[j + k];
    free(impulse);
    *pL = L; *pM = M; *pncoef = ncoef; *pcpp = cpp;
    return h;
}

wo_resample *wo_resample_create(int in_rate, int out_rate, double fc, int ncoef, double gain)
{
    wo_resample *a = (wo_resample *)zalloc(sizeof(*a));
    a->run = 1;
    a->in_rate = in_rate;
    a->out_rate = out_rate;
    a->h = wo_calc_resample_taps(in_rate, out_rate, fc, ncoef, gain, &a->L, &a->M, &a->ncoef, &a->cpp);
    a->ringsize = a->cpp;
    a->ring = (double *)zalloc((size_t)a->ringsize * 2 * sizeof(double));
    a->idx_in = a->ringsize - 1;
    a->phnum = 0;
    return a;
}

void wo_resample_destroy(wo_resample *a)
{
    if (!a) return;
    free(a->h); free(a->ring); free(a);
}

int wo_resample_exec(wo_resample *a, const double *in, int size, double *out)   /* xresample, resample.c:120-157 */
{
    int outsamps = 0;
    int i, j, n, idx_out;
    double I, Q;
    for (i = 0; i < size; i++) {
        a->ring[2 * a->idx_in + 0] = in[2 * i + 0];
        a->ring[2 * a->idx_in + 1] = in[2 * i + 1];
        while (a->phnum < a->L) {
            I = 0.0; Q = 0.0;
            n = a->cpp * a->phnum;
            for (j = 0; j < a->cpp; j++) {
                if ((idx_out = a->idx_in + j) >= a->ringsize) idx_out -= a->ringsize;
                I += a->h[n + j] * a->ring[2 * idx_out + 0];
                Q += a->h[n + j] * a->ring[2 * idx_out + 1];
            }
            out[2 * outsamps + 0] = I;
            out[2 * outsamps + 1] = Q;
            outsamps++;
            a->phnum += a->M;
        }
        a->phnum -= a->L;
        if (--a->idx_in < 0) a->idx_in = a->ringsize - 1;
    }
    return outsamps;
}

/* ------------------------------------------------------------------ meter */
typedef struct {           /* wdsp/meter.c:29-108 */
    double rate, mult_average, mult_peak, avg, peak;
    int enum_av, enum_pk, enum_gain;
} wo_meter;

/* mlog10, wdsp/meterlog10.c:29-32,547-554: log10(2) * (exponent + mtable[top 11 mantissa bits]) where
 * mtable[m] = log2(1 + m/2048) (the 2048-entry table is that function tabulated to 17 digits). */
static double wo_mlog10(double val)
{
    unsigned long long N;
    int e, m;
    memcpy(&N, &val, 8);
    e = (int)((N >> 52) & 2047) - 1023;
    m = (int)((N >> (52 - 11)) & 2047);
    return 0.301029995663981 * (e + log2(1.0 + m / 2048.0));
}

double wo_mlog10_value(double v) { return wo_mlog10(v); }      /* for analyzer_oracle.c */

static void meter_init(wo_meter *m, int rate, double tau_av, double tau_decay, int eav, int epk, int egain, double *result)
{
    m->rate = (double)rate;
    m->mult_average = exp(-1.0 / (m->rate * tau_av));
    m->mult_peak = exp(-1.0 / (m->rate * tau_decay));
    m->avg = 0.0; m->peak = 0.0;
    m->enum_av = eav; m->enum_pk = epk; m->enum_gain = egain;
    result[eav] = -400.0; result[epk] = -400.0;
    if (egain >= 0) result[egain] = -400.0;
}

static void meter_exec(wo_meter *m, const double *buff, int size, double *result, const double *pgain)
{
    int i;
    double smag, np = 0.0;
    for (i = 0; i < size; i++) {
        smag = buff[2 * i + 0] * buff[2 * i + 0] + buff[2 * i + 1] * buff[2 * i + 1];
        m->avg = m->avg * m->mult_average + (1.0 - m->mult_average) * smag;
        m->peak *= m->mult_peak;
        if (smag > np) np = smag;
    }
    if (np > m->peak) m->peak = np;
    result[m->enum_av] = 10.0 * wo_mlog10(m->avg + 1.0e-40);
    result[m->enum_pk] = 10.0 * wo_mlog10(m->peak + 1.0e-40);
    if (pgain && m->enum_gain >= 0) result[m->enum_gain] = 20.0 * wo_mlog10(*pgain + 1.0e-40);
}

/* meter indices, wdsp/RXA.h:47-57 */
enum { WO_S_PK = 0, WO_S_AV, WO_ADC_PK, WO_ADC_AV, WO_AGC_GAIN, WO_AGC_PK, WO_AGC_AV, WO_METERTYPE_LAST };

/* ------------------------------------------------------------------ channel */
#define AMD_STAGES 7
#define AMD_OUT_IDX (3 * AMD_STAGES)

struct wo_channel {
    /* ch[] fields, wdsp/channel.h */
    int in_size, dsp_size, in_rate, dsp_rate, out_rate;
    int dsp_insize, dsp_outsize, out_size;
    double tdelayup, tslewup, tdelaydown, tslewdown;
    int mode;
    double *inbuff, *midbuff, *outbuff;
    double meter[WO_METERTYPE_LAST];
    /* shift, wdsp/shift.h */
    struct { int run; double rate, shift, phase, delta, cos_delta, sin_delta; } shift;
    wo_resample *rsmpin, *rsmpout;
    wo_meter adcmeter, smeter, agcmeter;
    /* nbp0 (notches never run on this path: ndb master_run = 0, RXA.c:84-87) */
    struct { int run, nc, wintype; double flow, fhigh, gain, rate; wo_fircore *p;
             int fnfrun, autoincr; } nbp0;                         /* create_nbp, nbp.c:241-268 */
    struct { int master_run, nn; double tunefreq, shift;           /* notchdb, nbp.h */
             double fcenter[1024], fwidth[1024], nlow[1024], nhigh[1024]; int active[1024]; } ndb;
    /* amd, wdsp/amd.h */
    struct {
        int run, mode, levelfade, sbmode;
        double sample_rate, omega_min, omega_max, g1, g2, phs, fil_out, omega;
        double dc, dc_insert, mtauR, onem_mtauR, mtauI, onem_mtauI;
        double a[3 * AMD_STAGES + 3], b[3 * AMD_STAGES + 3], c[3 * AMD_STAGES + 3], d[3 * AMD_STAGES + 3];
        double c0[AMD_STAGES], c1[AMD_STAGES], dsI, dsQ;
    } amd;
    /* fmd, wdsp/fmd.h */
    struct {
        int run, nc_de, nc_aud, sntch_run;
        double rate, deviation, f_low, f_high, fmin, fmax, zeta, omegaN, tau, afgain, ctcss_freq;
        double omega_min, omega_max, g1, g2, phs, fil_out, omega, mtau, onem_mtau, fmdc, again;
        double *audio;
        wo_fircore *pde, *paud;
        int lim_run; double lim_pre_gain, lim_gain; wo_agc plim;      /* detector limiter, fmd.c:48-72,106-108 */
        /* snotch, wdsp/iir.c:35-95 */
        double a0, a1, a2, b1, b2, x1, x2, y1, y2;
    } fmd;
    wo_agc agc;
    struct { int run, nc, wintype, position; double f_low, f_high, gain; wo_fircore *p; } bp1;
    wo_emnr *emnr;                              /* wdsp/emnr.c, restated in emnr_oracle.c */
    int emnr_pending_run;                       /* the emnr_run argument of RXAbp1Check (RXA.c:800) */
    wo_snba *snba;                              /* wdsp/snb.c, restated in snba_oracle.c */
    int snba_pending_run;                       /* the snba_run argument of RXAbp1Check */
    /* bpsnba, snb.c:696-855: a notched bandpass fed ahead of nbp0 (position 0) or behind the detectors (position 1) */
    struct { int run, run_notches, position, nc, wintype, autoincr; double abs_low, abs_high, f_low, f_high; double *buff; wo_fircore *p; } bpsnba;
    /* amsq, wdsp/amsq.h */
    struct {
        int run, state, count, ntup, ntdown;
        double avm, onem_avm, avsig, tail_thresh, unmute_thresh, min_tail, max_tail, muted_gain, rate;
        double *cup, *cdown, *trigsig;
    } amsq;
    /* anf / anr, wdsp/anf.h:32-61, anr.h:32-61 (the two structs are the same) */
    struct wo_lms {
        int run, position, n_taps, delay, in_idx, mask;
        double two_mu, gamma, lidx, lidx_min, lidx_max, ngamma, den_mult, lincr, ldecr;
        double d[2048], w[2048];                /* ANF_DLINE_SIZE, anf.h:30 */
    } anf, anr;
    struct { int run, inselect, copy; double gain1, gain2I, gain2Q; } panel;
    /* iobuffs, wdsp/iobuffs.h */
    struct {
        int r1_outsize, r1_size, r2_insize, r2_size, r1_active, r2_active;
        double *r1, *r2;
        int r1_inidx, r1_outidx, r1_unqueued, r2_inidx, r2_outidx, r2_havesamps, r2_unqueued;
        int sem_buffready, sem_outready;
        int ustate, ucount, ndelup, ntup, upflag;
        double *cup;
    } iob;
};

int wo_dsp_insize(const wo_channel *c) { return c->dsp_insize; }
int wo_dsp_outsize(const wo_channel *c) { return c->dsp_outsize; }
int wo_out_size(const wo_channel *c) { return c->out_size; }

/* ---- shift (wdsp/shift.c:29-85) */
static void calc_shift(wo_channel *c)
{
    c->shift.delta = WO_TWOPI * c->shift.shift / c->shift.rate;
    c->shift.cos_delta = cos(c->shift.delta);
    c->shift.sin_delta = sin(c->shift.delta);
}

static void xshift(wo_channel *c, double *buf, int size)
{
    if (c->shift.run) {
        int i;
        double I1, Q1, t1, t2;
        double cos_phase = cos(c->shift.phase);
        double sin_phase = sin(c->shift.phase);
        for (i = 0; i < size; i++) {
            I1 = buf[2 * i + 0];
            Q1 = buf[2 * i + 1];
            buf[2 * i + 0] = I1 * cos_phase - Q1 * sin_phase;
            buf[2 * i + 1] = I1 * sin_phase + Q1 * cos_phase;
            t1 = cos_phase; t2 = sin_phase;
            cos_phase = t1 * c->shift.cos_delta - t2 * c->shift.sin_delta;
            sin_phase = t1 * c->shift.sin_delta + t2 * c->shift.cos_delta;
            c->shift.phase += c->shift.delta;
            if (c->shift.phase >= WO_TWOPI) c->shift.phase -= WO_TWOPI;
            if (c->shift.phase < 0.0) c->shift.phase += WO_TWOPI;
        }
    }
}

/* ---- nbp0 impulse without notches (calc_nbp_impulse else-branch, wdsp/nbp.c:234-238) */
/* make_nbp, wdsp/nbp.c:97-179: the passband [flow, fhigh] with the active notches cut out, as a list of bands */
int wo_make_nbp(int nn, const int *active, const double *center, const double *width, const double *nlow, const double *nhigh,
                double minwidth, int autoincr, double flow, double fhigh, double *bplow, double *bphigh, int *havnotch)
{
    int nbp, nnbp, adds, i, j, k;
    double nl, nh;
    int *del = (int *)zalloc(1024 * sizeof(int));
    if (fhigh > flow) { bplow[0] = flow; bphigh[0] = fhigh; nbp = 1; }
    else { free(del); return 0; }
    *havnotch = 0;
    for (k = 0; k < nn; k++) {
        if (autoincr && width[k] < minwidth) { nl = center[k] - 0.5 * minwidth; nh = center[k] + 0.5 * minwidth; }
        else { nl = nlow[k]; nh = nhigh[k]; }
        if (active[k] && (nh > flow && nl < fhigh)) {
            *havnotch = 1;
            adds = 0;
            for (i = 0; i < nbp; i++) {
                if (nh > bplow[i] && nl < bphigh[i]) {
                    if (nl <= bplow[i] && nh >= bphigh[i]) del[i] = 1;
                    else if (nl > bplow[i] && nh < bphigh[i]) {
                        bplow[nbp + adds] = nh; bphigh[nbp + adds] = bphigh[i]; bphigh[i] = nl; adds++;
                    }
                    else if (nl <= bplow[i] && nh > bplow[i]) bplow[i] = nh;
                    else if (nl < bphigh[i] && nh >= bphigh[i]) bphigh[i] = nl;
                }
            }
            nbp += adds;
            nnbp = nbp;
            for (i = 0; i < nbp; i++) {
                if (del[i] == 1) {
                    nnbp--;
                    for (j = i; j < nnbp; j++) { bplow[j] = bplow[j + 1]; bphigh[j] = bphigh[j + 1]; }
                    del[i] = 0;
                }
            }
            nbp = nnbp;
        }
    }
    free(del);
    return nbp;
}

static double nbp0_min_notch_width(wo_channel *c)      /* min_notch_width, nbp.c:82-95 */
{
    return (c->nbp0.wintype == 1 ? 2200.0 : 1600.0) / (c->nbp0.nc / 256) * (c->nbp0.rate / 48000);
}

static double *bpsnba_impulse(wo_channel *c)           /* calc_nbp_impulse for the nbp inside bpsnba (wintype 0, gain 1, autoincr 1: RXA.c:109-127) */
{
    double scale = 1.0 / (double)(2 * c->dsp_size);
    double rate = (double)c->dsp_rate;
    if (c->bpsnba.run_notches) {
        double *bplow = (double *)zalloc(1025 * sizeof(double)), *bphigh = (double *)zalloc(1025 * sizeof(double));
        double offset = c->ndb.tunefreq + c->ndb.shift;
        int havnotch = 0, i, k, numpb;
        double *impulse = (double *)zalloc((size_t)c->bpsnba.nc * 2 * sizeof(double));
        numpb = wo_make_nbp(c->ndb.nn, c->ndb.active, c->ndb.fcenter, c->ndb.fwidth, c->ndb.nlow, c->ndb.nhigh,
                            (c->bpsnba.wintype == 1 ? 2200.0 : 1600.0) / (c->bpsnba.nc / 256) * (rate / 48000), c->bpsnba.autoincr, c->bpsnba.f_low + offset, c->bpsnba.f_high + offset,
                            bplow, bphigh, &havnotch);
        for (k = 0; k < numpb; k++) {
            double *imp = wo_fir_bandpass(c->bpsnba.nc, bplow[k] - offset, bphigh[k] - offset, rate, c->bpsnba.wintype, 1, scale);
            for (i = 0; i < 2 * c->bpsnba.nc; i++) impulse[i] += imp[i];
            free(imp);
        }
        free(bplow); free(bphigh);
        return impulse;
    }
    return wo_fir_bandpass(c->bpsnba.nc, c->bpsnba.f_low, c->bpsnba.f_high, rate, c->bpsnba.wintype, 1, scale);
}

static double *nbp0_impulse(wo_channel *c)             /* calc_nbp_impulse, nbp.c:214-239 */
{
    double scale = c->nbp0.gain / (double)(2 * c->dsp_size);
    if (c->nbp0.fnfrun) {
        double *bplow = (double *)zalloc(1025 * sizeof(double)), *bphigh = (double *)zalloc(1025 * sizeof(double));
        double offset = c->ndb.tunefreq + c->ndb.shift;
        int havnotch = 0, i, k, numpb;
        double *impulse = (double *)zalloc((size_t)c->nbp0.nc * 2 * sizeof(double));
        numpb = wo_make_nbp(c->ndb.nn, c->ndb.active, c->ndb.fcenter, c->ndb.fwidth, c->ndb.nlow, c->ndb.nhigh,
                            nbp0_min_notch_width(c), c->nbp0.autoincr, c->nbp0.flow + offset, c->nbp0.fhigh + offset,
                            bplow, bphigh, &havnotch);
        for (k = 0; k < numpb; k++) {                   /* fir_mbandpass, nbp.c:64-80 */
            double *imp = wo_fir_bandpass(c->nbp0.nc, bplow[k] - offset, bphigh[k] - offset, c->nbp0.rate, c->nbp0.wintype, 1, scale);
            for (i = 0; i < 2 * c->nbp0.nc; i++) impulse[i] += imp[i];
            free(imp);
        }
        free(bplow); free(bphigh);
        return impulse;
    }
    return wo_fir_bandpass(c->nbp0.nc, c->nbp0.flow, c->nbp0.fhigh, c->nbp0.rate, c->nbp0.wintype, 1, scale);
}

static void nbp0_update(wo_channel *c)                 /* UpdateNBPFilters, nbp.c:342-356 (nbp0 part) */
{
    double *imp = nbp0_impulse(c);
    fircore_set_impulse(c->nbp0.p, imp);
    free(imp);
}

static void bpsnba_update(wo_channel *c)               /* recalc_bpsnba_filter, snb.c:807-822 */
{
    double *imp = bpsnba_impulse(c);
    fircore_set_impulse(c->bpsnba.p, imp);
    free(imp);
}

static void ndb_update(wo_channel *c)                  /* UpdateNBPFilters, nbp.c:342-356 */
{
    if (c->nbp0.fnfrun) nbp0_update(c);
    if (c->bpsnba.run_notches) bpsnba_update(c);
}

static double *bp1_impulse(wo_channel *c)      /* bandpass.c:302 */
{
    return wo_fir_bandpass(c->bp1.nc, c->bp1.f_low, c->bp1.f_high, (double)c->dsp_rate, c->bp1.wintype, 1,
                           c->bp1.gain / (double)(2 * c->dsp_size));
}

/* ---- amd (wdsp/amd.c:72-239) */
static void init_amd(wo_channel *c)
{
    static const double c0[AMD_STAGES] = { -0.328201924180698, -0.744171491539427, -0.923022915444215,
        -0.978490468768238, -0.994128272402075, -0.998458978159551, -0.999790306259206 };
    static const double c1[AMD_STAGES] = { -0.0991227952747244, -0.565619728761389, -0.857467122550052,
        -0.959123933111275, -0.988739372718090, -0.996959189310611, -0.999282492800792 };
    double fs = c->amd.sample_rate;
    double fmin = -2000.0, fmax = +2000.0, zeta = 1.0, omegaN = 250.0, tauR = 0.02, tauI = 1.4;  /* RXA.c:183-189 */
    c->amd.omega_min = WO_TWOPI * fmin / fs;
    c->amd.omega_max = WO_TWOPI * fmax / fs;
    c->amd.g1 = 1.0 - exp(-2.0 * omegaN * zeta / fs);
    c->amd.g2 = -c->amd.g1 + 2.0 * (1 - exp(-omegaN * zeta / fs) * cos(omegaN / fs * sqrt(1.0 - zeta * zeta)));
    c->amd.phs = 0.0; c->amd.fil_out = 0.0; c->amd.omega = 0.0;
    c->amd.dc = 0.0; c->amd.dc_insert = 0.0;
    c->amd.mtauR = exp(-1.0 / (fs * tauR));
    c->amd.onem_mtauR = 1.0 - c->amd.mtauR;
    c->amd.mtauI = exp(-1.0 / (fs * tauI));
    c->amd.onem_mtauI = 1.0 - c->amd.mtauI;
    memcpy(c->amd.c0, c0, sizeof(c0));
    memcpy(c->amd.c1, c1, sizeof(c1));
}

static void xamd(wo_channel *c, double *buf, int size)
{
    int i, j, k;
    double audio, vco[2], corr[2], det, del_out;
    double ai, bi, aq, bq;
    double ai_ps = 0, bi_ps = 0, aq_ps = 0, bq_ps = 0;
    if (!c->amd.run) return;
    switch (c->amd.mode) {
    case 0:     /* AM, amd.c:131-146 */
        for (i = 0; i < size; i++) {
            audio = sqrt(buf[2 * i + 0] * buf[2 * i + 0] + buf[2 * i + 1] * buf[2 * i + 1]);
            if (c->amd.levelfade) {
                c->amd.dc = c->amd.mtauR * c->amd.dc + c->amd.onem_mtauR * audio;
                c->amd.dc_insert = c->amd.mtauI * c->amd.dc_insert + c->amd.onem_mtauI * audio;
                audio += c->amd.dc_insert - c->amd.dc;
            }
            buf[2 * i + 0] = audio;
            buf[2 * i + 1] = audio;
        }
        break;
    case 1:     /* SAM, amd.c:148-232 */
        for (i = 0; i < size; i++) {
            vco[0] = cos(c->amd.phs);
            vco[1] = sin(c->amd.phs);
            ai = buf[2 * i + 0] * vco[0];
            bi = buf[2 * i + 0] * vco[1];
            aq = buf[2 * i + 1] * vco[0];
            bq = buf[2 * i + 1] * vco[1];
            if (c->amd.sbmode != 0) {
                double *a = c->amd.a, *b = c->amd.b, *cc = c->amd.c, *d = c->amd.d;
                a[0] = c->amd.dsI; b[0] = bi; cc[0] = c->amd.dsQ; d[0] = aq;
                c->amd.dsI = ai; c->amd.dsQ = bq;
                for (j = 0; j < AMD_STAGES; j++) {
                    k = 3 * j;
                    a[k + 3] = c->amd.c0[j] * (a[k] - a[k + 5]) + a[k + 2];
                    b[k + 3] = c->amd.c1[j] * (b[k] - b[k + 5]) + b[k + 2];
                    cc[k + 3] = c->amd.c0[j] * (cc[k] - cc[k + 5]) + cc[k + 2];
                    d[k + 3] = c->amd.c1[j] * (d[k] - d[k + 5]) + d[k + 2];
                }
                ai_ps = a[AMD_OUT_IDX]; bi_ps = b[AMD_OUT_IDX]; bq_ps = cc[AMD_OUT_IDX]; aq_ps = d[AMD_OUT_IDX];
                for (j = AMD_OUT_IDX + 2; j > 0; j--) {
                    a[j] = a[j - 1]; b[j] = b[j - 1]; cc[j] = cc[j - 1]; d[j] = d[j - 1];
                }
            }
            corr[0] = +ai + bq;
            corr[1] = -bi + aq;
            switch (c->amd.sbmode) {
            default:
            case 0: audio = corr[0]; break;
            case 1: audio = (ai_ps - bi_ps) + (aq_ps + bq_ps); break;
            case 2: audio = (ai_ps + bi_ps) - (aq_ps - bq_ps); break;
            }
            if (c->amd.levelfade) {
                c->amd.dc = c->amd.mtauR * c->amd.dc + c->amd.onem_mtauR * audio;
                c->amd.dc_insert = c->amd.mtauI * c->amd.dc_insert + c->amd.onem_mtauI * corr[0];
                audio += c->amd.dc_insert - c->amd.dc;
            }
            buf[2 * i + 0] = audio;
            buf[2 * i + 1] = audio;
            if ((corr[0] == 0.0) && (corr[1] == 0.0)) corr[0] = 1.0;
            det = atan2(corr[1], corr[0]);
            del_out = c->amd.fil_out;
            c->amd.omega += c->amd.g2 * det;
            if (c->amd.omega < c->amd.omega_min) c->amd.omega = c->amd.omega_min;
            if (c->amd.omega > c->amd.omega_max) c->amd.omega = c->amd.omega_max;
            c->amd.fil_out = c->amd.g1 * det + c->amd.omega;
            c->amd.phs += del_out;
            while (c->amd.phs >= WO_TWOPI) c->amd.phs -= WO_TWOPI;
            while (c->amd.phs < 0.0) c->amd.phs += WO_TWOPI;
        }
        break;
    }
}

/* ---- fmd (wdsp/fmd.c:29-188) */
static void calc_snotch(wo_channel *c)          /* iir.c:35-49 */
{
    double fn, qk, qr, csn, bw = 0.0002;       /* fmd.c:47 */
    fn = c->fmd.ctcss_freq / (double)(int)c->fmd.rate;
    csn = cos(WO_TWOPI * fn);
    qr = 1.0 - 3.0 * bw;
    qk = (1.0 - 2.0 * qr * csn + qr * qr) / (2.0 * (1.0 - csn));
    c->fmd.a0 = +qk;
    c->fmd.a1 = -2.0 * qk * csn;
    c->fmd.a2 = +qk;
    c->fmd.b1 = +2.0 * qr * csn;
    c->fmd.b2 = -qr * qr;
    c->fmd.x1 = c->fmd.x2 = c->fmd.y1 = c->fmd.y2 = 0.0;
}

static void calc_fmd(wo_channel *c)             /* fmd.c:29-47 */
{
    double zeta = c->fmd.zeta, omegaN = c->fmd.omegaN, rate = c->fmd.rate;
    c->fmd.omega_min = WO_TWOPI * c->fmd.fmin / rate;
    c->fmd.omega_max = WO_TWOPI * c->fmd.fmax / rate;
    c->fmd.g1 = 1.0 - exp(-2.0 * omegaN * zeta / rate);
    c->fmd.g2 = -c->fmd.g1 + 2.0 * (1 - exp(-omegaN * zeta / rate) * cos(omegaN / rate * sqrt(1.0 - zeta * zeta)));
    c->fmd.phs = 0.0; c->fmd.fil_out = 0.0; c->fmd.omega = 0.0;
    c->fmd.mtau = exp(-1.0 / (rate * c->fmd.tau));
    c->fmd.onem_mtau = 1.0 - c->fmd.mtau;
    c->fmd.fmdc = 0.0;
    c->fmd.again = rate / (c->fmd.deviation * WO_TWOPI);
    calc_snotch(c);
}

static double *fmd_de_impulse(wo_channel *c)    /* fmd.c:110 */
{
    return wo_fc_impulse(c->fmd.nc_de, c->fmd.f_low, c->fmd.f_high, +20.0 * log10(c->fmd.f_high / c->fmd.f_low),
                         0.0, 1, c->fmd.rate, 1.0 / (2.0 * c->dsp_size), 0, 0);
}

static double *fmd_aud_impulse(wo_channel *c)   /* fmd.c:114 */
{
    return wo_fir_bandpass(c->fmd.nc_aud, 0.8 * c->fmd.f_low, 1.1 * c->fmd.f_high, c->fmd.rate, 0, 1,
                           c->fmd.afgain / (2.0 * c->dsp_size));
}

static void fmd_make_limiter(wo_channel *c)          /* calc_fmd, fmd.c:48-72 */
{
    wo_agc_init(&c->fmd.plim, 1, 5, 1, (int)c->fmd.rate, 0.001, 0.008, 4, c->fmd.lim_gain, 1.0, 1.0, 1.0, 0.9,
                0.250, 0.004, 4.0, 0, 0.500, 0.500, 2.000, 0.100);
}

static void xfmd(wo_channel *c, double *buf, int size)   /* fmd.c:144-188 */
{
    int i;
    double det, del_out, vco[2], corr[2], x0;
    if (!c->fmd.run) return;
    for (i = 0; i < size; i++) {
        vco[0] = cos(c->fmd.phs);
        vco[1] = sin(c->fmd.phs);
        corr[0] = +buf[2 * i + 0] * vco[0] + buf[2 * i + 1] * vco[1];
        corr[1] = -buf[2 * i + 0] * vco[1] + buf[2 * i + 1] * vco[0];
        if ((corr[0] == 0.0) && (corr[1] == 0.0)) corr[0] = 1.0;
        det = atan2(corr[1], corr[0]);
        del_out = c->fmd.fil_out;
        c->fmd.omega += c->fmd.g2 * det;
        if (c->fmd.omega < c->fmd.omega_min) c->fmd.omega = c->fmd.omega_min;
        if (c->fmd.omega > c->fmd.omega_max) c->fmd.omega = c->fmd.omega_max;
        c->fmd.fil_out = c->fmd.g1 * det + c->fmd.omega;
        c->fmd.phs += del_out;
        while (c->fmd.phs >= WO_TWOPI) c->fmd.phs -= WO_TWOPI;
        while (c->fmd.phs < 0.0) c->fmd.phs += WO_TWOPI;
        c->fmd.fmdc = c->fmd.mtau * c->fmd.fmdc + c->fmd.onem_mtau * c->fmd.fil_out;
        c->fmd.audio[2 * i + 0] = c->fmd.again * (c->fmd.fil_out - c->fmd.fmdc);
        c->fmd.audio[2 * i + 1] = c->fmd.audio[2 * i + 0];
    }
    wo_fircore_exec(c->fmd.pde, c->fmd.audio, buf);   /* de-emphasis: audio -> out */
    wo_fircore_exec(c->fmd.paud, buf, buf);           /* audio filter, in place */
    if (c->fmd.sntch_run) {                           /* xsnotch, iir.c:76-95: filters the I part only */
        for (i = 0; i < size; i++) {
            x0 = buf[2 * i + 0];
            buf[2 * i + 0] = c->fmd.a0 * x0 + c->fmd.a1 * c->fmd.x1 + c->fmd.a2 * c->fmd.x2
                           + c->fmd.b1 * c->fmd.y1 + c->fmd.b2 * c->fmd.y2;
            c->fmd.y2 = c->fmd.y1;
            c->fmd.y1 = buf[2 * i + 0];
            c->fmd.x2 = c->fmd.x1;
            c->fmd.x1 = x0;
        }
    }
    if (c->fmd.lim_run) {                             /* fmd.c:179-184 */
        for (i = 0; i < 2 * size; i++) buf[i] *= c->fmd.lim_pre_gain;
        wo_agc_exec(&c->fmd.plim, buf, size);
    }
}

/* ---- agc (wcpAGC.c:161-342, restated in wcpagc_oracle.c) */
static void xwcpagc(wo_channel *c, double *buf, int size)
{
    wo_agc_exec(&c->agc, buf, size);
}

/* ---- panel (patchpanel.c:55-101): the run flag is not examined */
static void xpanel(wo_channel *c, double *buf, int size)
{
    int i;
    double I, Q;
    double gainI = c->panel.gain1 * c->panel.gain2I;
    double gainQ = c->panel.gain1 * c->panel.gain2Q;
    switch (c->panel.copy) {
    case 0:
        for (i = 0; i < size; i++) {
            I = buf[2 * i + 0] * (c->panel.inselect >> 1);
            Q = buf[2 * i + 1] * (c->panel.inselect & 1);
            buf[2 * i + 0] = gainI * I; buf[2 * i + 1] = gainQ * Q;
        }
        break;
    case 1:
        for (i = 0; i < size; i++) {
            I = buf[2 * i + 0] * (c->panel.inselect >> 1);
            Q = I;
            buf[2 * i + 0] = gainI * I; buf[2 * i + 1] = gainQ * Q;
        }
        break;
    case 2:
        for (i = 0; i < size; i++) {
            Q = buf[2 * i + 1] * (c->panel.inselect & 1);
            I = Q;
            buf[2 * i + 0] = gainI * I; buf[2 * i + 1] = gainQ * Q;
        }
        break;
    case 3:
        for (i = 0; i < size; i++) {
            Q = buf[2 * i + 0] * (c->panel.inselect >> 1);
            I = buf[2 * i + 1] * (c->panel.inselect & 1);
            buf[2 * i + 0] = gainI * I; buf[2 * i + 1] = gainQ * Q;
        }
        break;
    }
}

/* ---- AM squelch: calc_amsq / compute_slews (amsq.c:28-64), xamsqcap (:189-192), xamsq (:119-187) */
static void amsq_calc(wo_channel *c)
{
    int i;
    double delta, theta;
    c->amsq.rate = (double)c->dsp_rate;
    c->amsq.trigsig = (double *)zalloc((size_t)c->dsp_size * 2 * sizeof(double));
    c->amsq.avm = exp(-1.0 / (c->amsq.rate * 0.010));
    c->amsq.onem_avm = 1.0 - c->amsq.avm;
    c->amsq.ntup = (int)(0.070 * c->amsq.rate);
    c->amsq.ntdown = (int)(0.070 * c->amsq.rate);
    c->amsq.cup = (double *)zalloc((size_t)(c->amsq.ntup + 1) * sizeof(double));
    c->amsq.cdown = (double *)zalloc((size_t)(c->amsq.ntdown + 1) * sizeof(double));
    delta = WO_PI / (double)c->amsq.ntup; theta = 0.0;
    for (i = 0; i <= c->amsq.ntup; i++) { c->amsq.cup[i] = c->amsq.muted_gain + (1.0 - c->amsq.muted_gain) * 0.5 * (1.0 - cos(theta)); theta += delta; }
    delta = WO_PI / (double)c->amsq.ntdown; theta = 0.0;
    for (i = 0; i <= c->amsq.ntdown; i++) { c->amsq.cdown[i] = c->amsq.muted_gain + (1.0 - c->amsq.muted_gain) * 0.5 * (1.0 + cos(theta)); theta += delta; }
}

static void xamsq(wo_channel *c, double *buf, int size)
{
    enum { MUTED, INCREASE, UNMUTED, TAIL, DECREASE };
    int i;
    double sig, siglimit, g;
    if (!c->amsq.run) return;
    for (i = 0; i < size; i++) {
        sig = sqrt(c->amsq.trigsig[2 * i] * c->amsq.trigsig[2 * i] + c->amsq.trigsig[2 * i + 1] * c->amsq.trigsig[2 * i + 1]);
        c->amsq.avsig = c->amsq.avm * c->amsq.avsig + c->amsq.onem_avm * sig;
        g = 1.0;
        switch (c->amsq.state) {
        case MUTED:
            if (c->amsq.avsig > c->amsq.unmute_thresh) { c->amsq.state = INCREASE; c->amsq.count = c->amsq.ntup; }
            g = c->amsq.muted_gain;
            break;
        case INCREASE:
            g = c->amsq.cup[c->amsq.ntup - c->amsq.count];
            if (c->amsq.count-- == 0) c->amsq.state = UNMUTED;
            break;
        case UNMUTED:
            if (c->amsq.avsig < c->amsq.tail_thresh) {
                c->amsq.state = TAIL;
                if ((siglimit = c->amsq.avsig) > 1.0) siglimit = 1.0;
                c->amsq.count = (int)((c->amsq.min_tail + (c->amsq.max_tail - c->amsq.min_tail) * (1.0 - siglimit)) * c->amsq.rate);
            }
            break;
        case TAIL:
            if (c->amsq.avsig > c->amsq.unmute_thresh) c->amsq.state = UNMUTED;
            else if (c->amsq.count-- == 0) { c->amsq.state = DECREASE; c->amsq.count = c->amsq.ntdown; }
            break;
        case DECREASE:
            g = c->amsq.cdown[c->amsq.ntdown - c->amsq.count];
            if (c->amsq.count-- == 0) c->amsq.state = MUTED;
            break;
        }
        if (g != 1.0) { buf[2 * i] = g * buf[2 * i]; buf[2 * i + 1] = g * buf[2 * i + 1]; }      /* out = in elsewhere */
    }
}

/* ---- xanf (anf.c:82-133) and xanr (anr.c:82-133): the same leaky LMS line enhancer; the notch filter outputs the
 * error, the noise reduction the prediction.  Real part only; the imaginary part comes out zero. */
static void xlms(struct wo_lms *a, int is_anr, int position, double *buf, int size)
{
    int i, j, idx;
    double c0, c1, y, error, sigma, inv_sigp, nel, nev;
    if (!(a->run && a->position == position)) return;
    for (i = 0; i < size; i++) {
        a->d[a->in_idx] = buf[2 * i + 0];
        y = 0; sigma = 0;
        for (j = 0; j < a->n_taps; j++) {
            idx = (a->in_idx + j + a->delay) & a->mask;
            y += a->w[j] * a->d[idx];
            sigma += a->d[idx] * a->d[idx];
        }
        inv_sigp = 1.0 / (sigma + 1e-10);
        error = a->d[a->in_idx] - y;
        buf[2 * i + 0] = is_anr ? y : error;
        buf[2 * i + 1] = 0.0;
        if ((nel = error * (1.0 - a->two_mu * sigma * inv_sigp)) < 0.0) nel = -nel;
        if ((nev = a->d[a->in_idx] - (1.0 - a->two_mu * a->ngamma) * y - a->two_mu * error * sigma * inv_sigp) < 0.0) nev = -nev;
        if (nev < nel) { if ((a->lidx += a->lincr) > a->lidx_max) a->lidx = a->lidx_max; }
        else { if ((a->lidx -= a->ldecr) < a->lidx_min) a->lidx = a->lidx_min; }
        a->ngamma = a->gamma * (a->lidx * a->lidx) * (a->lidx * a->lidx) * a->den_mult;
        c0 = 1.0 - a->two_mu * a->ngamma;
        c1 = a->two_mu * error * inv_sigp;
        for (j = 0; j < a->n_taps; j++) {
            idx = (a->in_idx + j + a->delay) & a->mask;
            a->w[j] = c0 * a->w[j] + c1 * a->d[idx];
        }
        a->in_idx = (a->in_idx + a->mask) & a->mask;
    }
}

static void lms_flush(struct wo_lms *a)          /* flush_anf, anf.c:135-140 */
{
    memset(a->d, 0, sizeof(a->d));
    memset(a->w, 0, sizeof(a->w));
    a->in_idx = 0;
}

/* ---- RXAbp1Check / RXAbp1Set (RXA.c:800-827) */
static void bpsnba_check(wo_channel *c, int mode, int notch_run)     /* RXAbpsnbaCheck, RXA.c:829-881 */
{
    double f_low = 0.0, f_high = 0.0;
    int run_notches = 0;
    switch (mode) {
    case WO_LSB: case WO_CWL: case WO_DIGL: f_low = -c->bpsnba.abs_high; f_high = -c->bpsnba.abs_low; run_notches = notch_run; break;
    case WO_USB: case WO_CWU: case WO_DIGU: f_low = +c->bpsnba.abs_low; f_high = +c->bpsnba.abs_high; run_notches = notch_run; break;
    case WO_AM: case WO_SAM: case WO_DSB: case WO_FM: f_low = +c->bpsnba.abs_low; f_high = +c->bpsnba.abs_high; break;
    default: break;
    }
    if (c->bpsnba.f_low != f_low || c->bpsnba.f_high != f_high || c->bpsnba.run_notches != run_notches) {
        c->bpsnba.f_low = f_low; c->bpsnba.f_high = f_high; c->bpsnba.run_notches = run_notches;
        bpsnba_update(c);
    }
}

static void bpsnba_set(wo_channel *c)           /* RXAbpsnbaSet, RXA.c:883-917 */
{
    switch (c->mode) {
    case WO_LSB: case WO_CWL: case WO_DIGL: case WO_USB: case WO_CWU: case WO_DIGU:
        c->bpsnba.run = *wo_snba_run(c->snba); c->bpsnba.position = 0; break;
    case WO_AM: case WO_SAM: case WO_DSB: case WO_FM:
        c->bpsnba.run = *wo_snba_run(c->snba); c->bpsnba.position = 1; break;
    default: c->bpsnba.run = 0; break;
    }
}

static void bp1_check(wo_channel *c, int amd_run, int anf_run, int anr_run)
{
    double gain = (amd_run || anf_run || anr_run || c->emnr_pending_run || c->snba_pending_run) ? 2.0 : 1.0;
    if (c->bp1.gain != gain) {
        double *imp;
        c->bp1.gain = gain;
        imp = bp1_impulse(c);
        fircore_set_impulse(c->bp1.p, imp);
        free(imp);
    }
}

static void bp1_set(wo_channel *c)
{
    int old = c->bp1.run;
    c->bp1.run = (c->amd.run == 1 || c->anf.run == 1 || c->anr.run == 1 || *wo_emnr_run(c->emnr) == 1 || *wo_snba_run(c->snba) == 1) ? 1 : 0;
    if (!old && c->bp1.run) fircore_flush(c->bp1.p);
}

/* ---- xrxa (RXA.c:561-598); blocks that are run=0 on this path are omitted */
static void xrxa(wo_channel *c)
{
    int n = c->dsp_size;
    xshift(c, c->inbuff, c->dsp_insize);
    if (c->rsmpin->run) wo_resample_exec(c->rsmpin, c->inbuff, c->dsp_insize, c->midbuff);
    else memcpy(c->midbuff, c->inbuff, (size_t)c->dsp_insize * 2 * sizeof(double));
    meter_exec(&c->adcmeter, c->midbuff, n, c->meter, NULL);
    if (c->bpsnba.run && c->bpsnba.position == 0) memcpy(c->bpsnba.buff, c->midbuff, (size_t)n * 2 * sizeof(double));  /* xbpsnbain, RXA.c:567 */
    if (c->nbp0.run) wo_fircore_exec(c->nbp0.p, c->midbuff, c->midbuff);
    meter_exec(&c->smeter, c->midbuff, n, c->meter, NULL);
    memcpy(c->amsq.trigsig, c->midbuff, (size_t)n * 2 * sizeof(double));            /* xamsqcap, RXA.c:571 */
    if (c->bpsnba.run && c->bpsnba.position == 0) wo_fircore_exec(c->bpsnba.p, c->bpsnba.buff, c->midbuff);           /* xbpsnbaout, RXA.c:572 */
    xamd(c, c->midbuff, n);
    xfmd(c, c->midbuff, n);
    if (c->bpsnba.run && c->bpsnba.position == 1) wo_fircore_exec(c->bpsnba.p, c->midbuff, c->midbuff);              /* RXA.c:576-577 */
    wo_snba_exec(c->snba, c->midbuff);                                                                                 /* xsnba, RXA.c:578 */
    xlms(&c->anf, 0, 0, c->midbuff, n);
    xlms(&c->anr, 1, 0, c->midbuff, n);
    wo_emnr_exec(c->emnr, 0, c->midbuff);
    if (c->bp1.run && c->bp1.position == 0) wo_fircore_exec(c->bp1.p, c->midbuff, c->midbuff);
    xwcpagc(c, c->midbuff, n);
    xlms(&c->anf, 0, 1, c->midbuff, n);
    xlms(&c->anr, 1, 1, c->midbuff, n);
    wo_emnr_exec(c->emnr, 1, c->midbuff);
    if (c->bp1.run && c->bp1.position == 1) wo_fircore_exec(c->bp1.p, c->midbuff, c->midbuff);
    meter_exec(&c->agcmeter, c->midbuff, n, c->meter, &c->agc.gain);
    xpanel(c, c->midbuff, n);
    xamsq(c, c->midbuff, n);                                                        /* RXA.c:596 */
    if (c->rsmpout->run) wo_resample_exec(c->rsmpout, c->midbuff, n, c->outbuff);
    else memcpy(c->outbuff, c->midbuff, (size_t)n * 2 * sizeof(double));
}

void wo_xrxa_block(wo_channel *c, const double *in, double *out)
{
    memcpy(c->inbuff, in, (size_t)c->dsp_insize * 2 * sizeof(double));
    xrxa(c);
    memcpy(out, c->outbuff, (size_t)c->dsp_outsize * 2 * sizeof(double));
}

/* many blocks in one call (lets a Python caller release the GIL for the whole run) */
void wo_xrxa_blocks(wo_channel *c, const double *in, double *out, int nblk)
{
    int b;
    for (b = 0; b < nblk; b++)
        wo_xrxa_block(c, in + (size_t)b * c->dsp_insize * 2, out + (size_t)b * c->dsp_outsize * 2);
}

/* ---- slews (iobuffs.c:47-160) */
enum { SL_BEGIN = 0, SL_DELAYUP, SL_UPSLEW, SL_ON };

static void create_slews(wo_channel *c)
{
    int i;
    double delta, theta;
    c->iob.ustate = SL_BEGIN;
    c->iob.ucount = 0;
    c->iob.ndelup = (int)(c->tdelayup * c->in_rate);
    c->iob.ntup = (int)(c->tslewup * c->in_rate);
    c->iob.cup = (double *)zalloc((size_t)(c->iob.ntup + 1) * sizeof(double));
    delta = WO_PI / (double)c->iob.ntup;
    theta = 0.0;
    for (i = 0; i <= c->iob.ntup; i++) {
        c->iob.cup[i] = 0.5 * (1.0 - cos(theta));
        theta += delta;
    }
    c->iob.upflag = 0;
}

static void upslew0(wo_channel *c, const double *pin)
{
    int i;
    double *pout = c->iob.r1 + 2 * c->iob.r1_inidx;
    double I, Q;
    for (i = 0; i < c->in_size; i++) {
        I = pin[2 * i + 0];
        Q = pin[2 * i + 1];
        switch (c->iob.ustate) {
        case SL_BEGIN:
            pout[2 * i + 0] = 0.0; pout[2 * i + 1] = 0.0;
            if ((I != 0.0) || (Q != 0.0)) {
                if (c->iob.ndelup > 0) { c->iob.ustate = SL_DELAYUP; c->iob.ucount = c->iob.ndelup; }
                else if (c->iob.ntup > 0) { c->iob.ustate = SL_UPSLEW; c->iob.ucount = c->iob.ntup; }
                else c->iob.ustate = SL_ON;
            }
            break;
        case SL_DELAYUP:
            pout[2 * i + 0] = 0.0; pout[2 * i + 1] = 0.0;
            if (c->iob.ucount-- == 0) {
                if (c->iob.ntup > 0) { c->iob.ustate = SL_UPSLEW; c->iob.ucount = c->iob.ntup; }
                else c->iob.ustate = SL_ON;
            }
            break;
        case SL_UPSLEW:
            pout[2 * i + 0] = I * c->iob.cup[c->iob.ntup - c->iob.ucount];
            pout[2 * i + 1] = Q * c->iob.cup[c->iob.ntup - c->iob.ucount];
            if (c->iob.ucount-- == 0) c->iob.ustate = SL_ON;
            break;
        case SL_ON:
            pout[2 * i + 0] = I; pout[2 * i + 1] = Q;
            if (i == c->in_size - 1) { c->iob.ustate = SL_BEGIN; c->iob.upflag = 0; }
            break;
        }
    }
}

/* ---- dexchange (iobuffs.c:583-604) run synchronously for every queued block */
static void dexchange(wo_channel *c)
{
    int n;
    c->iob.r2_havesamps += c->iob.r2_insize;
    memcpy(c->iob.r2 + 2 * c->iob.r2_inidx, c->outbuff, (size_t)c->iob.r2_insize * 2 * sizeof(double));
    if ((c->iob.r2_inidx += c->iob.r2_insize) == c->iob.r2_active) c->iob.r2_inidx = 0;
    if ((c->iob.r2_unqueued += c->iob.r2_insize) >= c->out_size) {      /* bfo = 1 */
        n = c->iob.r2_unqueued / c->out_size;
        c->iob.sem_outready += n;
        c->iob.r2_unqueued -= n * c->out_size;
    }
    memcpy(c->inbuff, c->iob.r1 + 2 * c->iob.r1_outidx, (size_t)c->iob.r1_outsize * 2 * sizeof(double));
    if ((c->iob.r1_outidx += c->iob.r1_outsize) == c->iob.r1_active) c->iob.r1_outidx = 0;
}

void wo_fexchange0(wo_channel *c, const double *in, double *out, int *error)   /* iobuffs.c:464-516, bfo = 1 */
{
    int n;
    *error = 0;
    if (c->iob.upflag) upslew0(c, in);
    else memcpy(c->iob.r1 + 2 * c->iob.r1_inidx, in, (size_t)c->in_size * 2 * sizeof(double));
    if ((c->iob.r1_unqueued += c->in_size) >= c->iob.r1_outsize) {
        n = c->iob.r1_unqueued / c->iob.r1_outsize;
        c->iob.sem_buffready += n;
        c->iob.r1_unqueued -= n * c->iob.r1_outsize;
    }
    if ((c->iob.r1_inidx += c->in_size) == c->iob.r1_active) c->iob.r1_inidx = 0;
    /* DSP thread (main.c:40-58): one dexchange + xrxa per released BuffReady count */
    while (c->iob.sem_buffready > 0) {
        c->iob.sem_buffready--;
        dexchange(c);
        xrxa(c);
    }
    if ((c->iob.r2_havesamps -= c->out_size) < 0) c->iob.r2_havesamps = 0;
    if (c->iob.sem_outready > 0) {
        c->iob.sem_outready--;                  /* WaitForSingleObject(Sem_OutReady) returns */
        memcpy(out, c->iob.r2 + 2 * c->iob.r2_outidx, (size_t)c->out_size * 2 * sizeof(double));
    } else {
        /* the reference would block here for ever; report it the way the non-bfo path does */
        memset(out, 0, (size_t)c->out_size * 2 * sizeof(double));
        *error += -2;
    }
    if ((c->iob.r2_outidx += c->out_size) == c->iob.r2_active) c->iob.r2_outidx = 0;
}

/* ---- create (channel.c:37-103, iobuffs.c:384-423, RXA.c:31-490) */
wo_channel *wo_open(int in_size, int dsp_size, int in_rate, int dsp_rate, int out_rate,
                    double tdelayup, double tslewup, double tdelaydown, double tslewdown)
{
    wo_channel *c = (wo_channel *)zalloc(sizeof(*c));
    double *imp;
    int n, nc;
    c->in_size = in_size; c->dsp_size = dsp_size;
    c->in_rate = in_rate; c->dsp_rate = dsp_rate; c->out_rate = out_rate;
    c->tdelayup = tdelayup; c->tslewup = tslewup; c->tdelaydown = tdelaydown; c->tslewdown = tslewdown;
    /* pre_main_build, channel.c:39-52 */
    if (in_rate >= dsp_rate) c->dsp_insize = dsp_size * (in_rate / dsp_rate);
    else c->dsp_insize = dsp_size / (dsp_rate / in_rate);
    if (out_rate >= dsp_rate) c->dsp_outsize = dsp_size * (out_rate / dsp_rate);
    else c->dsp_outsize = dsp_size / (dsp_rate / out_rate);
    if (in_rate >= out_rate) c->out_size = in_size / (in_rate / out_rate);
    else c->out_size = in_size * (out_rate / in_rate);
    /* create_iobuffs, iobuffs.c:384-423 */
    c->iob.r1_outsize = c->dsp_insize;
    c->iob.r1_size = imax(c->iob.r1_outsize, in_size);
    c->iob.r2_insize = c->dsp_outsize;
    c->iob.r2_size = imax(c->out_size, c->iob.r2_insize);
    c->iob.r1_active = WO_DSP_MULT * c->iob.r1_size;
    c->iob.r2_active = WO_DSP_MULT * c->iob.r2_size;
    c->iob.r1 = (double *)zalloc((size_t)c->iob.r1_active * 2 * sizeof(double));
    c->iob.r2 = (double *)zalloc((size_t)c->iob.r2_active * 2 * sizeof(double));
    c->iob.r2_inidx = (WO_DSP_MULT - 1) * c->iob.r2_size;
    c->iob.r2_havesamps = (WO_DSP_MULT - 1) * c->iob.r2_size;
    n = c->iob.r2_havesamps / c->out_size;
    c->iob.r2_unqueued = c->iob.r2_havesamps - n * c->out_size;
    c->iob.sem_buffready = 0;
    c->iob.sem_outready = n;
    create_slews(c);
    c->iob.upflag = 1;                          /* OpenChannel with state 1, channel.c:94-95 */
    /* create_rxa, RXA.c:31-490 */
    c->mode = WO_LSB;
    c->inbuff = (double *)zalloc((size_t)c->dsp_insize * 2 * sizeof(double));
    c->outbuff = (double *)zalloc((size_t)c->dsp_outsize * 2 * sizeof(double));
    c->midbuff = (double *)zalloc((size_t)2 * imax(c->dsp_size, c->dsp_insize) * 2 * sizeof(double));
    c->shift.run = 1; c->shift.rate = (double)in_rate; c->shift.shift = 0.0; c->shift.phase = 0.0;
    calc_shift(c);
    c->rsmpin = wo_resample_create(in_rate, dsp_rate, 0.0, 0, 1.0);
    c->rsmpout = wo_resample_create(dsp_rate, out_rate, 0.0, 0, 1.0);
    c->rsmpin->run = (in_rate != dsp_rate);     /* RXAResCheck, RXA.c:789-798 */
    c->rsmpout->run = (dsp_rate != out_rate);
    meter_init(&c->adcmeter, dsp_rate, 0.100, 0.100, WO_ADC_AV, WO_ADC_PK, -1, c->meter);
    meter_init(&c->smeter, dsp_rate, 0.100, 0.100, WO_S_AV, WO_S_PK, -1, c->meter);
    meter_init(&c->agcmeter, dsp_rate, 0.100, 0.100, WO_AGC_AV, WO_AGC_PK, WO_AGC_GAIN, c->meter);
    nc = imax(2048, dsp_size);
    c->nbp0.run = 1; c->nbp0.nc = nc; c->nbp0.wintype = 0; c->nbp0.gain = 1.0;
    c->nbp0.fnfrun = 0; c->nbp0.autoincr = 1;                   /* RXA.c:92,104 */
    c->nbp0.flow = -4150.0; c->nbp0.fhigh = -150.0; c->nbp0.rate = (double)dsp_rate;
    imp = nbp0_impulse(c);
    c->nbp0.p = wo_fircore_create(dsp_size, nc, imp);
    free(imp);
    c->amd.run = 0; c->amd.mode = 0; c->amd.levelfade = 1; c->amd.sbmode = 0; c->amd.sample_rate = (double)dsp_rate;
    init_amd(c);
    c->fmd.run = 0; c->fmd.rate = (double)dsp_rate; c->fmd.deviation = 5000.0;
    c->fmd.f_low = 300.0; c->fmd.f_high = 3000.0; c->fmd.fmin = -8000.0; c->fmd.fmax = +8000.0;
    c->fmd.zeta = 1.0; c->fmd.omegaN = 20000.0; c->fmd.tau = 0.02; c->fmd.afgain = 0.5;
    c->fmd.sntch_run = 1; c->fmd.ctcss_freq = 254.1; c->fmd.nc_de = nc; c->fmd.nc_aud = nc;
    calc_fmd(c);
    c->fmd.lim_run = 0; c->fmd.lim_pre_gain = 0.4; c->fmd.lim_gain = 2.5;     /* fmd.c:106-108 */
    fmd_make_limiter(c);
    c->fmd.audio = (double *)zalloc((size_t)dsp_size * 2 * sizeof(double));
    imp = fmd_de_impulse(c);
    c->fmd.pde = wo_fircore_create(dsp_size, nc, imp);
    free(imp);
    imp = fmd_aud_impulse(c);
    c->fmd.paud = wo_fircore_create(dsp_size, nc, imp);
    free(imp);
    /* create_wcpagc arguments of create_rxa, RXA.c:335-358 */
    wo_agc_init(&c->agc, 1, 3, 1, dsp_rate, 0.001, 0.250, 4, 10000.0, 1.5, 1000.0, 1.0, 1.0, 0.250, 0.005, 5.0, 1, 0.500, 0.250, 0.250, 0.100);
    c->bp1.run = 1; c->bp1.nc = nc; c->bp1.wintype = 1; c->bp1.gain = 1.0; c->bp1.position = 0;
    c->emnr = wo_emnr_create(dsp_size, dsp_rate);
    c->snba = wo_snba_create(dsp_rate, dsp_size);
    c->bpsnba.run = 0; c->bpsnba.run_notches = 0; c->bpsnba.position = 0; c->bpsnba.nc = nc; c->bpsnba.wintype = 0; c->bpsnba.autoincr = 1;
    c->bpsnba.abs_low = 250.0; c->bpsnba.abs_high = 5700.0; c->bpsnba.f_low = -5700.0; c->bpsnba.f_high = -250.0;      /* RXA.c:109-127 */
    c->bpsnba.buff = (double *)zalloc((size_t)dsp_size * 2 * sizeof(double));
    imp = bpsnba_impulse(c);
    c->bpsnba.p = wo_fircore_create(dsp_size, nc, imp);
    free(imp);
    /* create_amsq arguments of create_rxa, RXA.c:158-172 */
    c->amsq.run = 0; c->amsq.tail_thresh = 0.009; c->amsq.unmute_thresh = 0.010; c->amsq.min_tail = 0.0; c->amsq.max_tail = 1.5;
    c->amsq.muted_gain = 0.0;
    amsq_calc(c);
    {   /* create_anf / create_anr arguments of create_rxa, RXA.c:278-315 */
        struct wo_lms *a = &c->anf;
        a->mask = 2047; a->n_taps = 64; a->delay = 16; a->two_mu = 0.0001; a->gamma = 0.1;
        a->lidx = 1.0; a->lidx_min = 0.0; a->lidx_max = 200.0; a->ngamma = 6.25e-12; a->den_mult = 6.25e-10; a->lincr = 1.0; a->ldecr = 3.0;
        c->anr = c->anf;
        c->anr.lidx = 120.0; c->anr.lidx_min = 120.0; c->anr.ngamma = 0.001;
    }
    c->bp1.f_low = -4150.0; c->bp1.f_high = -150.0;
    imp = bp1_impulse(c);
    c->bp1.p = wo_fircore_create(dsp_size, nc, imp);
    free(imp);
    c->panel.run = 1; c->panel.gain1 = 4.0; c->panel.gain2I = 1.0; c->panel.gain2Q = 1.0;
    c->panel.inselect = 3; c->panel.copy = 0;
    return c;
}

void wo_close(wo_channel *c)
{
    if (!c) return;
    wo_resample_destroy(c->rsmpin); wo_resample_destroy(c->rsmpout);
    wo_fircore_destroy(c->nbp0.p); wo_fircore_destroy(c->bp1.p);
    wo_fircore_destroy(c->fmd.pde); wo_fircore_destroy(c->fmd.paud);
    free(c->fmd.audio);
    free(c->inbuff); free(c->midbuff); free(c->outbuff);
    free(c->amsq.cup); free(c->amsq.cdown); free(c->amsq.trigsig);
    wo_emnr_free(c->emnr);
    wo_snba_free(c->snba); wo_fircore_destroy(c->bpsnba.p); free(c->bpsnba.buff);
    free(c->iob.r1); free(c->iob.r2); free(c->iob.cup);
    wo_agc_free(&c->agc);
    free(c);
}

/* ---- setters */
void wo_SetRXAMode(wo_channel *c, int mode)     /* RXA.c:748-787 */
{
    if (c->mode != mode) {
        int amd_run = (mode == WO_AM) || (mode == WO_SAM);
        bpsnba_check(c, mode, c->ndb.master_run);
        bp1_check(c, amd_run, c->anf.run, c->anr.run);
        c->mode = mode;
        c->amd.run = 0;
        c->fmd.run = 0;
        c->agc.run = 1;
        switch (mode) {
        case WO_AM: c->amd.run = 1; c->amd.mode = 0; break;
        case WO_SAM: c->amd.run = 1; c->amd.mode = 1; break;
        case WO_FM: c->fmd.run = 1; c->agc.run = 0; break;
        default: break;
        }
        bp1_set(c);
        bpsnba_set(c);
    }
}

void wo_SetRXABandpassFreqs(wo_channel *c, double f_low, double f_high)
{
    if ((f_low != c->bp1.f_low) || (f_high != c->bp1.f_high)) {
        double *imp;
        c->bp1.f_low = f_low; c->bp1.f_high = f_high;
        imp = bp1_impulse(c);
        fircore_set_impulse(c->bp1.p, imp);
        free(imp);
    }
}

void wo_RXANBPSetFreqs(wo_channel *c, double flow, double fhigh)
{
    if ((flow != c->nbp0.flow) || (fhigh != c->nbp0.fhigh)) {
        double *imp;
        c->nbp0.flow = flow; c->nbp0.fhigh = fhigh;
        imp = nbp0_impulse(c);
        fircore_set_impulse(c->nbp0.p, imp);
        free(imp);
    }
}

/* the notch database, wdsp/nbp.c:358-525 */
int wo_RXANBPAddNotch(wo_channel *c, int notch, double fcenter, double fwidth, int active)
{
    int i, j;
    if (notch >= 0 && notch <= c->ndb.nn && c->ndb.nn < 1024) {
        c->ndb.nn++;
        for (i = c->ndb.nn - 2, j = c->ndb.nn - 1; i >= notch; i--, j--) {
            c->ndb.fcenter[j] = c->ndb.fcenter[i]; c->ndb.fwidth[j] = c->ndb.fwidth[i];
            c->ndb.nlow[j] = c->ndb.nlow[i]; c->ndb.nhigh[j] = c->ndb.nhigh[i]; c->ndb.active[j] = c->ndb.active[i];
        }
        c->ndb.fcenter[notch] = fcenter; c->ndb.fwidth[notch] = fwidth;
        c->ndb.nlow[notch] = fcenter - 0.5 * fwidth; c->ndb.nhigh[notch] = fcenter + 0.5 * fwidth;
        c->ndb.active[notch] = active;
        ndb_update(c);
        return 0;
    }
    return -1;
}

int wo_RXANBPDeleteNotch(wo_channel *c, int notch)
{
    int i, j;
    if (notch >= 0 && notch < c->ndb.nn) {
        c->ndb.nn--;
        for (i = notch, j = notch + 1; i < c->ndb.nn; i++, j++) {
            c->ndb.fcenter[i] = c->ndb.fcenter[j]; c->ndb.fwidth[i] = c->ndb.fwidth[j];
            c->ndb.nlow[i] = c->ndb.nlow[j]; c->ndb.nhigh[i] = c->ndb.nhigh[j]; c->ndb.active[i] = c->ndb.active[j];
        }
        ndb_update(c);
        return 0;
    }
    return -1;
}

int wo_RXANBPEditNotch(wo_channel *c, int notch, double fcenter, double fwidth, int active)
{
    if (notch >= 0 && notch < c->ndb.nn) {
        c->ndb.fcenter[notch] = fcenter; c->ndb.fwidth[notch] = fwidth;
        c->ndb.nlow[notch] = fcenter - 0.5 * fwidth; c->ndb.nhigh[notch] = fcenter + 0.5 * fwidth;
        c->ndb.active[notch] = active;
        ndb_update(c);
        return 0;
    }
    return -1;
}

void wo_RXANBPSetTuneFrequency(wo_channel *c, double f) { if (f != c->ndb.tunefreq) { c->ndb.tunefreq = f; ndb_update(c); } }
void wo_RXANBPSetShiftFrequency(wo_channel *c, double f) { if (f != c->ndb.shift) { c->ndb.shift = f; ndb_update(c); } }
void wo_RXANBPSetNotchesRun(wo_channel *c, int run)
{
    if (run != c->ndb.master_run) { c->ndb.master_run = run; c->nbp0.fnfrun = run; bpsnba_check(c, c->mode, run); nbp0_update(c); bpsnba_set(c); }
}
void wo_RXANBPSetWindow(wo_channel *c, int wintype)     /* nbp.c:543-561 */
{
    if (c->nbp0.wintype != wintype) { c->nbp0.wintype = wintype; nbp0_update(c); }
    if (c->bpsnba.wintype != wintype) { c->bpsnba.wintype = wintype; bpsnba_update(c); }
}
void wo_RXANBPSetAutoIncrease(wo_channel *c, int autoincr)      /* nbp.c:601-619 */
{
    if (c->nbp0.autoincr != autoincr) { c->nbp0.autoincr = autoincr; nbp0_update(c); }
    if (c->bpsnba.autoincr != autoincr) { c->bpsnba.autoincr = autoincr; bpsnba_update(c); }
}

void wo_RXASetPassband(wo_channel *c, double f_low, double f_high)
{
    wo_SetRXABandpassFreqs(c, f_low, f_high);
    wo_snba_set_output_bandwidth(c->snba, f_low, f_high);       /* SetRXASNBAOutputBandwidth, snb.c:660 */
    wo_RXANBPSetFreqs(c, f_low, f_high);
}

void wo_RXASetMP(wo_channel *c, int mp)         /* RXA.c:948-958: nbp0, bp1, FM de-emphasis and audio filter */
{
    if (c->nbp0.p->mp != mp) fircore_set_mp(c->nbp0.p, mp);       /* RXANBPSetMP, nbp.c:580-590 */
    if (c->bpsnba.p->mp != mp) fircore_set_mp(c->bpsnba.p, mp);   /* RXABPSNBASetMP, snb.c:845-855 */
    if (c->bp1.p->mp != mp) fircore_set_mp(c->bp1.p, mp);         /* SetRXABandpassMP, bandpass.c:448-457 */
    if (c->fmd.pde->mp != mp) fircore_set_mp(c->fmd.pde, mp);     /* SetRXAFMMPde, fmd.c:296-305 */
    if (c->fmd.paud->mp != mp) fircore_set_mp(c->fmd.paud, mp);   /* SetRXAFMMPaud, fmd.c:325-334 */
}

void wo_RXASetNC(wo_channel *c, int nc)         /* RXA.c:934-946 */
{
    double *imp;
    if (c->nbp0.nc != nc) {                     /* RXANBPSetNC, nbp.c:580-593 */
        c->nbp0.nc = nc;
        imp = nbp0_impulse(c);
        fircore_set_nc(c->nbp0.p, nc, imp);
        free(imp);
    }
    if (c->bpsnba.nc != nc) {                   /* RXABPSNBASetNC, snb.c:830-843 */
        c->bpsnba.nc = nc;
        imp = bpsnba_impulse(c);
        fircore_set_nc(c->bpsnba.p, nc, imp);
        free(imp);
    }
    if (c->bp1.nc != nc) {                      /* SetRXABandpassNC, bandpass.c:428-444 */
        c->bp1.nc = nc;
        imp = bp1_impulse(c);
        fircore_set_nc(c->bp1.p, nc, imp);
        free(imp);
    }
    if (c->fmd.nc_de != nc) {                   /* SetRXAFMNCde, fmd.c:269-284 */
        c->fmd.nc_de = nc;
        imp = fmd_de_impulse(c);
        fircore_set_nc(c->fmd.pde, nc, imp);
        free(imp);
    }
    if (c->fmd.nc_aud != nc) {                  /* SetRXAFMNCaud, fmd.c:298-313 */
        c->fmd.nc_aud = nc;
        imp = fmd_aud_impulse(c);
        fircore_set_nc(c->fmd.paud, nc, imp);
        free(imp);
    }
}

void wo_SetRXAShiftRun(wo_channel *c, int run) { c->shift.run = run; }
void wo_SetRXAShiftFreq(wo_channel *c, double fshift) { c->shift.shift = fshift; calc_shift(c); }
void wo_RXANBPSetRun(wo_channel *c, int run) { c->nbp0.run = run; }
void wo_SetRXABandpassRun(wo_channel *c, int run) { c->bp1.run = run; }

void wo_SetRXAAGCMode(wo_channel *c, int mode) { wo_agc_set_mode(&c->agc, mode); }
void wo_SetRXAAGCAttack(wo_channel *c, int attack) { c->agc.tau_attack = (double)attack / 1000.0; wo_agc_load(&c->agc); }
void wo_SetRXAAGCDecay(wo_channel *c, int decay) { c->agc.tau_decay = (double)decay / 1000.0; wo_agc_load(&c->agc); }
void wo_SetRXAAGCHang(wo_channel *c, int hang) { c->agc.hangtime = (double)hang / 1000.0; wo_agc_load(&c->agc); }
void wo_SetRXAAGCTop(wo_channel *c, double max_agc) { c->agc.max_gain = pow(10.0, max_agc / 20.0); wo_agc_load(&c->agc); }
void wo_SetRXAAGCSlope(wo_channel *c, int slope) { c->agc.var_gain = pow(10.0, (double)slope / 20.0 / 10.0); wo_agc_load(&c->agc); }
void wo_SetRXAAGCHangThreshold(wo_channel *c, int t) { c->agc.hang_thresh = (double)t / 100.0; wo_agc_load(&c->agc); }

void wo_SetRXAAGCFixed(wo_channel *c, double fixed_agc_db)
{
    c->agc.fixed_gain = pow(10.0, fixed_agc_db / 20.0);
    wo_agc_load(&c->agc);
}

void wo_SetRXAPanelRun(wo_channel *c, int run) { c->panel.run = run; }
void wo_SetRXAPanelGain1(wo_channel *c, double gain) { c->panel.gain1 = gain; }
void wo_SetRXAPanelGain2(wo_channel *c, double gainI, double gainQ) { c->panel.gain2I = gainI; c->panel.gain2Q = gainQ; }
void wo_SetRXAPanelSelect(wo_channel *c, int select) { c->panel.inselect = select; }
void wo_SetRXAPanelCopy(wo_channel *c, int copy) { c->panel.copy = copy; }
void wo_SetRXAAMDSBMode(wo_channel *c, int sbmode) { c->amd.sbmode = sbmode; }

void wo_SetRXAAMDRun(wo_channel *c, int run)        /* amd.c:264-277 */
{
    if (c->amd.run != run) {
        bp1_check(c, run, c->anf.run, c->anr.run);
        c->amd.run = run;
        bp1_set(c);
    }
}
/* SetRXAEMNRRun ... SetRXAEMNRPosition, emnr.c:1096-1142 */
void wo_SetRXAEMNRRun(wo_channel *c, int run)
{
    if (*wo_emnr_run(c->emnr) != run) {
        c->emnr_pending_run = run;
        bp1_check(c, c->amd.run, c->anf.run, c->anr.run);
        *wo_emnr_run(c->emnr) = run;
        bp1_set(c);
    }
}
void wo_SetRXASNBARun(wo_channel *c, int run)          /* snb.c:579-593 */
{
    if (*wo_snba_run(c->snba) != run) {
        bpsnba_check(c, c->mode, c->ndb.master_run);
        c->snba_pending_run = run;
        bp1_check(c, c->amd.run, c->anf.run, c->anr.run);
        *wo_snba_run(c->snba) = run;
        bp1_set(c);
        bpsnba_set(c);
    }
}
void wo_SetRXASNBAovrlp(wo_channel *c, int ovrlp) { wo_snba_set_ovrlp(c->snba, ovrlp); }                       /* snb.c:595-603 */
void wo_SetRXASNBATuning(wo_channel *c, int which, double v) { wo_snba_set_tuning(c->snba, which, v); }      /* snb.c:604-658 */
void wo_SetRXAEMNRgainMethod(wo_channel *c, int method) { wo_emnr_set_gain_method(c->emnr, method); }
void wo_SetRXAEMNRnpeMethod(wo_channel *c, int method) { wo_emnr_set_npe_method(c->emnr, method); }
void wo_SetRXAEMNRaeRun(wo_channel *c, int run) { wo_emnr_set_ae_run(c->emnr, run); }
void wo_SetRXAEMNRaeZetaThresh(wo_channel *c, double v) { wo_emnr_set_scalars(c->emnr, v, -1, -1e301, -1); }       /* emnr.c:1145 */
void wo_SetRXAEMNRaePsi(wo_channel *c, double v) { wo_emnr_set_scalars(c->emnr, -1, v, -1e301, -1); }              /* emnr.c:1153 */
void wo_SetRXAEMNRtrainZetaThresh(wo_channel *c, double v) { wo_emnr_set_scalars(c->emnr, -1, -1, v, -1); }        /* emnr.c:1161 */
void wo_SetRXAEMNRtrainT2(wo_channel *c, double v) { wo_emnr_set_scalars(c->emnr, -1, -1, -1e301, v); }            /* emnr.c:1169 */
void wo_SetRXAEMNRPosition(wo_channel *c, int position) { *wo_emnr_position(c->emnr) = position; c->bp1.position = position; }
void wo_SetEMNRTables(wo_channel *c, const double *GG, const double *GGS, const double *zeta_hat, const int *zeta_true, double gmin, double gmax,
                      double ximin, double ximax)
{ wo_emnr_set_tables(c->emnr, GG, GGS, zeta_hat, zeta_true, gmin, gmax, ximin, ximax); }

void wo_SetRXAAMSQRun(wo_channel *c, int run) { c->amsq.run = run; }                 /* amsq.c:216-222 */
void wo_SetRXAAMSQThreshold(wo_channel *c, double threshold)                         /* amsq.c:224-232 */
{
    double thresh = pow(10.0, threshold / 20.0);
    c->amsq.tail_thresh = 0.9 * thresh;
    c->amsq.unmute_thresh = thresh;
}
void wo_SetRXAAMSQMaxTail(wo_channel *c, double tail)                                /* amsq.c:234-243 */
{
    if (tail < c->amsq.min_tail) tail = c->amsq.min_tail;
    c->amsq.max_tail = tail;
}

/* SetRXAANFRun ... SetRXAANFPosition (anf.c:175-239) and the ANR twins (anr.c:175-238) */
void wo_SetRXAANFRun(wo_channel *c, int run)
{
    if (c->anf.run != run) {
        bp1_check(c, c->amd.run, run, c->anr.run);
        c->anf.run = run;
        bp1_set(c);
        lms_flush(&c->anf);
    }
}
void wo_SetRXAANRRun(wo_channel *c, int run)
{
    if (c->anr.run != run) {
        bp1_check(c, c->amd.run, c->anf.run, run);
        c->anr.run = run;
        bp1_set(c);
        lms_flush(&c->anr);
    }
}
void wo_SetRXAANFVals(wo_channel *c, int taps, int delay, double gain, double leakage)
{ c->anf.n_taps = taps; c->anf.delay = delay; c->anf.two_mu = gain; c->anf.gamma = leakage; lms_flush(&c->anf); }
void wo_SetRXAANRVals(wo_channel *c, int taps, int delay, double gain, double leakage)
{ c->anr.n_taps = taps; c->anr.delay = delay; c->anr.two_mu = gain; c->anr.gamma = leakage; lms_flush(&c->anr); }
void wo_SetRXAANFPosition(wo_channel *c, int position) { c->anf.position = position; c->bp1.position = position; lms_flush(&c->anf); }
void wo_SetRXAANRPosition(wo_channel *c, int position) { c->anr.position = position; c->bp1.position = position; lms_flush(&c->anr); }

void wo_SetRXAAMDFadeLevel(wo_channel *c, int levelfade) { c->amd.levelfade = levelfade; }

void wo_SetRXAFMDeviation(wo_channel *c, double deviation)
{
    c->fmd.deviation = deviation;
    c->fmd.again = c->fmd.rate / (c->fmd.deviation * WO_TWOPI);
}

void wo_SetRXACTCSSFreq(wo_channel *c, double freq) { c->fmd.ctcss_freq = freq; calc_snotch(c); }
void wo_SetRXACTCSSRun(wo_channel *c, int run) { c->fmd.sntch_run = run; }
void wo_SetRXAFMLimRun(wo_channel *c, int run) { c->fmd.lim_run = run; }                       /* fmd.c:336-347 */
void wo_SetRXAFMLimGain(wo_channel *c, double gaindB)                                           /* fmd.c:349-362 */
{
    double gain = pow(10.0, gaindB / 20.0);
    if (c->fmd.lim_gain != gain) {          /* decalc_fmd + calc_fmd: a new limiter with cleared state */
        wo_agc_free(&c->fmd.plim);
        c->fmd.lim_gain = gain;
        fmd_make_limiter(c);
    }
}
double wo_GetRXAMeter(wo_channel *c, int mt) { return c->meter[mt]; }


/* ---- quisk_wdsp.c:12-69: wdspFexchange0, Quisk's re-blocking shim in front of fexchange0 ----------------------------
 * An arbitrary count of samples scaled to CLIP32 goes into a ring; every full in_size block is handed to fexchange0
 * (scaled to +-1.0) and the output is written back over cSamples, in_size samples per block, scaled to CLIP32 again.
 * Pinned bit for bit against the reference's own quisk_wdsp.c compiled into oracle/_ref (tests/test_oracle_wdsp_shim.py). */
#define WO_CLIP32 2147483647

wo_shim *wo_shim_create(void) { return (wo_shim *)zalloc(sizeof(wo_shim)); }
void wo_shim_destroy(wo_shim *s) { if (s) { free(s->cBuf); free(s); } }

void wo_shim_set_parameter(wo_shim *s, int in_size, int in_use)      /* quisk_wdsp.c:71-91 (channel and fexchange0 are the caller's) */
{
    if (in_size > 0) s->in_size = in_size;
    if (in_use >= 0) s->in_use = in_use;
}

int wo_shim_fexchange0(wo_shim *s, wo_fexchange0_fn fn, void *ctx, double *cSamples, int nSamples)   /* quisk_wdsp.c:24-69 */
{
    int i, error, in_size;
    if (!s->in_use) { s->Windex = 0; s->Rindex = 0; s->nBuf = 0; return nSamples; }
    if (!fn) return nSamples;
    if (nSamples <= 0) return nSamples;
    in_size = s->in_size;
    i = nSamples / in_size + 3;
    if (i * in_size > s->sizeBuf) {
        i *= in_size;
        s->sizeBuf = i;
        s->cBuf = (double *)realloc(s->cBuf, (size_t)i * 2 * sizeof(double));
    }
    for (i = 0; i < nSamples; i++) {
        s->cBuf[2 * s->Windex] = cSamples[2 * i] / WO_CLIP32;            /* complex / int: both parts divided */
        s->cBuf[2 * s->Windex + 1] = cSamples[2 * i + 1] / WO_CLIP32;
        if (++s->Windex >= s->sizeBuf) s->Windex = 0;
    }
    s->nBuf += nSamples;
    nSamples = 0;
    while (s->nBuf >= in_size) {
        fn(ctx, s->cBuf + 2 * s->Rindex, cSamples + 2 * nSamples, &error);
        s->Rindex += in_size;
        if (s->Rindex >= s->sizeBuf) s->Rindex = 0;
        nSamples += in_size;
        s->nBuf -= in_size;
    }
    for (i = 0; i < nSamples; i++) { cSamples[2 * i] *= WO_CLIP32; cSamples[2 * i + 1] *= WO_CLIP32; }
    return nSamples;
}
