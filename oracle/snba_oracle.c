/* snba_oracle.c -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 *
 * CPU restatement of WDSP's SNBA "spectral noise blanker": wdsp/snb.c:31-577 with asolve / median / trI / dR of
 * wdsp/lmath.c:29-186 and the polyphase resampler of wdsp/resample.c:35-157.  The block works on the real part of the
 * signal at 12 kHz: frames of xsize = 256 samples advancing by 64; per frame an order-64 linear predictor is fitted to the
 * frame, the prediction residue marks corrupt samples (median based threshold), and each run of marked samples is replaced
 * by the least-squares interpolation that the predictor implies, best-supported runs first.
 *
 * PARITY UNPINNED: the wdsp sources include <fftw3.h> through comm.h, which this image lacks, so the reference cannot be compiled
 * here.  The arithmetic (order of every sum included) follows the reference line by line so that a later run of the real
 * library can pin it.  Nothing in the product links or calls this file.
 */
#include "snba_oracle.h"
#include "wdsp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SN_MAXIMP 256           /* snb.c:29 */

typedef struct {                /* resample.h, real signals only: SNBA feeds the real part and reads the real part */
    int L, M, ncoef, cpp, idx_in, phnum, in_rate, out_rate;
    double fc_low, fc_in, gain;
    double *h, *ring;
} sn_resample;

static void *sn_zalloc(size_t n) { return calloc(1, n ? n : 1); }

static void sn_resample_calc(sn_resample *a)        /* calc_resample, resample.c:35-79 */
{
    int x = a->in_rate, y = a->out_rate, z, i, j, k, min_rate;
    double fc = a->fc_in, full_rate, hi, lo, *imp;
    while (y != 0) { z = y; y = x % y; x = z; }
    a->L = a->out_rate / x;
    a->M = a->in_rate / x;
    min_rate = a->in_rate < a->out_rate ? a->in_rate : a->out_rate;
    if (fc == 0.0) fc = 0.45 * (double)min_rate;
    full_rate = (double)(a->in_rate * a->L);
    hi = fc / full_rate;
    lo = a->fc_low < 0.0 ? -hi : a->fc_low / full_rate;
    a->ncoef = (int)(140.0 * full_rate / min_rate);
    a->ncoef = (a->ncoef / a->L + 1) * a->L;
    a->cpp = a->ncoef / a->L;
    free(a->h); free(a->ring);
    a->h = (double *)sn_zalloc((size_t)a->ncoef * sizeof(double));
    imp = wo_fir_bandpass(a->ncoef, lo, hi, 1.0, 1, 0, a->gain * (double)a->L);
    i = 0;
    for (j = 0; j < a->L; j++)
        for (k = 0; k < a->ncoef; k += a->L) a->h[i++] = imp[j + k];
    free(imp);
    a->ring = (double *)sn_zalloc((size_t)a->cpp * sizeof(double));
    a->idx_in = a->cpp - 1;
    a->phnum = 0;
}

static int sn_resample_exec(sn_resample *a, const double *in, int in_step, int size, double *out, int out_step)  /* xresample, resample.c:120-157 */
{
    int n_out = 0, i, j, n, idx;
    for (i = 0; i < size; i++) {
        a->ring[a->idx_in] = in[(size_t)i * in_step];
        while (a->phnum < a->L) {
            double acc = 0.0;
            n = a->cpp * a->phnum;
            for (j = 0; j < a->cpp; j++) {
                if ((idx = a->idx_in + j) >= a->cpp) idx -= a->cpp;
                acc += a->h[n + j] * a->ring[idx];
            }
            out[(size_t)n_out * out_step] = acc;
            n_out++;
            a->phnum += a->M;
        }
        a->phnum -= a->L;
        if (--a->idx_in < 0) a->idx_in = a->cpp - 1;
    }
    return n_out;
}

struct wo_snba {
    int run, inrate, internalrate, bsize, ovrlp, xsize, asize, npasses, b, pre, post;
    double k1, k2, pmultmin, out_low_cut, out_high_cut;
    int isize, incr, iasize, iainidx, iaoutidx, nsamps, oasize, oainidx, oaoutidx, init_oaoutidx;
    sn_resample rin, rout;
    double *inbuf, *outbuf, *inaccum, *outaccum, *xbase, *xaux;
    double *a, *v, *savex, *xhout, *vp, *vpwr;
    int *detout, *unfixed;
    double *w_r, *w_atai, *w_a1, *w_a2, *w_p1, *w_p2, *w_y, *w_v, *w_z;
};

static void sn_calc(wo_snba *d)         /* calc_snba, snb.c:31-66 */
{
    if (d->inrate >= d->internalrate) d->isize = d->bsize / (d->inrate / d->internalrate);
    else d->isize = d->bsize * (d->internalrate / d->inrate);
    d->inbuf = (double *)sn_zalloc((size_t)d->isize * sizeof(double));
    d->outbuf = (double *)sn_zalloc((size_t)d->isize * sizeof(double));
    d->rin.in_rate = d->inrate; d->rin.out_rate = d->internalrate; d->rin.fc_in = 0.0; d->rin.gain = 2.0; d->rin.fc_low = 250.0;
    sn_resample_calc(&d->rin);
    d->rout.in_rate = d->internalrate; d->rout.out_rate = d->inrate; d->rout.fc_in = 0.0; d->rout.gain = 2.0; d->rout.fc_low = 200.0;
    sn_resample_calc(&d->rout);
    d->incr = d->xsize / d->ovrlp;
    d->iasize = d->incr > d->isize ? d->incr : d->isize;
    d->iainidx = d->iaoutidx = 0;
    d->inaccum = (double *)sn_zalloc((size_t)d->iasize * sizeof(double));
    d->nsamps = 0;
    if (d->incr > d->isize) { d->oasize = d->incr; d->oainidx = 0; d->oaoutidx = d->isize; }
    else { d->oasize = d->isize; d->oainidx = 0; d->oaoutidx = 0; }
    d->init_oaoutidx = d->oaoutidx;
    d->outaccum = (double *)sn_zalloc((size_t)d->oasize * sizeof(double));
}

wo_snba *wo_snba_create(int rate, int bsize)
{
    wo_snba *d = (wo_snba *)sn_zalloc(sizeof(*d));
    const int xs = 256;
    d->inrate = rate; d->internalrate = 12000; d->bsize = bsize; d->ovrlp = 4; d->xsize = xs; d->asize = 64; d->npasses = 2;
    d->k1 = 8.0; d->k2 = 20.0; d->b = 10; d->pre = 2; d->post = 2; d->pmultmin = 0.5;
    d->out_low_cut = 200.0; d->out_high_cut = 5400.0;
    sn_calc(d);
    d->xbase = (double *)sn_zalloc((size_t)2 * xs * sizeof(double));
    d->xaux = d->xbase + xs;
    d->a = (double *)sn_zalloc(xs * sizeof(double));
    d->v = (double *)sn_zalloc(xs * sizeof(double));
    d->savex = (double *)sn_zalloc(xs * sizeof(double));
    d->xhout = (double *)sn_zalloc(xs * sizeof(double));
    d->vp = (double *)sn_zalloc(xs * sizeof(double));
    d->vpwr = (double *)sn_zalloc(xs * sizeof(double));
    d->detout = (int *)sn_zalloc(xs * sizeof(int));
    d->unfixed = (int *)sn_zalloc(xs * sizeof(int));
    {
        const size_t a1r = (size_t)xs + d->asize, a2c = (size_t)xs + 2 * d->asize;     /* snb.c:103-112 */
        d->w_r = (double *)sn_zalloc(xs * sizeof(double));
        d->w_atai = (double *)sn_zalloc((size_t)xs * xs * sizeof(double));
        d->w_a1 = (double *)sn_zalloc(a1r * xs * sizeof(double));
        d->w_a2 = (double *)sn_zalloc(a1r * a2c * sizeof(double));
        d->w_p1 = (double *)sn_zalloc(xs * a2c * sizeof(double));
        d->w_p2 = (double *)sn_zalloc(xs * sizeof(double));
        d->w_y = (double *)sn_zalloc(xs * sizeof(double));
        d->w_v = (double *)sn_zalloc(xs * sizeof(double));
        d->w_z = (double *)sn_zalloc(xs * sizeof(double));
    }
    return d;
}

void wo_snba_free(wo_snba *d)
{
    if (!d) return;
    free(d->inbuf); free(d->outbuf); free(d->inaccum); free(d->outaccum); free(d->xbase); free(d->a); free(d->v); free(d->savex);
    free(d->xhout); free(d->vp); free(d->vpwr); free(d->detout); free(d->unfixed); free(d->w_r); free(d->w_atai); free(d->w_a1);
    free(d->w_a2); free(d->w_p1); free(d->w_p2); free(d->w_y); free(d->w_v); free(d->w_z);
    free(d->rin.h); free(d->rin.ring); free(d->rout.h); free(d->rout.ring);
    free(d);
}

int *wo_snba_run(wo_snba *d) { return &d->run; }
int wo_snba_xsize(const wo_snba *d) { return d->xsize; }

void wo_snba_flush(wo_snba *d)          /* flush_snba, snb.c:161-185 */
{
    const int xs = d->xsize;
    d->iainidx = 0; d->iaoutidx = 0; d->nsamps = 0; d->oainidx = 0; d->oaoutidx = d->init_oaoutidx;
    memset(d->inaccum, 0, (size_t)d->iasize * sizeof(double));
    memset(d->outaccum, 0, (size_t)d->oasize * sizeof(double));
    memset(d->xaux, 0, (size_t)xs * sizeof(double));      /* only the frame half; the history half in front of it is kept */
    memset(d->a, 0, xs * sizeof(double)); memset(d->v, 0, xs * sizeof(double));
    memset(d->detout, 0, xs * sizeof(int)); memset(d->savex, 0, xs * sizeof(double));
    memset(d->xhout, 0, xs * sizeof(double)); memset(d->unfixed, 0, xs * sizeof(int));
    memset(d->vp, 0, xs * sizeof(double)); memset(d->vpwr, 0, xs * sizeof(double));
    memset(d->inbuf, 0, (size_t)d->isize * sizeof(double)); memset(d->outbuf, 0, (size_t)d->isize * sizeof(double));
    memset(d->rin.ring, 0, (size_t)d->rin.cpp * sizeof(double)); d->rin.idx_in = d->rin.cpp - 1; d->rin.phnum = 0;
    memset(d->rout.ring, 0, (size_t)d->rout.cpp * sizeof(double)); d->rout.idx_in = d->rout.cpp - 1; d->rout.phnum = 0;
}

void wo_snba_set_output_bandwidth(wo_snba *d, double flow, double fhigh)    /* snb.c:660-694 */
{
    double f_low = d->rout.fc_low, f_high = d->rout.fc_in;     /* (the reference leaves them unset when the signs are mixed the other way) */
    const double lc = d->out_low_cut, hc = d->out_high_cut;
    if (flow >= 0 && fhigh >= 0) {
        if (fhigh < lc) fhigh = lc;
        if (flow > hc) flow = hc;
        f_low = lc > flow ? lc : flow;
        f_high = hc < fhigh ? hc : fhigh;
    } else if (flow <= 0 && fhigh <= 0) {
        if (flow > -lc) flow = -lc;
        if (fhigh < -hc) fhigh = -hc;
        f_low = lc > -fhigh ? lc : -fhigh;
        f_high = hc < -flow ? hc : -flow;
    } else if (flow < 0 && fhigh > 0) {
        double absmax = -flow > fhigh ? -flow : fhigh;
        if (absmax < lc) absmax = lc;
        f_low = lc;
        f_high = hc < absmax ? hc : absmax;
    }
    if (f_low != d->rout.fc_low || f_high != d->rout.fc_in) {       /* setBandwidth_resample, resample.c:195-204 */
        d->rout.fc_low = f_low;
        d->rout.fc_in = f_high;
        sn_resample_calc(&d->rout);
    }
}

/* ---- lmath.c */
void wo_snba_asolve(int xsize, int asize, const double *x, double *a)   /* lmath.c:96-127 */
{
    double r[SN_MAXIMP + 1], z[SN_MAXIMP + 1], beta, alpha, t;
    int i, j, k;
    memset(r, 0, sizeof(r)); memset(z, 0, sizeof(z));
    for (i = 0; i <= asize; i++)
        for (j = 0; j < xsize; j++) r[i] += x[j] * x[j - i];
    z[0] = 1.0;
    beta = r[0];
    for (k = 0; k < asize; k++) {
        alpha = 0.0;
        for (j = 0; j <= k; j++) alpha -= z[j] * r[k + 1 - j];
        alpha /= beta;
        for (i = 0; i <= (k + 1) / 2; i++) {
            t = z[k + 1 - i] + alpha * z[i];
            z[i] = z[i] + alpha * z[k + 1 - i];
            z[k + 1 - i] = t;
        }
        beta *= 1.0 - alpha * alpha;
    }
    for (i = 0; i < asize; i++) {
        a[i] = -z[i + 1];
        if (a[i] != a[i]) a[i] = 0.0;
    }
}

void wo_snba_median(int n, double *a, double *med)      /* lmath.c:129-186: quickselect of the element of rank n/2 */
{
    int lo = 0, hi = n - 1, k = n / 2, i, j, m;
    double x, t;
#define SN_SWAP(p, q) do { t = a[p]; a[p] = a[q]; a[q] = t; } while (0)
    while (hi > lo + 1) {
        m = (lo + hi) / 2;
        SN_SWAP(m, lo + 1);
        if (a[lo] > a[hi]) SN_SWAP(lo, hi);
        if (a[lo + 1] > a[hi]) SN_SWAP(lo + 1, hi);
        if (a[lo] > a[lo + 1]) SN_SWAP(lo, lo + 1);
        i = lo + 1; j = hi; x = a[lo + 1];
        do i++; while (a[i] < x);
        do j--; while (a[j] > x);
        while (j >= i) {
            SN_SWAP(i, j);
            do i++; while (a[i] < x);
            do j--; while (a[j] > x);
        }
        a[lo + 1] = a[j];
        a[j] = x;
        if (j >= k) hi = j - 1;
        if (j <= k) lo = i;
    }
    if (hi == lo + 1 && a[hi] < a[lo]) SN_SWAP(lo, hi);
#undef SN_SWAP
    *med = a[k];
}

static void sn_dR(int n, double *r, double *y, double *z)      /* dR, lmath.c:29-50: Durbin's recursion for the Yule-Walker system */
{
    int i, j, k;
    double alpha, beta, gamma;
    memset(z, 0, (size_t)(n - 1) * sizeof(double));
    y[0] = -r[1];
    alpha = -r[1];
    beta = 1.0;
    for (k = 0; k < n - 1; k++) {
        beta *= 1.0 - alpha * alpha;
        gamma = 0.0;
        for (i = k + 1, j = 0; i > 0; i--, j++) gamma += r[i] * y[j];
        alpha = -(r[k + 2] + gamma) / beta;
        for (i = 0, j = k; i <= k; i++, j--) z[i] = y[i] + alpha * y[j];
        memcpy(y, z, (size_t)(k + 1) * sizeof(double));
        y[k + 1] = alpha;
    }
}

static void sn_trI(int n, double *r, double *B, double *y, double *v, double *z)   /* trI, lmath.c:52-94: inverse of a symmetric Toeplitz matrix */
{
    int i, j, ni, nj;
    double gamma, t, scale, b;
    memset(y, 0, (size_t)(n - 1) * sizeof(double));
    memset(v, 0, (size_t)(n - 1) * sizeof(double));
    scale = 1.0 / r[0];
    for (i = 0; i < n; i++) r[i] *= scale;
    sn_dR(n - 1, r, y, z);
    t = 0.0;
    for (i = 0; i < n - 1; i++) t += r[i + 1] * y[i];
    gamma = 1.0 / (1.0 + t);
    for (i = 0, j = n - 2; i < n - 1; i++, j--) v[i] = gamma * y[j];
    B[0] = gamma;
    for (i = 1, j = n - 2; i < n; i++, j--) B[i] = v[j];
    for (i = 1; i <= (n - 1) / 2; i++)
        for (j = i; j < n - i; j++)
            B[i * n + j] = B[(i - 1) * n + (j - 1)] + (v[n - j - 1] * v[n - i - 1] - v[i - 1] * v[j - 1]) / gamma;
    for (i = 0; i <= (n - 1) / 2; i++)
        for (j = i; j < n - i; j++) {
            b = B[i * n + j] *= scale;
            B[j * n + i] = b;
            ni = n - i - 1; nj = n - j - 1;
            B[ni * n + nj] = b;
            B[nj * n + ni] = b;
        }
}

/* ---- snb.c:209-304: the least-squares replacement of xusize samples given asize good samples either side */
static void sn_xhat(wo_snba *d, int xusize, int asize, const double *xk, const double *a, double *xout)
{
    const int a1rows = xusize + asize, a2cols = xusize + 2 * asize;
    double *r = d->w_r, *ATAI = d->w_atai, *A1 = d->w_a1, *A2 = d->w_a2, *P1 = d->w_p1, *P2 = d->w_p2;
    int i, j, k;
    memset(r, 0, (size_t)xusize * sizeof(double));
    memset(ATAI, 0, (size_t)xusize * xusize * sizeof(double));
    memset(A1, 0, (size_t)a1rows * xusize * sizeof(double));
    memset(A2, 0, (size_t)a1rows * a2cols * sizeof(double));
    memset(P1, 0, (size_t)xusize * a2cols * sizeof(double));
    memset(P2, 0, (size_t)xusize * sizeof(double));
    for (i = 0; i < xusize; i++) {                              /* snb.c:280-286 */
        A1[i * xusize + i] = 1.0;
        k = i + 1;
        for (j = k; j < k + asize; j++) A1[j * xusize + i] = -a[j - k];
    }
    for (i = 0; i < asize; i++)                                 /* snb.c:288-292 */
        for (k = asize - i - 1, j = 0; k < asize; k++, j++) A2[j * a2cols + i] = a[k];
    for (i = asize + xusize; i < 2 * asize + xusize; i++) {     /* snb.c:293-298 */
        A2[(i - asize) * a2cols + i] = -1.0;
        for (j = i - asize + 1, k = 0; j < xusize + asize; j++, k++) A2[j * a2cols + i] = a[k];
    }
    for (i = 0; i < xusize; i++)                                /* ATAc0, snb.c:209-216 */
        for (j = 0; j < a1rows; j++) r[i] += A1[j * xusize + i] * A1[j * xusize + 0];
    sn_trI(xusize, r, ATAI, d->w_y, d->w_v, d->w_z);
    {                                                           /* multA1TA2, snb.c:218-239 */
        const int m = xusize, n = a2cols, q = a1rows, p = q - m;
        for (i = 0; i < m; i++)
            for (j = 0; j < n; j++) {
                if (j < p) {
                    int kmax = i + p < j ? i + p : j;
                    for (k = i; k <= kmax; k++) P1[i * n + j] += A1[k * m + i] * A2[k * n + j];
                }
                if (j >= n - p) {
                    int kmin = i > q - (n - j) ? i : q - (n - j);
                    for (k = kmin; k <= i + p; k++) P1[i * n + j] += A1[k * m + i] * A2[k * n + j];
                }
            }
    }
    {                                                           /* multXKE, snb.c:241-252 */
        const int m = xusize, q = a2cols, p = asize;
        for (i = 0; i < m; i++) {
            for (k = i; k < p; k++) P2[i] += P1[i * q + k] * xk[k];
            for (k = q - p; k <= q - m + i; k++) P2[i] += P1[i * q + k] * xk[k];
        }
    }
    for (i = 0; i < xusize; i++) {                              /* multAv, snb.c:254-263 */
        xout[i] = 0.0;
        for (k = 0; k < xusize; k++) xout[i] += ATAI[i * xusize + k] * P2[k];
    }
}

static void sn_invf(int xsize, int asize, const double *a, const double *x, double *v)     /* invf, snb.c:306-322 */
{
    int i, j;
    memset(v, 0, (size_t)xsize * sizeof(double));
    for (i = asize; i < xsize - asize; i++) {
        for (j = 0; j < asize; j++) v[i] += a[j] * (x[i - 1 - j] + x[i + 1 + j]);
        v[i] = x[i] - 0.5 * v[i];
    }
    for (i = xsize - asize; i < xsize; i++) {
        for (j = 0; j < asize; j++) v[i] += a[j] * x[i - 1 - j];
        v[i] = x[i] - v[i];
    }
}

static void sn_det(wo_snba *d, int asize, const double *v, int *detout)        /* det, snb.c:324-402 */
{
    const int xs = d->xsize;
    int i, j, bstate = 0, bcount = 0, bsamp = 0;
    double medpwr, t1, t2 = 0.0;
    for (i = asize, j = 0; i < xs; i++, j++) { d->vpwr[i] = v[i] * v[i]; d->vp[j] = d->vpwr[i]; }
    wo_snba_median(xs - asize, d->vp, &medpwr);
    t1 = d->k1 * medpwr;
    for (i = asize; i < xs; i++) {
        if (d->vpwr[i] <= t1) t2 += d->vpwr[i];
        else if (d->vpwr[i] <= 2.0 * t1) t2 += 2.0 * t1 - d->vpwr[i];
    }
    t2 *= d->k2 / (double)(xs - asize);
    for (i = asize; i < xs; i++) detout[i] = d->vpwr[i] > t2 ? 1 : 0;
    for (i = asize; i < xs; i++) {          /* bridge gaps of up to b clean samples between detections */
        switch (bstate) {
        case 0: if (detout[i] == 1) bstate = 1; break;
        case 1: if (detout[i] == 0) { bstate = 2; bsamp = i; bcount = 1; } break;
        case 2:
            ++bcount;
            if (bcount > d->b) bstate = detout[i] == 1 ? 1 : 0;
            else if (detout[i] == 1) {
                for (j = bsamp; j < bsamp + bcount - 1; j++) detout[j] = 1;
                bstate = 1;
            }
            break;
        }
    }
    for (i = asize; i < xs; i++)
        if (detout[i] == 1)
            for (j = i - 1; j > i - 1 - d->pre; j--) if (j >= asize) detout[j] = 1;
    for (i = xs - 1; i >= asize; i--)
        if (detout[i] == 1)
            for (j = i + 1; j < i + 1 + d->post; j++) if (j < xs) detout[j] = 1;
}

static int sn_scan(int xsize, int pval, double pmultmin, const int *det, int *bimp, int *limp, int *befimp, int *aftimp,
                   int *p_opt, int *next)      /* scanFrame, snb.c:404-490 */
{
    int inflag = 0, i = 0, j, k, nimp = 0, ti, nextlist[SN_MAXIMP];
    double td, merit[SN_MAXIMP] = { 0 };
    memset(befimp, 0, SN_MAXIMP * sizeof(int));
    memset(aftimp, 0, SN_MAXIMP * sizeof(int));
    while (i < xsize && nimp < SN_MAXIMP) {
        if (det[i] == 1 && inflag == 0) { inflag = 1; bimp[nimp] = i; limp[nimp] = 1; nimp++; }
        else if (det[i] == 1) limp[nimp - 1]++;
        else {
            inflag = 0;
            befimp[nimp]++;
            if (nimp > 0) aftimp[nimp - 1]++;
        }
        i++;
    }
    for (i = 0; i < nimp; i++) {
        p_opt[i] = befimp[i] < aftimp[i] ? befimp[i] : aftimp[i];
        if (p_opt[i] > pval) p_opt[i] = pval;
        if (p_opt[i] < (int)(pmultmin * limp[i])) p_opt[i] = -1;
    }
    for (i = 0; i < nimp; i++) { merit[i] = (double)p_opt[i] / (double)limp[i]; nextlist[i] = i; }
    for (j = 0; j < nimp - 1; j++)
        for (k = 0; k < nimp - j - 1; k++)
            if (merit[k] < merit[k + 1]) {
                td = merit[k]; ti = nextlist[k];
                merit[k] = merit[k + 1]; nextlist[k] = nextlist[k + 1];
                merit[k + 1] = td; nextlist[k + 1] = ti;
            }
    i = 1;
    if (nimp > 0) while (i < nimp && merit[i] == merit[0]) i++;
    for (j = 0; j < i - 1; j++)
        for (k = 0; k < i - j - 1; k++)
            if (limp[nextlist[k]] < limp[nextlist[k + 1]]) {
                td = merit[k]; ti = nextlist[k];
                merit[k] = merit[k + 1]; nextlist[k] = nextlist[k + 1];
                merit[k + 1] = td; nextlist[k + 1] = ti;
            }
    *next = nextlist[0];        /* (indeterminate in the reference when nimp == 0; never used then) */
    return nimp;
}

void wo_snba_frame(wo_snba *d, double *x)       /* execFrame, snb.c:492-537 */
{
    const int xs = d->xsize;
    int i, k, pass, nimp, next = 0, p;
    int bimp[SN_MAXIMP], limp[SN_MAXIMP], befimp[SN_MAXIMP + 1], aftimp[SN_MAXIMP], p_opt[SN_MAXIMP];
    memcpy(d->savex, x, xs * sizeof(double));
    wo_snba_asolve(xs, d->asize, x, d->a);
    sn_invf(xs, d->asize, d->a, x, d->v);
    sn_det(d, d->asize, d->v, d->detout);
    for (i = 0; i < xs; i++) if (d->detout[i] != 0) x[i] = 0.0;
    nimp = sn_scan(xs, d->asize, d->pmultmin, d->detout, bimp, limp, befimp, aftimp, p_opt, &next);
    for (pass = 0; pass < d->npasses; pass++) {
        memcpy(d->unfixed, d->detout, xs * sizeof(int));
        for (k = 0; k < nimp; k++) {
            if (k > 0) sn_scan(xs, d->asize, d->pmultmin, d->unfixed, bimp, limp, befimp, aftimp, p_opt, &next);
            if ((p = p_opt[next]) > 0) {
                wo_snba_asolve(xs, p, x, d->a);
                sn_xhat(d, limp[next], p, &x[bimp[next] - p], d->a, d->xhout);
                memcpy(&x[bimp[next]], d->xhout, (size_t)limp[next] * sizeof(double));
                memset(&d->unfixed[bimp[next]], 0, (size_t)limp[next] * sizeof(int));
            } else
                memcpy(&x[bimp[next]], &d->savex[bimp[next]], (size_t)limp[next] * sizeof(double));
        }
    }
}

void wo_snba_exec(wo_snba *d, double *buf)      /* xsnba, snb.c:539-571 (in == out: nothing to do when off) */
{
    int i;
    if (!d->run) return;
    sn_resample_exec(&d->rin, buf, 2, d->bsize, d->inbuf, 1);
    for (i = 0; i < d->isize; i++) {
        d->inaccum[d->iainidx] = d->inbuf[i];
        d->iainidx = (d->iainidx + 1) % d->iasize;
    }
    d->nsamps += d->isize;
    while (d->nsamps >= d->incr) {
        memcpy(&d->xaux[d->xsize - d->incr], &d->inaccum[d->iaoutidx], (size_t)d->incr * sizeof(double));
        wo_snba_frame(d, d->xaux);
        d->iaoutidx = (d->iaoutidx + d->incr) % d->iasize;
        d->nsamps -= d->incr;
        memcpy(&d->outaccum[d->oainidx], d->xaux, (size_t)d->incr * sizeof(double));
        d->oainidx = (d->oainidx + d->incr) % d->oasize;
        memmove(d->xbase, &d->xbase[d->incr], (size_t)(2 * d->xsize - d->incr) * sizeof(double));
    }
    for (i = 0; i < d->isize; i++) {
        d->outbuf[i] = d->outaccum[d->oaoutidx];
        d->oaoutidx = (d->oaoutidx + 1) % d->oasize;
    }
    sn_resample_exec(&d->rout, d->outbuf, 1, d->isize, buf, 2);
    for (i = 0; i < d->bsize; i++) buf[2 * i + 1] = 0.0;       /* the imaginary part of outbuff is zero, snb.c:565 */
}

void wo_snba_set_ovrlp(wo_snba *d, int ovrlp)                    /* SetRXASNBAovrlp, snb.c:595-603: decalc_snba + calc_snba */
{
    free(d->rout.h); free(d->rout.ring); d->rout.h = d->rout.ring = NULL;      /* decalc_snba, snb.c:121-129 */
    free(d->rin.h); free(d->rin.ring); d->rin.h = d->rin.ring = NULL;
    free(d->outbuf); free(d->inbuf); free(d->outaccum); free(d->inaccum);
    d->ovrlp = ovrlp;
    sn_calc(d);
}

void wo_snba_set_tuning(wo_snba *d, int which, double v)        /* snb.c:604-658 */
{
    switch (which) {
    case 0: d->asize = (int)v; break;
    case 1: d->npasses = (int)v; break;
    case 2: d->k1 = v; break;
    case 3: d->k2 = v; break;
    case 4: d->b = (int)v; break;
    case 5: d->pre = (int)v; break;
    case 6: d->post = (int)v; break;
    default: d->pmultmin = v; break;
    }
}
