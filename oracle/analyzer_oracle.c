/* analyzer_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * Restatement of WDSP's display engine, wdsp/analyzer.c: Spectrum0 / Spectrum2 / Spectrum feed a sample ring per sub-span,
 * a frame of `size` samples is windowed and transformed every `size - overlap` samples, the power spectrum is clipped /
 * re-ordered / stitched into one span, reduced to pixels by one of five detectors (or interpolated when there are more
 * pixels than bins), averaged in one of five ways, converted to dB and handed out by GetPixels.
 *
 * PARITY UNPINNED by reference execution: analyzer.c needs <fftw3.h> (wdsp/comm.h:55) and the Windows thread API shims;
 * neither FFTW3 nor a way to build it is in this image.  Checked against numpy recomputation in
 * tests/test_oracle_analyzer.py.
 *
 * The reference runs a dispatcher thread (sendbuf, analyzer.c:884-917) that polls the rings and queues one worker per
 * (sub-span, LO) whose ring holds a whole frame; the last worker of a set stitches and publishes.  This restatement runs the
 * same steps synchronously inside the Spectrum* call, in the dispatcher's scan order -- what the reference computes when its
 * threads keep up with the input.
 */
#include "analyzer_oracle.h"
#include "fft_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AO_PI 3.1415926535897932

double wo_mlog10_value(double v);       /* wdsp_oracle.c */

struct ao_disp {
    int max_size, max_stitch;
    int num_pixout, type, size, out_size, window_type, overlap, clip, num_stitch, num_pixels, buff_size, incr, bsize;
    int flip, begin_ss, end_ss, fscL, fscH, max_writeahead, cal_set, cal_changed, sample_rate;
    double PiAlpha, fsclipL, fsclipH, pix_per_bin, bin_per_pix, det_offset, scale, f_min, f_max;
    double inv_coherent_gain, inherent_power_gain, inv_enb, norm_oneHz;
    double *window;
    float *I[AO_MAX_STITCH], *Q[AO_MAX_STITCH];             /* dINREAL is float outside Thetis, wdsp/comm.h:128-132 */
    int in_idx[AO_MAX_STITCH], out_idx[AO_MAX_STITCH], have[AO_MAX_STITCH], ready[AO_MAX_STITCH], busy[AO_MAX_STITCH];
    unsigned stitch_flag;
    double *result[AO_MAX_STITCH];
    int ss_bins[AO_MAX_STITCH];
    double *fft;                                            /* interleaved work vector */
    double *snap_buff; int snap_ss, snap_taken;             /* SnapSpectrum, analyzer.c:1337-1346 */
    double *pre_av_out;
    int det_type[AO_MAX_PIXOUTS], av_mode[AO_MAX_PIXOUTS], num_average[AO_MAX_PIXOUTS], normalize[AO_MAX_PIXOUTS];
    int avail_frames[AO_MAX_PIXOUTS], av_in_idx[AO_MAX_PIXOUTS], av_out_idx[AO_MAX_PIXOUTS];
    double av_backmult[AO_MAX_PIXOUTS];
    double *av_sum[AO_MAX_PIXOUTS], *t_pixels[AO_MAX_PIXOUTS], *av_buff[AO_MAX_PIXOUTS][AO_MAX_AVERAGE];
    float *pixels[AO_MAX_PIXOUTS][AO_NUM_PIXEL_BUFFS];
    int w_pix[AO_MAX_PIXOUTS], r_pix[AO_MAX_PIXOUTS], last_pix[AO_MAX_PIXOUTS], pb_ready[AO_MAX_PIXOUTS][AO_NUM_PIXEL_BUFFS];
    double *cd;
    int n_freqs[AO_MAX_CAL_SETS];
    double *freqs[AO_MAX_CAL_SETS], *ac3[AO_MAX_CAL_SETS], *ac2[AO_MAX_CAL_SETS], *ac1[AO_MAX_CAL_SETS], *ac0[AO_MAX_CAL_SETS];
    long frames;
};

static void *zalloc(size_t n) { return calloc(n ? n : 1, 1); }

/* modified Bessel function I0, the polynomial pair of analyzer.c:33-50 (Abramowitz & Stegun 9.8.1 / 9.8.2) */
static double ao_bessi0(double x)
{
    static const double small[7] = {1.0, 3.5156229, 3.0899424, 1.2067492, 0.2659732, 0.360768e-1, 0.45813e-2};
    static const double large[9] = {0.39894228, 0.1328592e-1, 0.225319e-2, -0.157565e-2, 0.916281e-2, -0.2057706e-1,
                                    0.2635537e-1, -0.1647633e-1, 0.392377e-2};
    const double ax = fabs(x);
    double y, p;
    int k;
    if (ax < 3.75) {
        y = x / 3.75; y = y * y;
        p = small[6];
        for (k = 5; k >= 0; k--) p = small[k] + y * p;
        return p;
    }
    y = 3.75 / ax;
    p = large[8];
    for (k = 7; k >= 0; k--) p = large[k] + y * p;
    return (exp(ax) / sqrt(ax)) * p;
}

/* new_window, analyzer.c:52-176: the window is scaled to unit coherent gain; the equivalent noise bandwidth follows */
void ao_window(int type, int size, double PiAlpha, double *w, double *inv_coherent_gain, double *inherent_power_gain, double *inv_enb)
{
    const double step = 2.0 * AO_PI / ((double)size - 1.0);
    double cg = 0.0, ig = 0.0, icg = 1.0;
    int i;
    for (i = 0; i < size; i++) {
        const double a = step * (double)i;
        double v = 1.0;
        switch (type) {
        case 1: v = 0.35875 - 0.48829 * cos(a) + 0.14128 * cos(2.0 * a) - 0.01168 * cos(3.0 * a); break;
        case 2: v = 0.5 * (1.0 - cos((double)i * step)); break;
        case 3: v = 0.21557895 - 0.41663158 * cos(a) + 0.277263158 * cos(2.0 * a) - 0.083578947 * cos(3.0 * a) + 0.006947368 * cos(4.0 * a); break;
        case 4: v = 0.54 - 0.46 * cos((double)i * step); break;
        case 5: v = ao_bessi0(PiAlpha * sqrt(1.0 - pow(2.0 * (double)i / (double)(size - 1) - 1.0, 2))) / ao_bessi0(PiAlpha); break;
        case 6: {
            const double c = cos(a);
            v = 6.3964424114390378e-02 + c * (-2.3993864599352804e-01 + c * (3.5015956323820469e-01 + c * (-2.4774111897080783e-01
                + c * (8.5438256055858031e-02 + c * (-1.2320203369293225e-02 + c * 4.3778825791773474e-04)))));
            break; }
        default: break;
        }
        w[i] = v;
        cg += v;
        ig += v * v;
    }
    if (type == 0) ig = (double)size;
    else {
        icg = (double)size / cg;
        for (i = 0; i < size; i++) w[i] *= icg;
    }
    *inv_coherent_gain = icg;
    *inherent_power_gain = ig / (double)size;
    *inv_enb = 1.0 / (*inherent_power_gain * icg * icg);
}

ao_disp *ao_create(int max_size, int max_stitch)          /* XCreateAnalyzer, analyzer.c:1140-1236 (max_num_fft is 1, comm.h:125) */
{
    ao_disp *a = (ao_disp *)zalloc(sizeof(ao_disp));
    int i, j;
    a->max_size = max_size; a->max_stitch = max_stitch;
    a->window = (double *)zalloc(sizeof(double) * max_size);
    a->fft = (double *)zalloc(sizeof(double) * 2 * max_size);
    a->pre_av_out = (double *)zalloc(sizeof(double) * max_size * max_stitch);
    a->bsize = max_size * AO_SAMP_BUFF_MULT;
    for (i = 0; i < max_stitch; i++) {
        a->result[i] = (double *)zalloc(sizeof(double) * max_size);
        a->I[i] = (float *)zalloc(sizeof(float) * a->bsize);
        a->Q[i] = (float *)zalloc(sizeof(float) * a->bsize);
    }
    for (i = 0; i < AO_MAX_PIXOUTS; i++) {
        a->av_sum[i] = (double *)zalloc(sizeof(double) * AO_MAX_PIXELS);
        a->t_pixels[i] = (double *)zalloc(sizeof(double) * AO_MAX_PIXELS);
        for (j = 0; j < AO_MAX_AVERAGE; j++) a->av_buff[i][j] = (double *)zalloc(sizeof(double) * AO_MAX_PIXELS);
        for (j = 0; j < AO_NUM_PIXEL_BUFFS; j++) a->pixels[i][j] = (float *)zalloc(sizeof(float) * AO_MAX_PIXELS);
    }
    a->cd = (double *)zalloc(sizeof(double) * AO_MAX_PIXELS);
    for (j = 0; j < AO_MAX_PIXELS; j++) a->cd[j] = 1.0;
    for (i = 0; i < AO_MAX_CAL_SETS; i++) {
        a->freqs[i] = (double *)zalloc(sizeof(double) * AO_MAX_N);
        a->ac3[i] = (double *)zalloc(sizeof(double) * AO_MAX_N); a->ac2[i] = (double *)zalloc(sizeof(double) * AO_MAX_N);
        a->ac1[i] = (double *)zalloc(sizeof(double) * AO_MAX_N); a->ac0[i] = (double *)zalloc(sizeof(double) * AO_MAX_N);
    }
    a->size = -1; a->window_type = -1; a->num_pixels = -1; a->cal_set = -1; a->f_min = -1.0; a->f_max = -1.0;
    return a;
}

void ao_destroy(ao_disp *a)
{
    int i, j;
    if (!a) return;
    free(a->window); free(a->fft); free(a->pre_av_out); free(a->cd);
    for (i = 0; i < a->max_stitch; i++) { free(a->result[i]); free(a->I[i]); free(a->Q[i]); }
    for (i = 0; i < AO_MAX_PIXOUTS; i++) {
        free(a->av_sum[i]); free(a->t_pixels[i]);
        for (j = 0; j < AO_MAX_AVERAGE; j++) free(a->av_buff[i][j]);
        for (j = 0; j < AO_NUM_PIXEL_BUFFS; j++) free(a->pixels[i][j]);
    }
    for (i = 0; i < AO_MAX_CAL_SETS; i++) { free(a->freqs[i]); free(a->ac3[i]); free(a->ac2[i]); free(a->ac1[i]); free(a->ac0[i]); }
    free(a);
}

/* interpolate, analyzer.c:747-798: the calibration magnitude at every pixel's frequency from the cubic pieces, squared */
static void ao_interpolate(ao_disp *a, int set, double fmin, double fmax, int num_pixels)
{
    const int n = a->n_freqs[set];
    const double *fr = a->freqs[set];
    int i, k = 0, kmin = 0, kmax = n - 1;
    for (i = 0; i < num_pixels; i++) {
        const double f = fmin + (double)i * (fmax - fmin) / (double)(num_pixels - 1);
        double dx, mag;
        if (f < fr[0]) k = 0;
        else if (f > fr[n - 1]) k = n - 2;
        else {
            int kdelta = 1;
            while (f < fr[kmin]) { kmin = kmin - kdelta > 0 ? kmin - kdelta : 0; kdelta += kdelta; }
            while (f > fr[kmax]) { kmax = kmax + kdelta < n - 1 ? kmax + kdelta : n - 1; kdelta += kdelta; }
            while (kmax - kmin > 1) {
                k = (kmin + kmax) / 2;
                if (f > fr[k]) kmin = k; else kmax = k--;
            }
        }
        dx = f - fr[k];
        mag = ((a->ac3[set][k] * dx + a->ac2[set][k]) * dx + a->ac1[set][k]) * dx + a->ac0[set][k];
        a->cd[i] = mag * mag;
    }
}

/* SetCalibration + build_interpolants, analyzer.c:800-882,1380-1410 (dMAX_M = 1): natural-ish cubic spline through the sorted,
 * de-duplicated table; cal = n_points rows of (frequency, value) */
void ao_set_calibration(ao_disp *a, int set, int n_points, double *cal)
{
    double y[AO_MAX_N], dx[AO_MAX_N], idx[AO_MAX_N], dmain[AO_MAX_N] = {0}, dsub[AO_MAX_N] = {0}, dsup[AO_MAX_N] = {0}, d[AO_MAX_N] = {0}, S[AO_MAX_N] = {0},
           b[AO_MAX_N] = {0}, v[AO_MAX_N] = {0};
    double *x = a->freqs[set];
    int i, j, k = 0, n;
    for (i = 1; i < n_points; i++) {            /* sort rows by frequency (the reference: qsort on the first column) */
        const double f = cal[2 * i], val = cal[2 * i + 1];
        for (j = i - 1; j >= 0 && cal[2 * j] > f; j--) { cal[2 * j + 2] = cal[2 * j]; cal[2 * j + 3] = cal[2 * j + 1]; }
        cal[2 * j + 2] = f; cal[2 * j + 3] = val;
    }
    for (i = 0; i < n_points; i++)
        if (i == n_points - 1 || cal[2 * i] != cal[2 * i + 2]) { x[k] = cal[2 * i]; y[k] = cal[2 * i + 1]; k++; }
    a->n_freqs[set] = n = k;
    a->cal_changed = 1;
    for (i = 0; i < n - 1; i++) {
        dx[i] = x[i + 1] - x[i];
        if (dx[i] < 1e-30) return;
        idx[i] = 1.0 / dx[i];
    }
    for (i = 1; i <= n - 2; i++) {
        if (i == 1) { dsub[i] = 0.0; dmain[i] = 3.0 * dx[i - 1] + 2.0 * dx[i]; dsup[i] = dx[i]; }
        else if (i == n - 2) { dsub[i] = dx[i - 1]; dmain[i] = 2.0 * dx[i - 1] + 3.0 * dx[i]; dsup[i] = 0.0; }
        else { dsub[i] = dx[i - 1]; dmain[i] = 2.0 * (dx[i - 1] + dx[i]); dsup[i] = dx[i]; }
        d[i] = 6.0 * ((y[i + 1] - y[i]) * idx[i] - (y[i] - y[i - 1]) * idx[i - 1]);
    }
    b[1] = dmain[1]; v[1] = d[1];
    for (i = 2; i <= n - 2; i++) {
        const double t = dsub[i] / b[i - 1];
        b[i] = dmain[i] - t * dsup[i - 1];
        v[i] = d[i] - t * v[i - 1];
    }
    S[n - 2] = v[n - 2] / b[n - 2];
    for (i = n - 3; i >= 1; i--) S[i] = (v[i] - dsup[i] * S[i + 1]) / b[i];
    S[0] = S[1]; S[n - 1] = S[n - 2];
    for (i = 0; i < n - 1; i++) {
        a->ac3[set][i] = (S[i + 1] - S[i]) / (6.0 * dx[i]);
        a->ac2[set][i] = 0.5 * S[i];
        a->ac1[set][i] = (y[i + 1] - y[i]) * idx[i] - (2.0 * dx[i] * S[i] + dx[i] * S[i + 1]) / 6.0;
        a->ac0[set][i] = y[i];
    }
}

/* SetAnalyzer, analyzer.c:999-1137 */
void ao_set_analyzer(ao_disp *a, int n_pixout, int typ, int flip, int sz, int bf_sz, int win_type, double pi, int ovrlp, int clp,
                     double fscLin, double fscHin, int n_pix, int n_stch, int calset, double fmin, double fmax, int max_w)
{
    int i, j, span;
    a->num_pixout = n_pixout; a->type = typ; a->buff_size = bf_sz; a->flip = flip; a->overlap = ovrlp; a->clip = clp;
    a->fsclipL = fscLin; a->fsclipH = fscHin; a->num_stitch = n_stch;
    if (sz != a->size || win_type != a->window_type || pi != a->PiAlpha)
        ao_window(win_type, sz, pi, a->window, &a->inv_coherent_gain, &a->inherent_power_gain, &a->inv_enb);
    a->size = sz; a->window_type = win_type; a->PiAlpha = pi; a->max_writeahead = max_w;
    a->norm_oneHz = 10.0 * wo_mlog10_value(1.0 / ((double)a->sample_rate / (double)a->size));
    if ((fmin != a->f_min || fmax != a->f_max) && fmin == 0.0 && fmax == 0.0)
        for (i = 0; i < AO_MAX_PIXELS; i++) a->cd[i] = 1.0;
    if ((fmax != 0.0 || fmin != 0.0) && (n_pix != a->num_pixels || fmin != a->f_min || fmax != a->f_max || calset != a->cal_set || a->cal_changed))
        ao_interpolate(a, calset, fmin, fmax, n_pix);
    a->incr = a->size - a->overlap;
    a->num_pixels = n_pix; a->f_min = fmin; a->f_max = fmax; a->cal_set = calset; a->cal_changed = 0;
    if (a->type == 0) { a->out_size = a->size / 2 + 1; a->scale = 4.0 / ((double)a->size * (double)a->size); }
    else { a->out_size = a->size; a->scale = 1.0 / ((double)a->size * (double)a->size); }
    span = a->out_size - 1 - 2 * a->clip;
    a->begin_ss = 0; a->end_ss = a->num_stitch - 1;
    a->fscL = (int)a->fsclipL; a->fscH = (int)a->fsclipH;
    while (a->fscL >= span) { a->fscL -= span; a->ss_bins[a->begin_ss] = 0; a->begin_ss++; }
    while (a->fscH >= span) { a->fscH -= span; a->ss_bins[a->end_ss] = 0; a->end_ss--; }
    a->pix_per_bin = (double)a->num_pixels / ((double)(a->num_stitch * span) - a->fsclipL - a->fsclipH - 1.0);
    a->det_offset = -a->pix_per_bin * (a->fsclipL - floor(a->fsclipL));
    a->bin_per_pix = ((double)(a->num_stitch * span) - 1.0 - a->fsclipL - a->fsclipH) / ((double)a->num_pixels - 1.0);
    a->stitch_flag = 0;
    for (i = 0; i < AO_MAX_PIXOUTS; i++) {
        a->w_pix[i] = a->r_pix[i] = a->last_pix[i] = 0;
        for (j = 0; j < AO_NUM_PIXEL_BUFFS; j++) a->pb_ready[i][j] = 0;
    }
    for (i = 0; i < AO_MAX_STITCH; i++) { a->busy[i] = a->ready[i] = a->have[i] = a->in_idx[i] = a->out_idx[i] = 0; }
}

/* Celiminate / eliminate with one LO (analyzer.c:179-279): the kept bins of sub-span ss, in display order, as powers */
static void ao_eliminate(ao_disp *a, int ss)
{
    const double *X = a->fft;
    double *r = a->result[ss];
    const int ilim = a->out_size - 1;
    int i, k = 0;
    if (a->type == 0) {
        const int begin = ss == a->begin_ss ? a->fscL + a->clip : a->clip;
        const int end = ss == a->end_ss ? a->out_size - 1 - a->clip - a->fscH : a->out_size - 1 - a->clip;
        if (a->flip) for (i = ilim - begin; i > ilim - end; i--) r[k++] = X[2 * i] * X[2 * i] + X[2 * i + 1] * X[2 * i + 1];
        else for (i = begin; i < end; i++) r[k++] = X[2 * i] * X[2 * i] + X[2 * i + 1] * X[2 * i + 1];
    } else {
        int begin0, end0, begin1, end1;
        if (ss == a->begin_ss) {
            begin0 = a->out_size / 2 + 1 + a->clip + a->fscL;
            begin1 = begin0 > a->out_size ? begin0 - a->out_size : 0;
        } else { begin0 = a->out_size / 2 + 1 + a->clip; begin1 = 0; }
        if (ss == a->end_ss) {
            end1 = a->out_size / 2 - a->clip - a->fscH;
            end0 = end1 < 0 ? a->out_size + end1 : a->out_size;
        } else { end0 = a->out_size; end1 = a->out_size / 2 - a->clip; }
        if (a->flip) {
            for (i = ilim - begin0; i > ilim - end0; i--) r[k++] = X[2 * i] * X[2 * i] + X[2 * i + 1] * X[2 * i + 1];
            for (i = ilim - begin1; i > ilim - end1; i--) r[k++] = X[2 * i] * X[2 * i] + X[2 * i + 1] * X[2 * i + 1];
        } else {
            for (i = begin0; i < end0; i++) r[k++] = X[2 * i] * X[2 * i] + X[2 * i + 1] * X[2 * i + 1];
            for (i = begin1; i < end1; i++) r[k++] = X[2 * i] * X[2 * i] + X[2 * i + 1] * X[2 * i + 1];
        }
    }
    a->ss_bins[ss] = k;
}

/* detector, analyzer.c:282-461 */
void ao_detector(int det_type, int m, int num_pixels, double pix_per_bin, double bin_per_pix, const double *bins, double *pixels,
                 double inv_enb, double fsclipL, double fsclipH, double det_offset)
{
    int i, pix_count = 0;
    if (pix_per_bin <= 1.0) {
        const int imin = fsclipL == floor(fsclipL) ? 0 : 1, ilim = fsclipH == floor(fsclipH) ? m : m - 1;
        int last, bcount = 0, rose = 0, fell = 0;
        double psum = 0.0, mini = 1.0e300, maxi = -1.0e300, prev_maxi = -1.0e300;
        if (det_type == 0) for (i = 0; i < num_pixels; i++) pixels[i] = -1.0e300;
        for (i = imin; i < ilim; i++) {
            last = pix_count;
            pix_count = (int)(det_offset + (double)i * pix_per_bin);
            if (pix_count >= num_pixels) pix_count = num_pixels - 1;
            switch (det_type) {
            case 0:                                         /* positive peak */
                if (bins[i] > pixels[pix_count]) pixels[pix_count] = bins[i];
                break;
            case 1: {                                       /* rosenfell: the next bin's pixel is taken WITHOUT the offset */
                const int next = (int)((double)(i + 1) * pix_per_bin);
                if (bins[i] < mini) mini = bins[i];
                if (bins[i] > maxi) maxi = bins[i];
                if (next == pix_count && i < ilim - 1) {
                    if (bins[i + 1] > bins[i]) rose = 1;
                    if (bins[i + 1] < bins[i]) fell = 1;
                } else {
                    if (rose && fell) pixels[pix_count] = (pix_count & 1) ? (prev_maxi > maxi ? prev_maxi : maxi) : mini;
                    else pixels[pix_count] = maxi;
                    rose = fell = 0; prev_maxi = maxi; mini = 1.0e300; maxi = -1.0e300;
                }
                break; }
            case 2: case 4: {                               /* average / rms over the bins of a pixel, noise-bandwidth corrected */
                const double t = det_type == 2 ? bins[i] : bins[i] * bins[i];
                if (pix_count == last) { psum += t; bcount++; }
                else {
                    pixels[last] = (det_type == 2 ? psum / (double)bcount : sqrt(psum / (double)bcount)) * inv_enb;
                    psum = t; bcount = 1;
                }
                if (i == ilim - 1) pixels[pix_count] = (det_type == 2 ? psum / (double)bcount : sqrt(psum / (double)bcount)) * inv_enb;
                break; }
            case 3:                                         /* sample: the middle bin of the pixel */
                if (pix_count == last) bcount++;
                else { pixels[last] = bins[i - bcount / 2 - 1] * inv_enb; bcount = 1; }
                if (i == ilim - 1) pixels[pix_count] = bins[i - bcount / 2] * inv_enb;
                break;
            default: break;
            }
        }
    } else {                                                /* more pixels than bins: straight lines between bins */
        const int ampl_comp = det_type == 2 || det_type == 3 || det_type == 4;
        double pix_pos = fsclipL - floor(fsclipL);
        for (i = 1; i < m; i++)
            while (pix_pos < (double)i + 1.0e-06 && pix_count < num_pixels) {
                const double frac = pix_pos - (double)(i - 1);
                pixels[pix_count] = bins[i - 1] * (1.0 - frac) + bins[i] * frac;
                if (ampl_comp) pixels[pix_count] *= inv_enb;
                pix_count++;
                pix_pos += bin_per_pix;
            }
    }
}

/* avenger, analyzer.c:463-553 */
static void ao_avenger(ao_disp *a, int o, float *pixels)
{
    const int n = a->num_pixels;
    const double *t = a->t_pixels[o], *cd = a->cd;
    double *sum = a->av_sum[o];
    const double scale = a->scale, back = a->av_backmult[o], onem = 1.0 - back;
    int i;
    switch (a->av_mode[o]) {
    case -1:
        for (i = 0; i < n; i++) {
            if (t[i] > sum[i]) sum[i] = t[i];
            pixels[i] = (float)(10.0 * wo_mlog10_value(scale * cd[i] * sum[i] + 1.0e-60));
        }
        break;
    case 1:
        for (i = 0; i < n; i++) {
            sum[i] = back * sum[i] + onem * t[i];
            pixels[i] = (float)(10.0 * wo_mlog10_value(scale * cd[i] * sum[i] + 1.0e-60));
        }
        break;
    case 2: {
        double factor;
        if (a->avail_frames[o] < a->num_average[o]) {
            factor = scale / (double)++a->avail_frames[o];
            for (i = 0; i < n; i++) {
                sum[i] += t[i];
                a->av_buff[o][a->av_in_idx[o]][i] = t[i];
                pixels[i] = (float)(10.0 * wo_mlog10_value(cd[i] * sum[i] * factor + 1.0e-60));
            }
        } else {
            factor = scale / (double)a->avail_frames[o];
            for (i = 0; i < n; i++) {
                sum[i] += t[i] - a->av_buff[o][a->av_out_idx[o]][i];
                a->av_buff[o][a->av_in_idx[o]][i] = t[i];
                pixels[i] = (float)(10.0 * wo_mlog10_value(cd[i] * sum[i] * factor + 1.0e-60));
            }
            if (++a->av_out_idx[o] == AO_MAX_AVERAGE) a->av_out_idx[o] = 0;
        }
        if (++a->av_in_idx[o] == AO_MAX_AVERAGE) a->av_in_idx[o] = 0;
        break; }
    case 3:
        for (i = 0; i < n; i++) {
            sum[i] = back * sum[i] + onem * (10.0 * wo_mlog10_value(scale * cd[i] * t[i] + 1e-60));
            pixels[i] = (float)sum[i];
        }
        break;
    default:
        for (i = 0; i < n; i++) pixels[i] = (float)(10.0 * wo_mlog10_value(scale * cd[i] * t[i] + 1.0e-60));
        break;
    }
    if (a->normalize[o]) for (i = 0; i < n; i++) pixels[i] += (float)a->norm_oneHz;
}

/* stitch, analyzer.c:555-600 */
static void ao_stitch(ao_disp *a)
{
    int n, m = 0, o, j, k;
    double *p = a->pre_av_out;
    for (n = a->begin_ss; n <= a->end_ss; n++) {
        memcpy(p, a->result[n], sizeof(double) * a->ss_bins[n]);
        p += a->ss_bins[n]; m += a->ss_bins[n];
    }
    for (o = 0; o < a->num_pixout; o++) {
        k = o;
        for (j = o - 1; j >= 0; j--) if (a->det_type[o] == a->det_type[j]) k = j;     /* an earlier output with the same detector */
        if (k == o) ao_detector(a->det_type[o], m, a->num_pixels, a->pix_per_bin, a->bin_per_pix, a->pre_av_out, a->t_pixels[o], a->inv_enb,
                                a->fsclipL, a->fsclipH, a->det_offset);
        else memcpy(a->t_pixels[o], a->t_pixels[k], sizeof(double) * a->num_pixels);
        ao_avenger(a, o, a->pixels[o][a->w_pix[o]]);
        a->last_pix[o] = a->w_pix[o];
        do a->w_pix[o] = (a->w_pix[o] + 1) % AO_NUM_PIXEL_BUFFS; while (a->w_pix[o] == a->r_pix[o]);
        a->pb_ready[o][a->last_pix[o]] = 1;
    }
    a->frames++;
}

/* spectra / Cspectra for sub-span ss (analyzer.c:602-745) */
static void ao_frame(ao_disp *a, int ss, int idx)
{
    int i;
    if (ss >= a->begin_ss && ss <= a->end_ss) {
        for (i = 0; i < a->size; i++) {
            a->fft[2 * i] = a->window[i] * (double)a->I[ss][idx];
            a->fft[2 * i + 1] = a->type == 0 ? 0.0 : a->window[i] * (double)a->Q[ss][idx];
            if (++idx >= a->bsize) idx -= a->bsize;
        }
        fo_fft(a->fft, a->size, -1);
    }
    if (a->snap_buff && a->snap_ss == ss) {                     /* Cspectra, analyzer.c:708-713: the second half of fft_out first */
        memcpy(a->snap_buff, a->fft + a->size, (size_t)a->size * sizeof(double));
        memcpy(a->snap_buff + a->size, a->fft, (size_t)a->size * sizeof(double));
        a->snap_buff = NULL;
        a->snap_taken = 1;
    }
    if (ss >= a->begin_ss && ss <= a->end_ss) {
        ao_eliminate(a, ss);
    }
    a->stitch_flag |= 1u << ss;
    if (a->stitch_flag == (1u << a->num_stitch) - 1u) {
        a->stitch_flag = 0;
        for (i = 0; i < AO_MAX_STITCH; i++) a->busy[i] = 0;
        ao_stitch(a);
    }
}

/* the dispatcher's scan, analyzer.c:884-917, until nothing is left to start */
static void ao_dispatch(ao_disp *a)
{
    int started = 1, ss;
    while (started) {
        started = 0;
        for (ss = 0; ss < a->num_stitch; ss++)
            if (!a->busy[ss] && a->ready[ss]) {
                const int idx = a->out_idx[ss];
                a->busy[ss] = 1;
                if ((a->out_idx[ss] += a->incr) >= a->bsize) a->out_idx[ss] -= a->bsize;
                if ((a->have[ss] -= a->incr) < a->size) a->ready[ss] = 0;
                ao_frame(a, ss, idx);
                started = 1;
            }
    }
}

static void ao_close_buffer(ao_disp *a, int ss)            /* the bookkeeping shared by Spectrum / Spectrum0 / Spectrum2 / CloseBuffer */
{
    if (a->have[ss] > a->max_writeahead) {
        if ((a->out_idx[ss] += a->have[ss] - a->max_writeahead) >= a->bsize) a->out_idx[ss] -= a->bsize;
        a->have[ss] = a->max_writeahead;
    }
    if ((a->have[ss] += a->buff_size) >= a->size) a->ready[ss] = 1;
    if ((a->in_idx[ss] += a->buff_size) >= a->bsize) a->in_idx[ss] = 0;
    ao_dispatch(a);
}

/* Spectrum0, analyzer.c:1536-1579: interleaved doubles, the SECOND of each pair is I */
void ao_spectrum0(ao_disp *a, int run, int ss, const double *pbuff)
{
    int i;
    if (!run) return;
    for (i = 0; i < a->buff_size; i++) {
        a->I[ss][a->in_idx[ss] + i] = (float)pbuff[2 * i + 1];
        a->Q[ss][a->in_idx[ss] + i] = (float)pbuff[2 * i];
    }
    ao_close_buffer(a, ss);
}

/* Spectrum, analyzer.c:1451-1487: separate I and Q vectors of dINREAL */
void ao_spectrum(ao_disp *a, int ss, const float *pI, const float *pQ)
{
    memcpy(a->I[ss] + a->in_idx[ss], pI, sizeof(float) * a->buff_size);
    memcpy(a->Q[ss] + a->in_idx[ss], pQ, sizeof(float) * a->buff_size);
    ao_close_buffer(a, ss);
}

/* GetPixels, analyzer.c:1315-1334 */
int ao_get_pixels(ao_disp *a, int pixout, float *pix)
{
    a->r_pix[pixout] = a->last_pix[pixout];
    if (!a->pb_ready[pixout][a->r_pix[pixout]]) return 0;
    memcpy(pix, a->pixels[pixout][a->r_pix[pixout]], sizeof(float) * a->num_pixels);
    a->pb_ready[pixout][a->r_pix[pixout]] = 0;
    return 1;
}

/* ResetPixelBuffers, analyzer.c:927-996.  (Mode 2 keeps av_sum: its case only has a comment, the sum is not cleared.) */
/* SnapSpectrum, analyzer.c:1337-1346, without the wait: the next frame of sub-span ss is copied to buf (2 * size doubles) */
void ao_snap(ao_disp *a, int ss, double *buf) { a->snap_buff = buf; a->snap_ss = ss; a->snap_taken = 0; }
int ao_snap_taken(const ao_disp *a) { return a->snap_taken; }

void ao_reset_pixel_buffers(ao_disp *a)
{
    int i, j, k;
    for (i = 0; i < AO_MAX_PIXOUTS; i++) {
        for (j = 0; j < AO_MAX_PIXELS; j++) a->t_pixels[i][j] = 0.0;
        for (j = 0; j < AO_MAX_AVERAGE; j++) for (k = 0; k < AO_MAX_PIXELS; k++) a->av_buff[i][j][k] = 0.0;
        switch (a->av_mode[i]) {
        case 1: for (j = 0; j < AO_MAX_PIXELS; j++) a->av_sum[i][j] = 1.0e-12; break;
        case 2: break;
        case 3: for (j = 0; j < AO_MAX_PIXELS; j++) a->av_sum[i][j] = -160.0; break;
        default: memset(a->av_sum[i], 0, sizeof(double) * AO_MAX_PIXELS); break;
        }
        a->avail_frames[i] = a->av_in_idx[i] = a->av_out_idx[i] = 0;
        a->w_pix[i] = a->r_pix[i] = a->last_pix[i] = 0;
        for (j = 0; j < AO_NUM_PIXEL_BUFFS; j++) a->pb_ready[i][j] = 0;
    }
    memset(a->pre_av_out, 0, sizeof(double) * a->max_size * a->max_stitch);
    a->stitch_flag = 0;
    for (i = 0; i < AO_MAX_STITCH; i++) { a->busy[i] = a->ready[i] = a->have[i] = a->in_idx[i] = a->out_idx[i] = 0; }
}

void ao_set_detector_mode(ao_disp *a, int pixout, int mode) { a->det_type[pixout] = mode; }     /* analyzer.c:1582 */
void ao_set_average_mode(ao_disp *a, int pixout, int mode)                                      /* analyzer.c:1594-1623 */
{
    int i;
    if (a->av_mode[pixout] == mode) return;
    a->av_mode[pixout] = mode;
    if (mode == 2) { a->avail_frames[pixout] = a->av_in_idx[pixout] = a->av_out_idx[pixout] = 0; return; }
    for (i = 0; i < AO_MAX_PIXELS; i++) a->av_sum[pixout][i] = mode == 1 ? 1.0e-12 : mode == 3 ? -160.0 : 0.0;
}
void ao_set_num_average(ao_disp *a, int pixout, int num)                                        /* analyzer.c:1626-1638 */
{
    if (a->num_average[pixout] == num) return;
    a->num_average[pixout] = num;
    a->avail_frames[pixout] = a->av_in_idx[pixout] = a->av_out_idx[pixout] = 0;
}
void ao_set_av_backmult(ao_disp *a, int pixout, double mult) { a->av_backmult[pixout] = mult; } /* analyzer.c:1641 */
void ao_set_sample_rate(ao_disp *a, int rate)                                                   /* analyzer.c:1653-1663, 919-924 */
{
    a->sample_rate = rate;
    a->norm_oneHz = 10.0 * wo_mlog10_value(1.0 / ((double)a->sample_rate / (double)a->size));
}
void ao_set_norm_onehz(ao_disp *a, int pixout, int norm) { a->normalize[pixout] = norm; }       /* analyzer.c:1666 */
double ao_get_enb(ao_disp *a) { return 1.0 / a->inv_enb; }                                      /* analyzer.c:1678 */
long ao_frames(ao_disp *a) { return a->frames; }
const double *ao_window_ptr(ao_disp *a) { return a->window; }
const double *ao_cd_ptr(ao_disp *a) { return a->cd; }
