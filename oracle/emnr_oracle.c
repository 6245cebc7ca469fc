/* emnr_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of WDSP's EMNR ("NR2"), wdsp/emnr.c, for one channel:
 * overlap-add STFT (fsize 4096, overlap 4, sqrt-Hamming analysis and synthesis windows), noise power estimate by
 * minimum statistics (LambdaD, emnr.c:604-739) or the speech-presence estimator (LambdaDs, :741-754), gain by the
 * Gaussian / log / gamma-table / trained methods (calc_gain, :885-1013), artifact-elimination post-filter (aepf, :777-816).
 * PARITY UNPINNED by reference execution (every wdsp source needs <fftw3.h>); the gain tables GG / GGS and zetaHat are
 * data the reference loads at run time (emnr.c:317-334) and are handed in by the caller. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fft_oracle.h"
#include "emnr_oracle.h"

#define PI 3.1415926535897932       /* wdsp/comm.h:146 */
#define dmin(a, b) ((a) < (b) ? (a) : (b))
#define dmax(a, b) ((a) > (b) ? (a) : (b))

struct wo_emnr {
    int run, position, bsize, fsize, ovrlp, incr, iasize, iainidx, iaoutidx, oasize, oainidx, init_oainidx, oaoutidx, msize, nsamps, saveidx;
    double rate, ogain, gain;
    double *window, *inaccum, *forfftin, *forfftout, *mask, *revfftin, *revfftout, *save[4], *outaccum;
    /* g */
    int gain_method, npe_method, ae_run, dim_zeta;
    double *lambda_y, *lambda_d, *prev_gamma, *prev_mask, gf1p5, alpha, eps_floor, gamma_max, xi_min, q, gmax;
    const double *GG, *GGS, *zeta_hat;
    const int *zeta_true;
    double zeta_thresh, z_gamma_min, z_gamma_max, z_xihat_min, z_xihat_max;
    /* np */
    double alphaCsmooth, alphaMax, alphaCmin, alphaMin_max_value, snrq, betamax, invQeqMax, av, Dtime, MofD, MofV, invQbar_points[4], nsmax[4];
    int U, V, D, subwc, amb_idx;
    double alphaC, *p, *alphaOptHat, *alphaHat, *sigma2N, *pbar, *p2bar, *Qeq, *bmin, *bmin_sub, *actmin, *actmin_sub, *pmin_u, *actminbuff[16];
    int *k_mod, *lmin_flag;
    /* nps */
    double alpha_pow, alpha_Pbar, epsH1, epsH1r, *s_sigma2N, *PH1y, *Pbar, *EN2y;
    /* npl */
    double l_eta, l_gamma, l_beta, l_alpha_d, l_alpha_p, delta_LF, delta_MF, delta_0, delta_1, delta_2, *lP, *lPmin, *lp, *lD;
    /* ae */
    double zetaThresh, psi, t2, *nmask;
};

static double *dz(int n) { return (double *)calloc((size_t)n, sizeof(double)); }

static double bessI0(double x)                  /* emnr.c:43-82 */
{
    double res, p;
    if (x == 0.0) return 1.0;
    if (x < 0.0) x = -x;
    if (x <= 3.75) {
        p = x / 3.75; p = p * p;
        res = ((((( 0.0045813 * p + 0.0360768) * p + 0.2659732) * p + 1.2067492) * p + 3.0899424) * p + 3.5156229) * p + 1.0;
    } else {
        p = 3.75 / x;
        res = exp(x) / sqrt(x) * (((((((( + 0.00392377 * p - 0.01647633) * p + 0.02635537) * p - 0.02057706) * p + 0.00916281) * p
              - 0.00157565) * p + 0.00225319) * p + 0.01328592) * p + 0.39894228);
    }
    return res;
}

static double bessI1(double x)                  /* emnr.c:84-124 */
{
    double res, p;
    if (x == 0.0) return 0.0;
    if (x < 0.0) x = -x;
    if (x <= 3.75) {
        p = x / 3.75; p = p * p;
        res = x * (((((( 0.00032411 * p + 0.00301532) * p + 0.02658733) * p + 0.15084934) * p + 0.51498869) * p + 0.87890594) * p + 0.5);
    } else {
        p = 3.75 / x;
        res = exp(x) / sqrt(x) * (((((((( - 0.00420059 * p + 0.01787654) * p - 0.02895312) * p + 0.02282967) * p - 0.01031555) * p
              + 0.00163801) * p - 0.00362018) * p - 0.03988024) * p + 0.39894228);
    }
    return res;
}

static double e1xb(double x)                    /* emnr.c:132-165 */
{
    double e1, ga, r, t, t0;
    int k, m;
    if (x == 0.0) e1 = 1.0e300;
    else if (x <= 1.0) {
        e1 = 1.0; r = 1.0;
        for (k = 1; k <= 25; k++) {
            r = -r * k * x / ((k + 1.0) * (k + 1.0));
            e1 = e1 + r;
            if (fabs(r) <= fabs(e1) * 1.0e-15) break;
        }
        ga = 0.5772156649015328;
        e1 = -ga - log(x) + x * e1;
    } else {
        m = 20 + (int)(80.0 / x);
        t0 = 0.0;
        for (k = m; k >= 1; k--) t0 = (double)k / (1.0 + k / (x + t0));
        t = 1.0 / (x + t0);
        e1 = exp(-x) * t;
    }
    return e1;
}

static void interpM(double *res, double x, int nvals, const double *xvals, const double *yvals)     /* emnr.c:185-202 */
{
    if (x <= xvals[0]) *res = yvals[0];
    else if (x >= xvals[nvals - 1]) *res = yvals[nvals - 1];
    else {
        int idx = 0;
        double xllow, xlhigh, frac;
        while (x >= xvals[idx]) idx++;
        xllow = log10(xvals[idx - 1]);
        xlhigh = log10(xvals[idx]);
        frac = (log10(x) - xllow) / (xlhigh - xllow);
        *res = yvals[idx - 1] + frac * (yvals[idx] - yvals[idx - 1]);
    }
}

wo_emnr *wo_emnr_create(int bsize, int rate)    /* create_emnr with create_rxa's arguments (RXA.c:319-332) + calc_emnr (emnr.c:240-497) */
{
    static const double Dvals[18] = { 1.0, 2.0, 5.0, 8.0, 10.0, 15.0, 20.0, 30.0, 40.0, 60.0, 80.0, 120.0, 140.0, 160.0, 180.0, 220.0, 260.0, 300.0 };
    static const double Mvals[18] = { 0.000, 0.260, 0.480, 0.580, 0.610, 0.668, 0.705, 0.762, 0.800, 0.841, 0.865, 0.890, 0.900, 0.910, 0.920,
                                      0.930, 0.935, 0.940 };
    int i, k, ku;
    double arg, sum, tau, db;
    wo_emnr *a = (wo_emnr *)calloc(1, sizeof(*a));
    a->bsize = bsize; a->fsize = 4096; a->ovrlp = 4; a->rate = rate; a->ogain = 1.0;
    a->gain_method = 2; a->npe_method = 0; a->ae_run = 1;
    a->incr = a->fsize / a->ovrlp;
    a->gain = a->ogain / a->fsize / (double)a->ovrlp;
    if (a->fsize > a->bsize) a->iasize = a->fsize; else a->iasize = a->bsize + a->fsize - a->incr;
    if (a->fsize > a->bsize) {
        if (a->bsize > a->incr) a->oasize = a->bsize; else a->oasize = a->incr;
        a->oainidx = (a->fsize - a->bsize - a->incr) % a->oasize;
    } else {
        a->oasize = a->bsize;
        a->oainidx = a->fsize - a->incr;
    }
    a->init_oainidx = a->oainidx;
    a->msize = a->fsize / 2 + 1;
    a->window = dz(a->fsize); a->inaccum = dz(a->iasize); a->forfftin = dz(a->fsize); a->forfftout = dz(2 * a->msize);
    a->mask = dz(a->msize); a->revfftin = dz(2 * a->msize); a->revfftout = dz(a->fsize);
    for (i = 0; i < a->ovrlp; i++) a->save[i] = dz(a->fsize);
    a->outaccum = dz(a->oasize);
    /* calc_window, wintype 0 (emnr.c:160-183) */
    arg = 2.0 * PI / (double)a->fsize; sum = 0.0;
    for (i = 0; i < a->fsize; i++) { a->window[i] = sqrt(0.54 - 0.46 * cos((double)i * arg)); sum += a->window[i]; }
    for (i = 0; i < a->fsize; i++) a->window[i] *= (double)a->fsize / sum;
    /* g */
    a->lambda_y = dz(a->msize); a->lambda_d = dz(a->msize); a->prev_gamma = dz(a->msize); a->prev_mask = dz(a->msize);
    a->gf1p5 = sqrt(PI) / 2.0;
    tau = -128.0 / 8000.0 / log(0.985);
    a->alpha = exp(-a->incr / a->rate / tau);
    a->eps_floor = 1.0e-300; a->gamma_max = 40.0; a->xi_min = pow(10.0, -40.0 / 10.0); a->q = 0.2;
    for (i = 0; i < a->msize; i++) { a->prev_mask[i] = 1.0; a->prev_gamma[i] = 1.0; }
    a->gmax = 10000.0;
    a->dim_zeta = 60; a->zeta_thresh = -2.0;
    /* np */
    tau = -128.0 / 8000.0 / log(0.7);  a->alphaCsmooth = exp(-a->incr / a->rate / tau);
    tau = -128.0 / 8000.0 / log(0.96); a->alphaMax = exp(-a->incr / a->rate / tau);
    tau = -128.0 / 8000.0 / log(0.7);  a->alphaCmin = exp(-a->incr / a->rate / tau);
    tau = -128.0 / 8000.0 / log(0.3);  a->alphaMin_max_value = exp(-a->incr / a->rate / tau);
    a->snrq = -a->incr / (0.064 * a->rate);
    tau = -128.0 / 8000.0 / log(0.8);  a->betamax = exp(-a->incr / a->rate / tau);
    a->invQeqMax = 0.5; a->av = 2.12; a->Dtime = 8.0 * 12.0 * 128.0 / 8000.0;
    a->U = 8;
    a->V = (int)(0.5 + (a->Dtime * a->rate / (a->U * a->incr)));
    if (a->V < 4) a->V = 4;
    if ((a->U = (int)(0.5 + (a->Dtime * a->rate / (a->V * a->incr)))) < 1) a->U = 1;
    a->D = a->U * a->V;
    interpM(&a->MofD, a->D, 18, Dvals, Mvals);
    interpM(&a->MofV, a->V, 18, Dvals, Mvals);
    a->invQbar_points[0] = 0.03; a->invQbar_points[1] = 0.05; a->invQbar_points[2] = 0.06; a->invQbar_points[3] = 1.0e300;
    db = 10.0 * log10(8.0) / (12.0 * 128 / 8000); a->nsmax[0] = pow(10.0, db / 10.0 * a->V * a->incr / a->rate);
    db = 10.0 * log10(4.0) / (12.0 * 128 / 8000); a->nsmax[1] = pow(10.0, db / 10.0 * a->V * a->incr / a->rate);
    db = 10.0 * log10(2.0) / (12.0 * 128 / 8000); a->nsmax[2] = pow(10.0, db / 10.0 * a->V * a->incr / a->rate);
    db = 10.0 * log10(1.2) / (12.0 * 128 / 8000); a->nsmax[3] = pow(10.0, db / 10.0 * a->V * a->incr / a->rate);
    a->p = dz(a->msize); a->alphaOptHat = dz(a->msize); a->alphaHat = dz(a->msize); a->sigma2N = dz(a->msize); a->pbar = dz(a->msize);
    a->p2bar = dz(a->msize); a->Qeq = dz(a->msize); a->bmin = dz(a->msize); a->bmin_sub = dz(a->msize);
    a->k_mod = (int *)calloc((size_t)a->msize, sizeof(int)); a->actmin = dz(a->msize); a->actmin_sub = dz(a->msize);
    a->lmin_flag = (int *)calloc((size_t)a->msize, sizeof(int)); a->pmin_u = dz(a->msize);
    for (i = 0; i < a->U; i++) a->actminbuff[i] = dz(a->msize);
    a->alphaC = 1.0; a->subwc = a->V; a->amb_idx = 0;
    for (k = 0; k < a->msize; k++) a->lambda_y[k] = 0.5;
    memcpy(a->p, a->lambda_y, (size_t)a->msize * sizeof(double));
    memcpy(a->sigma2N, a->lambda_y, (size_t)a->msize * sizeof(double));
    memcpy(a->pbar, a->lambda_y, (size_t)a->msize * sizeof(double));
    memcpy(a->pmin_u, a->lambda_y, (size_t)a->msize * sizeof(double));
    for (k = 0; k < a->msize; k++) {
        a->p2bar[k] = a->lambda_y[k] * a->lambda_y[k];
        a->actmin[k] = 1.0e300; a->actmin_sub[k] = 1.0e300;
        for (ku = 0; ku < a->U; ku++) a->actminbuff[ku][k] = 1.0e300;
    }
    /* nps */
    tau = -128.0 / 8000.0 / log(0.8); a->alpha_pow = exp(-a->incr / a->rate / tau);
    tau = -128.0 / 8000.0 / log(0.9); a->alpha_Pbar = exp(-a->incr / a->rate / tau);
    a->epsH1 = pow(10.0, 15.0 / 10.0); a->epsH1r = a->epsH1 / (1.0 + a->epsH1);
    a->s_sigma2N = dz(a->msize); a->PH1y = dz(a->msize); a->Pbar = dz(a->msize); a->EN2y = dz(a->msize);
    for (i = 0; i < a->msize; i++) { a->s_sigma2N[i] = 0.5; a->Pbar[i] = 0.5; }
    /* npl (emnr.c:458-489) */
    a->lP = dz(a->msize); a->lPmin = dz(a->msize); a->lp = dz(a->msize); a->lD = dz(a->msize);
    tau = -256.0 / (20100.0 * log(0.7));   a->l_eta = exp(-a->incr / (a->rate * tau));
    tau = -256.0 / (20100.0 * log(0.998)); a->l_gamma = exp(-a->incr / (a->rate * tau));
    tau = -256.0 / (20100.0 * log(0.8));   a->l_beta = exp(-a->incr / (a->rate * tau));
    tau = -256.0 / (20100.0 * log(0.85));  a->l_alpha_d = exp(-a->incr / (a->rate * tau));
    tau = -256.0 / (20100.0 * log(0.2));   a->l_alpha_p = exp(-a->incr / (a->rate * tau));
    a->delta_LF = 1000.0 / (a->rate / 2) * a->msize; a->delta_MF = 3000.0 / (a->rate / 2) * a->msize;
    a->delta_0 = 2.0; a->delta_1 = 2.0; a->delta_2 = 5.0;
    /* ae */
    a->zetaThresh = 0.75; a->psi = 20.0; a->t2 = 0.20; a->nmask = dz(a->msize);
    return a;
}

void wo_emnr_free(wo_emnr *a)
{
    int i;
    if (!a) return;
    free(a->window); free(a->inaccum); free(a->forfftin); free(a->forfftout); free(a->mask); free(a->revfftin); free(a->revfftout);
    for (i = 0; i < a->ovrlp; i++) free(a->save[i]);
    free(a->outaccum); free(a->lambda_y); free(a->lambda_d); free(a->prev_gamma); free(a->prev_mask);
    free(a->p); free(a->alphaOptHat); free(a->alphaHat); free(a->sigma2N); free(a->pbar); free(a->p2bar); free(a->Qeq); free(a->bmin);
    free(a->bmin_sub); free(a->k_mod); free(a->actmin); free(a->actmin_sub); free(a->lmin_flag); free(a->pmin_u);
    for (i = 0; i < 16; i++) free(a->actminbuff[i]);
    free(a->s_sigma2N); free(a->PH1y); free(a->Pbar); free(a->EN2y); free(a->nmask); free(a->lP); free(a->lPmin); free(a->lp); free(a->lD);
    free(a);
}

void wo_emnr_set_tables(wo_emnr *a, const double *GG, const double *GGS, const double *zeta_hat, const int *zeta_true, double gmin,
                        double gmax, double ximin, double ximax)
{
    a->GG = GG; a->GGS = GGS; a->zeta_hat = zeta_hat; a->zeta_true = zeta_true;
    a->z_gamma_min = gmin; a->z_gamma_max = gmax; a->z_xihat_min = ximin; a->z_xihat_max = ximax;
}

void wo_emnr_flush(wo_emnr *a)                  /* emnr.c:583-596 */
{
    int i;
    memset(a->inaccum, 0, (size_t)a->iasize * sizeof(double));
    for (i = 0; i < a->ovrlp; i++) memset(a->save[i], 0, (size_t)a->fsize * sizeof(double));
    memset(a->outaccum, 0, (size_t)a->oasize * sizeof(double));
    a->nsamps = 0; a->iainidx = 0; a->iaoutidx = 0; a->oainidx = a->init_oainidx; a->oaoutidx = 0; a->saveidx = 0;
}

int *wo_emnr_run(wo_emnr *a) { return &a->run; }
int *wo_emnr_position(wo_emnr *a) { return &a->position; }
void wo_emnr_set_gain_method(wo_emnr *a, int m) { a->gain_method = m; }
void wo_emnr_set_npe_method(wo_emnr *a, int m) { a->npe_method = m; }
void wo_emnr_set_ae_run(wo_emnr *a, int run) { a->ae_run = run; }
void wo_emnr_set_scalars(wo_emnr *a, double zetaThresh, double psi, double zeta_thresh, double t2)   /* emnr.c:1145-1174: < 0 keeps a value */
{
    if (zetaThresh >= 0) a->zetaThresh = zetaThresh;
    if (psi >= 0) a->psi = psi;
    if (zeta_thresh > -1e300) a->zeta_thresh = zeta_thresh;
    if (t2 >= 0) a->t2 = t2;
}

static void LambdaD(wo_emnr *a)                 /* emnr.c:604-739 */
{
    int k, ku;
    double f0, f1, f2, f3, sum_prev_p = 0.0, sum_lambda_y = 0.0, alphaCtilda, sum_prev_sigma2N = 0.0, alphaMin, SNR, beta, varHat, invQeq,
           invQbar, bc, QeqTilda, QeqTildaSub, noise_slope_max, mn;
    for (k = 0; k < a->msize; k++) { sum_prev_p += a->p[k]; sum_lambda_y += a->lambda_y[k]; sum_prev_sigma2N += a->sigma2N[k]; }
    for (k = 0; k < a->msize; k++) { f0 = a->p[k] / a->sigma2N[k] - 1.0; a->alphaOptHat[k] = 1.0 / (1.0 + f0 * f0); }
    SNR = sum_prev_p / sum_prev_sigma2N;
    alphaMin = dmin(a->alphaMin_max_value, pow(SNR, a->snrq));
    for (k = 0; k < a->msize; k++) if (a->alphaOptHat[k] < alphaMin) a->alphaOptHat[k] = alphaMin;
    f1 = sum_prev_p / sum_lambda_y - 1.0;
    alphaCtilda = 1.0 / (1.0 + f1 * f1);
    a->alphaC = a->alphaCsmooth * a->alphaC + (1.0 - a->alphaCsmooth) * dmax(alphaCtilda, a->alphaCmin);
    f2 = a->alphaMax * a->alphaC;
    for (k = 0; k < a->msize; k++) a->alphaHat[k] = f2 * a->alphaOptHat[k];
    for (k = 0; k < a->msize; k++) a->p[k] = a->alphaHat[k] * a->p[k] + (1.0 - a->alphaHat[k]) * a->lambda_y[k];
    invQbar = 0.0;
    for (k = 0; k < a->msize; k++) {
        beta = dmin(a->betamax, a->alphaHat[k] * a->alphaHat[k]);
        a->pbar[k] = beta * a->pbar[k] + (1.0 - beta) * a->p[k];
        a->p2bar[k] = beta * a->p2bar[k] + (1.0 - beta) * a->p[k] * a->p[k];
        varHat = a->p2bar[k] - a->pbar[k] * a->pbar[k];
        invQeq = varHat / (2.0 * a->sigma2N[k] * a->sigma2N[k]);
        if (invQeq > a->invQeqMax) invQeq = a->invQeqMax;
        a->Qeq[k] = 1.0 / invQeq;
        invQbar += invQeq;
    }
    invQbar /= (double)a->msize;
    bc = 1.0 + a->av * sqrt(invQbar);
    for (k = 0; k < a->msize; k++) {
        QeqTilda = (a->Qeq[k] - 2.0 * a->MofD) / (1.0 - a->MofD);
        QeqTildaSub = (a->Qeq[k] - 2.0 * a->MofV) / (1.0 - a->MofV);
        a->bmin[k] = 1.0 + 2.0 * (a->D - 1.0) / QeqTilda;
        a->bmin_sub[k] = 1.0 + 2.0 * (a->V - 1.0) / QeqTildaSub;
    }
    memset(a->k_mod, 0, (size_t)a->msize * sizeof(int));
    for (k = 0; k < a->msize; k++) {
        f3 = a->p[k] * a->bmin[k] * bc;
        if (f3 < a->actmin[k]) { a->actmin[k] = f3; a->actmin_sub[k] = a->p[k] * a->bmin_sub[k] * bc; a->k_mod[k] = 1; }
    }
    if (a->subwc == a->V) {
        if (invQbar < a->invQbar_points[0]) noise_slope_max = a->nsmax[0];
        else if (invQbar < a->invQbar_points[1]) noise_slope_max = a->nsmax[1];
        else if (invQbar < a->invQbar_points[2]) noise_slope_max = a->nsmax[2];
        else noise_slope_max = a->nsmax[3];
        for (k = 0; k < a->msize; k++) {
            if (a->k_mod[k]) a->lmin_flag[k] = 0;
            a->actminbuff[a->amb_idx][k] = a->actmin[k];
            mn = 1.0e300;
            for (ku = 0; ku < a->U; ku++) if (a->actminbuff[ku][k] < mn) mn = a->actminbuff[ku][k];
            a->pmin_u[k] = mn;
            if ((a->lmin_flag[k] == 1) && (a->actmin_sub[k] < noise_slope_max * a->pmin_u[k]) && (a->actmin_sub[k] > a->pmin_u[k])) {
                a->pmin_u[k] = a->actmin_sub[k];
                for (ku = 0; ku < a->U; ku++) a->actminbuff[ku][k] = a->actmin_sub[k];
            }
            a->lmin_flag[k] = 0;
            a->actmin[k] = 1.0e300;
            a->actmin_sub[k] = 1.0e300;
        }
        if (++a->amb_idx == a->U) a->amb_idx = 0;
        a->subwc = 1;
    } else {
        if (a->subwc > 1) {
            for (k = 0; k < a->msize; k++) {
                if (a->k_mod[k]) {
                    a->lmin_flag[k] = 1;
                    a->sigma2N[k] = dmin(a->actmin_sub[k], a->pmin_u[k]);
                    a->pmin_u[k] = a->sigma2N[k];
                }
            }
        }
        ++a->subwc;
    }
    memcpy(a->lambda_d, a->sigma2N, (size_t)a->msize * sizeof(double));
}

static void LambdaDs(wo_emnr *a)                /* emnr.c:741-754 */
{
    int k;
    for (k = 0; k < a->msize; k++) {
        a->PH1y[k] = 1.0 / (1.0 + (1.0 + a->epsH1) * exp(-a->epsH1r * a->lambda_y[k] / a->s_sigma2N[k]));
        a->Pbar[k] = a->alpha_Pbar * a->Pbar[k] + (1.0 - a->alpha_Pbar) * a->PH1y[k];
        if (a->Pbar[k] > 0.99) a->PH1y[k] = dmin(a->PH1y[k], 0.99);
        a->EN2y[k] = (1.0 - a->PH1y[k]) * a->lambda_y[k] + a->PH1y[k] * a->s_sigma2N[k];
        a->s_sigma2N[k] = a->alpha_pow * a->s_sigma2N[k] + (1.0 - a->alpha_pow) * a->EN2y[k];
    }
    memcpy(a->lambda_d, a->s_sigma2N, (size_t)a->msize * sizeof(double));
}

static void LambdaDl(wo_emnr *a)                /* emnr.c:756-775 */
{
    double P_old, c, Sr, delta, I, alpha_s;
    int k;
    c = (1.0 - a->l_gamma) / (1.0 - a->l_beta);
    for (k = 0; k < a->msize; k++) {
        P_old = a->lP[k];
        a->lP[k] = a->l_eta * P_old + (1.0 - a->l_eta) * a->lambda_y[k];
        if (a->lPmin[k] < a->lP[k]) a->lPmin[k] = a->l_gamma * a->lPmin[k] + c * (a->lP[k] - a->l_beta * P_old);
        else a->lPmin[k] = a->lP[k];
        Sr = a->lP[k] / a->lPmin[k];
        if (k <= a->delta_LF) delta = a->delta_0;
        else if (k <= a->delta_MF) delta = a->delta_1;
        else delta = a->delta_2;
        I = Sr > delta ? 1.0 : 0.0;
        a->lp[k] = a->l_alpha_p * a->lp[k] + (1.0 - a->l_alpha_p) * I;
        alpha_s = a->l_alpha_d + (1.0 - a->l_alpha_d) * a->lp[k];
        a->lD[k] = alpha_s * a->lD[k] + (1.0 - alpha_s) * a->lambda_y[k];
    }
    memcpy(a->lambda_d, a->lD, (size_t)a->msize * sizeof(double));
}

static void aepf(wo_emnr *a)                    /* emnr.c:777-816 */
{
    int k, m, N, n;
    double sumPre = 0.0, sumPost = 0.0, zeta, zetaT;
    for (k = 0; k < a->msize; k++) { sumPre += a->lambda_y[k]; sumPost += a->mask[k] * a->mask[k] * a->lambda_y[k]; }
    zeta = sumPost / sumPre;
    zetaT = zeta >= a->zetaThresh ? 1.0 : zeta;
    if (zetaT == 1.0) N = 1;
    else N = 1 + 2 * (int)(0.5 + a->psi * (1.0 - zetaT / a->zetaThresh));
    n = N / 2;
    for (k = 0; k < n; k++) {
        a->nmask[k] = 0.0;
        for (m = 0; m <= 2 * k; m++) a->nmask[k] += a->mask[m];
        a->nmask[k] /= (double)(2 * k + 1);
    }
    for (k = n; k < (a->msize - n); k++) {
        a->nmask[k] = 0.0;
        for (m = k - n; m <= (k + n); m++) a->nmask[k] += a->mask[m];
        a->nmask[k] /= (double)N;
    }
    for (k = a->msize - n; k < a->msize; k++) {
        a->nmask[k] = 0.0;
        for (m = (a->msize - 1); m >= (-a->msize + 2 * k + 1); m--) a->nmask[k] += a->mask[m];
        a->nmask[k] /= (double)(2 * (a->msize - k) - 1);
    }
    memcpy(a->mask, a->nmask, (size_t)a->msize * sizeof(double));
    if (a->gain_method == 3 && zetaT < a->t2)
        for (k = 0; k < a->msize; k++) a->mask[k] *= 0.05;
}

static double getKey(const double *type, double gamma, double xi)      /* emnr.c:818-862 */
{
    int ngamma1, ngamma2, nxi1, nxi2;
    double tg, tx, dg, dx;
    const double dmn = 0.001, dmx = 1000.0;
    if (gamma <= dmn) { ngamma1 = ngamma2 = 0; tg = 0.0; }
    else if (gamma >= dmx) { ngamma1 = ngamma2 = 240; tg = 60.0; }
    else { tg = 10.0 * log10(gamma / dmn); ngamma1 = (int)(4.0 * tg); ngamma2 = ngamma1 + 1; }
    if (xi <= dmn) { nxi1 = nxi2 = 0; tx = 0.0; }
    else if (xi >= dmx) { nxi1 = nxi2 = 240; tx = 60.0; }
    else { tx = 10.0 * log10(xi / dmn); nxi1 = (int)(4.0 * tx); nxi2 = nxi1 + 1; }
    dg = (tg - 0.25 * ngamma1) / 0.25;
    dx = (tx - 0.25 * nxi1) / 0.25;
    return (1.0 - dg) * (1.0 - dx) * type[241 * nxi1 + ngamma1] + (1.0 - dg) * dx * type[241 * nxi2 + ngamma1]
         + dg * (1.0 - dx) * type[241 * nxi1 + ngamma2] + dg * dx * type[241 * nxi2 + ngamma2];
}

static double mlog10(double val)                /* wdsp/meterlog10.c:29-32,547-554 (see wdsp_oracle.c wo_mlog10) */
{
    unsigned long long N;
    int e, m;
    memcpy(&N, &val, 8);
    e = (int)((N >> 52) & 2047) - 1023;
    m = (int)((N >> (52 - 11)) & 2047);
    return 0.301029995663981 * (e + log2(1.0 + m / 2048.0));
}

static int getZeta(wo_emnr *a, double gamma, double eps, double *zeta)  /* emnr.c:864-883, with its xi_dB >= dim_zeta test */
{
    int index, i_gamma, i_xi, ztvalue;
    double gamma_dB = 10.0 * mlog10(gamma), xi_dB = 10.0 * mlog10(eps);
    double gamma_per_cell = (a->z_gamma_max - a->z_gamma_min) / a->dim_zeta, xi_per_cell = (a->z_xihat_max - a->z_xihat_min) / a->dim_zeta;
    i_gamma = (int)floor((gamma_dB - a->z_gamma_min) / gamma_per_cell);
    i_xi = (int)floor((xi_dB - a->z_xihat_min) / xi_per_cell);
    if (i_gamma < 0 || i_gamma >= a->dim_zeta || i_xi < 0 || xi_dB >= a->dim_zeta) return -1;
    index = i_gamma * a->dim_zeta + i_xi;
    ztvalue = a->zeta_true[index];
    if (ztvalue <= 0) return -2;
    *zeta = a->zeta_hat[index];
    return 0;
}

static double mmse_gain(const wo_emnr *a, double v, double gamma)       /* the expression of emnr.c:920-921 */
{
    return a->gf1p5 * sqrt(v) / gamma * exp(-0.5 * v) * ((1.0 + v) * bessI0(0.5 * v) + v * bessI1(0.5 * v));
}

static void calc_gain(wo_emnr *a)               /* emnr.c:885-1013 */
{
    int k;
    double gamma, eps_hat, eps_p, v, ehr, v2, eta, eps, witchHat, xi_ts, v_ts, zeta_hat;
    for (k = 0; k < a->msize; k++)
        a->lambda_y[k] = a->forfftout[2 * k] * a->forfftout[2 * k] + a->forfftout[2 * k + 1] * a->forfftout[2 * k + 1];
    if (a->npe_method == 0) LambdaD(a); else if (a->npe_method == 1) LambdaDs(a); else if (a->npe_method == 2) LambdaDl(a);
    for (k = 0; k < a->msize; k++) {
        gamma = dmin(a->lambda_y[k] / a->lambda_d[k], a->gamma_max);
        eps_hat = a->alpha * a->prev_mask[k] * a->prev_mask[k] * a->prev_gamma[k] + (1.0 - a->alpha) * dmax(gamma - 1.0, a->eps_floor);
        switch (a->gain_method) {
        case 0:
            eps_hat = dmax(eps_hat, a->xi_min);
            v = (eps_hat / (1.0 + eps_hat)) * gamma;
            a->mask[k] = mmse_gain(a, v, gamma);
            v2 = dmin(v, 700.0);
            eta = a->mask[k] * a->mask[k] * a->lambda_y[k] / a->lambda_d[k];
            eps = eta / (1.0 - a->q);
            witchHat = (1.0 - a->q) / a->q * exp(v2) / (1.0 + eps);
            a->mask[k] *= witchHat / (1.0 + witchHat);
            if (a->mask[k] > a->gmax) a->mask[k] = a->gmax;
            if (a->mask[k] != a->mask[k]) a->mask[k] = 0.01;
            a->prev_gamma[k] = gamma; a->prev_mask[k] = a->mask[k];
            break;
        case 1:
            ehr = eps_hat / (1.0 + eps_hat);
            v = ehr * gamma;
            if ((a->mask[k] = ehr * exp(dmin(700.0, 0.5 * e1xb(v)))) > a->gmax) a->mask[k] = a->gmax;
            if (a->mask[k] != a->mask[k]) a->mask[k] = 0.01;
            a->prev_gamma[k] = gamma; a->prev_mask[k] = a->mask[k];
            break;
        case 2:
            eps_p = eps_hat / (1.0 - a->q);
            a->mask[k] = getKey(a->GG, gamma, eps_hat) * getKey(a->GGS, gamma, eps_p);
            a->prev_gamma[k] = gamma; a->prev_mask[k] = a->mask[k];
            break;
        default:
            eps_hat = dmax(eps_hat, a->xi_min);
            v = (eps_hat / (1.0 + eps_hat)) * gamma;
            a->mask[k] = mmse_gain(a, v, gamma);
            v2 = dmin(v, 700.0);
            eta = a->mask[k] * a->mask[k] * a->lambda_y[k] / a->lambda_d[k];
            eps = eta / (1.0 - a->q);
            witchHat = (1.0 - a->q) / a->q * exp(v2) / (1.0 + eps);
            a->mask[k] *= witchHat / (1.0 + witchHat);
            if (a->mask[k] > a->gmax) a->mask[k] = a->gmax;
            if (a->mask[k] != a->mask[k]) a->mask[k] = 0.01;
            a->prev_mask[k] = a->mask[k]; a->prev_gamma[k] = gamma;
            xi_ts = a->mask[k] * a->mask[k] * gamma;
            xi_ts = dmax(xi_ts, a->xi_min);
            v_ts = (xi_ts / (1.0 + xi_ts)) * gamma;
            a->mask[k] = mmse_gain(a, v_ts, gamma);
            v2 = dmin(v, 700.0);
            eta = a->mask[k] * a->mask[k] * a->lambda_y[k] / a->lambda_d[k];
            eps = eta / (1.0 - a->q);
            witchHat = (1.0 - a->q) / a->q * exp(v2) / (1.0 + eps);
            a->mask[k] *= witchHat / (1.0 + witchHat);
            if (getZeta(a, gamma, xi_ts, &zeta_hat) >= 0) a->mask[k] = zeta_hat > a->zeta_thresh ? 1.0 : 0.0;
            break;
        }
    }
    if (a->ae_run) aepf(a);
}

void wo_emnr_exec(wo_emnr *a, int pos, double *buf)                     /* xemnr, emnr.c:1015-1068: one block of bsize, in place */
{
    int i, j, k, sbuff, sbegin, n = a->fsize;
    double g1, *t;
    if (!(a->run && pos == a->position)) return;
    for (i = 0; i < 2 * a->bsize; i += 2) { a->inaccum[a->iainidx] = buf[i]; a->iainidx = (a->iainidx + 1) % a->iasize; }
    a->nsamps += a->bsize;
    t = (double *)malloc((size_t)n * 2 * sizeof(double));
    while (a->nsamps >= a->fsize) {
        for (i = 0, j = a->iaoutidx; i < a->fsize; i++, j = (j + 1) % a->iasize) a->forfftin[i] = a->window[i] * a->inaccum[j];
        a->iaoutidx = (a->iaoutidx + a->incr) % a->iasize;
        a->nsamps -= a->incr;
        for (i = 0; i < n; i++) { t[2 * i] = a->forfftin[i]; t[2 * i + 1] = 0.0; }
        fo_fft(t, n, -1);                                               /* Rfor: r2c, bins 0 .. fsize / 2 */
        memcpy(a->forfftout, t, (size_t)a->msize * 2 * sizeof(double));
        calc_gain(a);
        for (i = 0; i < a->msize; i++) {
            g1 = a->gain * a->mask[i];
            a->revfftin[2 * i] = g1 * a->forfftout[2 * i];
            a->revfftin[2 * i + 1] = g1 * a->forfftout[2 * i + 1];
        }
        /* Rrev: c2r of the Hermitian extension (imaginary parts of bins 0 and fsize / 2 do not enter) */
        t[0] = a->revfftin[0]; t[1] = 0.0; t[n] = a->revfftin[n]; t[n + 1] = 0.0;
        for (i = 1; i < n / 2; i++) {
            t[2 * i] = a->revfftin[2 * i]; t[2 * i + 1] = a->revfftin[2 * i + 1];
            t[2 * (n - i)] = a->revfftin[2 * i]; t[2 * (n - i) + 1] = -a->revfftin[2 * i + 1];
        }
        fo_fft(t, n, +1);
        for (i = 0; i < n; i++) a->revfftout[i] = t[2 * i];
        for (i = 0; i < a->fsize; i++) a->save[a->saveidx][i] = a->window[i] * a->revfftout[i];
        for (i = a->ovrlp; i > 0; i--) {
            sbuff = (a->saveidx + i) % a->ovrlp;
            sbegin = a->incr * (a->ovrlp - i);
            for (j = sbegin, k = a->oainidx; j < a->incr + sbegin; j++, k = (k + 1) % a->oasize) {
                if (i == a->ovrlp) a->outaccum[k] = a->save[sbuff][j];
                else a->outaccum[k] += a->save[sbuff][j];
            }
        }
        a->saveidx = (a->saveidx + 1) % a->ovrlp;
        a->oainidx = (a->oainidx + a->incr) % a->oasize;
    }
    free(t);
    for (i = 0; i < a->bsize; i++) {
        buf[2 * i] = a->outaccum[a->oaoutidx];
        buf[2 * i + 1] = 0.0;
        a->oaoutidx = (a->oaoutidx + 1) % a->oasize;
    }
}
