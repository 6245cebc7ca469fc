/* fft_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). See fft_oracle.h. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fft_oracle.h"

#define FO_MAXLOG 24

/* twiddle cache: one table per log2(n), tw[k] = exp(-2*pi*i*k/n), k < n/2 */
static double *fo_tw[FO_MAXLOG + 1];

static const double *fo_twiddles(int logn)
{
    if (!fo_tw[logn]) {
        int n = 1 << logn, k;
        double *t = (double *)malloc(sizeof(double) * (size_t)n);
        for (k = 0; k < n / 2; k++) {
            /* long double keeps the table at <=0.5 ulp of double */
            long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
            t[2 * k + 0] = (double)cosl(a);
            t[2 * k + 1] = (double)sinl(a);
        }
        fo_tw[logn] = t;
    }
    return fo_tw[logn];
}

/* Sizes that are not a power of two (Quisk's panadapter: fft_size = data_width * fft_mult, quisk.py:186-194,4179): the
 * definition itself, X[k] = sum_n x[n] exp(sign 2 pi i n k / N), with the phase index n k mod N kept exact. */
static void fo_dft_direct(double *x, int n, int sign)
{
    double *c = (double *)malloc(sizeof(double) * 2 * (size_t)n), *y = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    int k, m;
    for (m = 0; m < n; m++) {
        long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
        c[2 * m] = (double)cosl(a); c[2 * m + 1] = (double)(sign < 0 ? -sinl(a) : sinl(a));
    }
    for (k = 0; k < n; k++) {
        long double sr = 0.0L, si = 0.0L;
        long long idx = 0;
        for (m = 0; m < n; m++) {
            sr += (long double)x[2 * m] * c[2 * idx] - (long double)x[2 * m + 1] * c[2 * idx + 1];
            si += (long double)x[2 * m] * c[2 * idx + 1] + (long double)x[2 * m + 1] * c[2 * idx];
            idx += k; if (idx >= n) idx -= n;
        }
        y[2 * k] = (double)sr; y[2 * k + 1] = (double)si;
    }
    memcpy(x, y, sizeof(double) * 2 * (size_t)n);
    free(c); free(y);
}

void fo_fft(double *x, int n, int sign)
{
    int logn = 0, i, j, len;
    const double *tw;
    if (n > 1 && (n & (n - 1))) { fo_dft_direct(x, n, sign); return; }
    while ((1 << logn) < n) logn++;
    if (n <= 1) return;
    tw = fo_twiddles(logn);
    /* bit reversal */
    for (i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double tr = x[2 * i], ti = x[2 * i + 1];
            x[2 * i] = x[2 * j]; x[2 * i + 1] = x[2 * j + 1];
            x[2 * j] = tr; x[2 * j + 1] = ti;
        }
    }
    for (len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len, k;
        for (i = 0; i < n; i += len) {
            for (k = 0; k < half; k++) {
                double wr = tw[2 * (k * step)], wi = tw[2 * (k * step) + 1];
                double ar, ai, br, bi, tr, ti;
                if (sign > 0) wi = -wi;
                ar = x[2 * (i + k)]; ai = x[2 * (i + k) + 1];
                br = x[2 * (i + k + half)]; bi = x[2 * (i + k + half) + 1];
                tr = br * wr - bi * wi;
                ti = br * wi + bi * wr;
                x[2 * (i + k)] = ar + tr; x[2 * (i + k) + 1] = ai + ti;
                x[2 * (i + k + half)] = ar - tr; x[2 * (i + k + half) + 1] = ai - ti;
            }
        }
    }
}

void fo_fft_oop(const double *in, double *out, int n, int sign)
{
    if (in != out) memcpy(out, in, sizeof(double) * 2 * (size_t)n);
    fo_fft(out, n, sign);
}
