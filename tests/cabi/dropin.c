/* tests/cabi/dropin.c -- the drop-in boundary from a C translation unit (tests/test_gpu_cabi_dropin.py compiles it with gcc -std=c99,
 * links -lquiskhip and runs it on the GPU box).  What ctypes cannot see: the layouts of struct quisk_cFilter / quisk_cHB45Filter as a C
 * compiler lays them out from include/quiskhip.h (filter.h:1-37), `complex double` arguments, and quisk_dC_out's complex double returned
 * BY VALUE (filter.c:83-104, microphone.c:469).
 *
 *   dropin <dir>      <dir> holds the little-endian binaries the test wrote (inputs from seeds, expectations from
 *                     tests/golden/filter_golden.npz and the CPU restatement); prints one line per case, exit code = failed cases
 *
 * Cases: quisk_cDecim2HB45 and quisk_cDecimate (98 taps, / 2) in ragged calls against the golden vectors; Quisk's 192 ksps plan --
 * HB45 then the 98-tap / 2 (quisk.c:1769-1833) -- on one stream; quisk_dC_out by value against qh_quisk_dC_out through a pointer and
 * against its expectation; wdspFexchange0 (quisk_wdsp.c:24-69) on an SSB channel opened through the WDSP names. */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "quiskhip.h"

static char g_dir[1024];

static void *load(const char *name, size_t elem, long *count)
{
    char path[1200];
    FILE *f;
    long bytes;
    void *p;
    snprintf(path, sizeof path, "%s/%s", g_dir, name);
    f = fopen(path, "rb");
    if (!f) { *count = -1; return NULL; }
    fseek(f, 0, SEEK_END); bytes = ftell(f); fseek(f, 0, SEEK_SET);
    p = malloc(bytes > 0 ? (size_t)bytes : 1);
    if (fread(p, 1, (size_t)bytes, f) != (size_t)bytes) { fclose(f); free(p); *count = -1; return NULL; }
    fclose(f);
    *count = bytes / (long)elem;
    return p;
}

static double rel_rms_c(const complex double *a, const complex double *b, long n)
{
    double num = 0.0, den = 0.0;
    long i;
    for (i = 0; i < n; i++) {
        const complex double d = a[i] - b[i];
        num += creal(d) * creal(d) + cimag(d) * cimag(d);
        den += creal(b[i]) * creal(b[i]) + cimag(b[i]) * cimag(b[i]);
    }
    return den > 0.0 ? sqrt(num / den) : sqrt(num);
}

static int report(const char *what, long got, long want, double err, double tol)
{
    const int ok = got == want && err <= tol;
    printf("%-44s %s  count %ld / %ld  rel rms %.3e (<= %.0e)%s%s\n", what, ok ? "ok  " : "FAIL", got, want, err, tol,
           ok ? "" : "  last error: ", ok ? "" : qh_last_error());
    return ok ? 0 : 1;
}

int main(int argc, char **argv)
{
    long nx, nt, ns, ne, i, pos, k;
    int fails = 0;
    complex double *xc, *work, *got;
    double *t98;
    long long *splits;
    setvbuf(stdout, NULL, _IONBF, 0);
    if (argc < 2) { fprintf(stderr, "usage: dropin <dir>\n"); return 99; }
    snprintf(g_dir, sizeof g_dir, "%s", argv[1]);
    if (qh_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 98; }
    xc = (complex double *)load("xc.bin", sizeof(complex double), &nx);
    t98 = (double *)load("taps98.bin", sizeof(double), &nt);
    splits = (long long *)load("splits.bin", sizeof(long long), &ns);
    if (nx <= 0 || nt != 98 || ns <= 0) { fprintf(stderr, "bad inputs in %s\n", g_dir); return 97; }
    work = (complex double *)malloc((size_t)nx * sizeof *work);
    got = (complex double *)malloc((size_t)nx * sizeof *got);

    {   /* ---- quisk_cDecim2HB45, ragged calls (filter.c:377-417) */
        struct quisk_cHB45Filter hb;
        complex double *want = (complex double *)load("expect_hb45.bin", sizeof(complex double), &ne);
        long n = 0;
        memset(&hb, 0, sizeof hb);
        for (pos = 0, k = 0; k < ns; k++) {
            const int cnt = (int)splits[k];
            int m;
            memcpy(work, xc + pos, (size_t)cnt * sizeof *work);
            m = quisk_cDecim2HB45((double *)work, cnt, &hb);
            memcpy(got + n, work, (size_t)m * sizeof *work);
            n += m; pos += cnt;
        }
        fails += report("quisk_cDecim2HB45 (golden)", n, ne, ne > 0 && n == ne ? rel_rms_c(got, want, n) : 1.0, 1e-12);
        free(hb.cBuf); free(want);
    }
    {   /* ---- quisk_cDecimate, 98 taps, / 2, ragged calls (filter.c:203-229) */
        struct quisk_cFilter f;
        complex double *want = (complex double *)load("expect_dec98.bin", sizeof(complex double), &ne);
        long n = 0;
        memset(&f, 0, sizeof f);
        quisk_filt_cInit(&f, t98, 98);
        if (f.nTaps != 98 || !f.cSamples || f.ptcSamp != f.cSamples || f.decim_index != 0 || f.dCoefs != t98) { printf("quisk_filt_cInit: struct fields FAIL\n"); fails++; }
        for (pos = 0, k = 0; k < ns; k++) {
            const int cnt = (int)splits[k];
            int m;
            memcpy(work, xc + pos, (size_t)cnt * sizeof *work);
            m = quisk_cDecimate((double *)work, cnt, &f, 2);
            memcpy(got + n, work, (size_t)m * sizeof *work);
            n += m; pos += cnt;
        }
        fails += report("quisk_cDecimate 98 taps / 2 (golden)", n, ne, ne > 0 && n == ne ? rel_rms_c(got, want, n) : 1.0, 1e-12);
        /* the state the reference would have left: the write pointer inside the ring, the decimation phase in [0, 2) */
        if (f.ptcSamp < f.cSamples || f.ptcSamp >= f.cSamples + 2 * 98 || f.decim_index < 0 || f.decim_index > 1) { printf("quisk_cDecimate: state FAIL\n"); fails++; }
        free(f.cSamples); free(f.cBuf); free(want);
    }
    {   /* ---- Quisk's plan for 192 ksps: HB45 to 96 k, then the 98-tap filter / 2 to 48 k (quisk.c:1769-1833), blocks of 1024 */
        struct quisk_cHB45Filter hb;
        struct quisk_cFilter f;
        complex double *want = (complex double *)load("expect_plan192.bin", sizeof(complex double), &ne);
        long n = 0;
        memset(&hb, 0, sizeof hb); memset(&f, 0, sizeof f);
        quisk_filt_cInit(&f, t98, 98);
        for (pos = 0; pos < nx; pos += 1024) {
            int cnt = (int)(nx - pos < 1024 ? nx - pos : 1024), m;
            memcpy(work, xc + pos, (size_t)cnt * sizeof *work);
            m = quisk_cDecim2HB45((double *)work, cnt, &hb);
            m = quisk_cDecimate((double *)work, m, &f, 2);
            memcpy(got + n, work, (size_t)m * sizeof *work);
            n += m;
        }
        fails += report("192 k plan: HB45 + 98 taps / 2", n, ne, ne > 0 && n == ne ? rel_rms_c(got, want, n) : 1.0, 1e-12);
        free(hb.cBuf); free(f.cSamples); free(f.cBuf); free(want);
    }
    {   /* ---- quisk_dC_out: complex double by value (filter.c:83-104) */
        struct quisk_cFilter fa, fb;
        long nr;
        double *xr = (double *)load("xr.bin", sizeof(double), &nr);
        complex double *want = (complex double *)load("expect_dcout.bin", sizeof(complex double), &ne);
        double worst = 0.0;
        memset(&fa, 0, sizeof fa); memset(&fb, 0, sizeof fb);
        quisk_filt_dInit(&fa, t98, 98); quisk_filt_dInit(&fb, t98, 98);       /* (a dFilter: the history ring holds doubles) */
        quisk_filt_tune(&fa, 0.0625, 1); quisk_filt_tune(&fb, 0.0625, 1);
        for (i = 0; i < nr && i < ne; i++) {
            double re_im[2];
            const complex double v = quisk_dC_out(xr[i], &fa);
            qh_quisk_dC_out(xr[i], &fb, re_im);
            got[i] = v;
            if (fabs(creal(v) - re_im[0]) > worst) worst = fabs(creal(v) - re_im[0]);
            if (fabs(cimag(v) - re_im[1]) > worst) worst = fabs(cimag(v) - re_im[1]);
        }
        fails += report("quisk_dC_out by value (vs restatement)", i, ne, ne > 0 ? rel_rms_c(got, want, ne) : 1.0, 1e-12);
        if (worst != 0.0) { printf("quisk_dC_out by value vs qh_quisk_dC_out: differ by %.3e FAIL\n", worst); fails++; }
        free(fa.cSamples); free(fb.cSamples); free(fa.cpxCoefs); free(fb.cpxCoefs); free(xr); free(want);
    }
    {   /* ---- wdspFexchange0 (quisk_wdsp.c:24-69) on a channel opened through the WDSP names: 48 k in, DSP and out, USB */
        long nw;
        complex double *xw = (complex double *)load("wdsp_in.bin", sizeof(complex double), &nw);
        complex double *want = (complex double *)load("expect_wdsp.bin", sizeof(complex double), &ne);
        free(work); free(got);                  /* (a call may hand back more than it was given: a block that was waiting) */
        work = (complex double *)malloc((size_t)(nw + 2048) * sizeof *work);
        got = (complex double *)malloc((size_t)(nw + 2048) * sizeof *got);
        const int ch = 5, in_size = 512, piece = 700;           /* pieces that are not the block size: the shim re-blocks */
        long n = 0;
        OpenChannel(ch, in_size, 256, 48000, 48000, 48000, 0, 1, 0.010, 0.025, 0.0, 0.010, 1);      /* Quisk's rates (quisk.py: 48 k throughout) */
        SetRXAShiftRun(ch, 1); SetRXAShiftFreq(ch, -9000.0); RXANBPSetRun(ch, 1); SetRXAMode(ch, 1); RXASetPassband(ch, 300.0, 3000.0);
        SetRXAAGCMode(ch, 0); SetRXAAGCFixed(ch, 0.0);
        qh_wdsp_set_parameter(ch, in_size, 1);
        for (pos = 0; pos < nw; pos += piece) {
            const int cnt = (int)(nw - pos < piece ? nw - pos : piece);
            int m;
            memcpy(work, xw + pos, (size_t)cnt * sizeof *work);
            m = wdspFexchange0(ch, (double *)work, cnt);
            memcpy(got + n, work, (size_t)m * sizeof *work);
            n += m;
        }
        CloseChannel(ch);
        fails += report("wdspFexchange0, 700-sample pieces", n, ne, ne > 0 && n == ne ? rel_rms_c(got, want, n) : 1.0, 1e-9);
        free(xw); free(want);
    }
    free(xc); free(t98); free(splits); free(work); free(got);
    printf("%d case(s) failed\n", fails);
    return fails;
}
