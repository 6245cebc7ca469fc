/* snba_oracle.h -- TEST INFRASTRUCTURE ONLY.  WDSP's spectral noise blanker (wdsp/snb.c, with the linear-algebra helpers
 * of wdsp/lmath.c) for one channel; see snba_oracle.c.  PARITY UNPINNED by reference execution (wdsp needs <fftw3.h>). */
#ifndef SNBA_ORACLE_H
#define SNBA_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct wo_snba wo_snba;
/* create_snba as create_rxa calls it (RXA.c:237-255): 12 kHz inside, frames of 256 with overlap 4, LPC order 64 */
wo_snba *wo_snba_create(int rate, int bsize);
void wo_snba_free(wo_snba *d);
void wo_snba_flush(wo_snba *d);                                         /* flush_snba, snb.c:161-185 */
int *wo_snba_run(wo_snba *d);
void wo_snba_set_output_bandwidth(wo_snba *d, double flow, double fhigh);  /* SetRXASNBAOutputBandwidth, snb.c:660-694 */
void wo_snba_set_ovrlp(wo_snba *d, int ovrlp);              /* SetRXASNBAovrlp, snb.c:595-603 */
void wo_snba_set_tuning(wo_snba *d, int which, double v);   /* SetRXASNBAasize .. pmultmin, snb.c:604-658: which = 0 asize, 1 npasses, 2 k1, 3 k2, 4 bridge, 5 presamps, 6 postsamps, 7 pmultmin */
void wo_snba_exec(wo_snba *d, double *buf);                             /* xsnba on one block of bsize complex samples, in place */
/* the pieces, for the unit tests */
void wo_snba_asolve(int xsize, int asize, const double *x, double *a);  /* asolve, lmath.c:96-127; reads x[-asize .. xsize-1] */
void wo_snba_median(int n, double *a, double *med);                      /* median, lmath.c:129-186 */
void wo_snba_frame(wo_snba *d, double *x);                               /* execFrame, snb.c:492-537; x has xsize samples of history in front */
int wo_snba_xsize(const wo_snba *d);
#ifdef __cplusplus
}
#endif
#endif
