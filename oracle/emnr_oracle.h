/* emnr_oracle.h -- TEST INFRASTRUCTURE ONLY.  WDSP's EMNR (wdsp/emnr.c) for one channel; see emnr_oracle.c.
 * PARITY UNPINNED by reference execution (wdsp needs <fftw3.h>). */
#ifndef EMNR_ORACLE_H
#define EMNR_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct wo_emnr wo_emnr;
wo_emnr *wo_emnr_create(int bsize, int rate);                   /* create_emnr as create_rxa calls it, RXA.c:319-332 */
void wo_emnr_free(wo_emnr *a);
/* GG, GGS: 241 x 241 (the `calculus` data, emnr.c:317-326); zeta_hat / zeta_true: 60 x 60 with their ranges (zetaHat.bin) */
void wo_emnr_set_tables(wo_emnr *a, const double *GG, const double *GGS, const double *zeta_hat, const int *zeta_true, double gmin,
                        double gmax, double ximin, double ximax);
void wo_emnr_flush(wo_emnr *a);
int *wo_emnr_run(wo_emnr *a);
int *wo_emnr_position(wo_emnr *a);
void wo_emnr_set_gain_method(wo_emnr *a, int m);                /* SetRXAEMNRgainMethod, emnr.c:1112 */
void wo_emnr_set_npe_method(wo_emnr *a, int m);                 /* SetRXAEMNRnpeMethod, emnr.c:1120; 0, 1, 2 */
void wo_emnr_set_scalars(wo_emnr *a, double ae_zeta_thresh, double ae_psi, double train_zeta_thresh, double train_t2);
void wo_emnr_set_ae_run(wo_emnr *a, int run);                   /* SetRXAEMNRaeRun, emnr.c:1128 */
void wo_emnr_exec(wo_emnr *a, int pos, double *buf);            /* xemnr on one block of bsize complex samples, in place */
#ifdef __cplusplus
}
#endif
#endif
