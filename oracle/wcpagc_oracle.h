/* wcpagc_oracle.h -- TEST INFRASTRUCTURE ONLY.  Restatement of WDSP's AGC (wdsp/wcpAGC.c:29-342), modes 0-5:
 * look-ahead ring, running maximum, 5-state attack / fast-decay / hang / decay machine, log-slope gain.
 * PARITY UNPINNED by reference execution (wdsp needs <fftw3.h>); checked against an independent numpy
 * transcription of the state machine in tests/test_oracle_wdsp.py. */
#ifndef WCPAGC_ORACLE_H
#define WCPAGC_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct wo_agc {
    int run, mode, pmode, n_tau, hang_enable;
    double sample_rate, tau_attack, tau_decay, max_gain, var_gain, fixed_gain, max_input, out_targ;
    double tau_fast_backaverage, tau_fast_decay, pop_ratio, tau_hang_backmult, hangtime, hang_thresh, tau_hang_decay;
    /* derived, loadWcpAGC */
    int ring_buffsize, attack_buffsize, in_index, out_index, hang_counter, decay_type, state;
    double attack_mult, decay_mult, fast_decay_mult, fast_backmult, onemfast_backmult, out_target, min_volts, inv_out_target;
    double slope_constant, inv_max_input, hang_level, hang_backmult, onemhang_backmult, hang_decay_mult;
    double ring_max, volts, save_volts, fast_backaverage, hang_backaverage, gain;
    double *ring, *abs_ring;
} wo_agc;

void wo_agc_init(wo_agc *a, int run, int mode, int pmode, int sample_rate, double tau_attack, double tau_decay, int n_tau,
                 double max_gain, double var_gain, double fixed_gain, double max_input, double out_targ,
                 double tau_fast_backaverage, double tau_fast_decay, double pop_ratio, int hang_enable,
                 double tau_hang_backmult, double hangtime, double hang_thresh, double tau_hang_decay);   /* create_wcpagc */
void wo_agc_free(wo_agc *a);
void wo_agc_load(wo_agc *a);                                    /* loadWcpAGC, wcpAGC.c:115-146 */
void wo_agc_set_mode(wo_agc *a, int mode);                      /* SetRXAAGCMode, wcpAGC.c:369-411 */
void wo_agc_exec(wo_agc *a, double *buf, int size);             /* xwcpagc, in place, wcpAGC.c:161-342 */
#ifdef __cplusplus
}
#endif
#endif
