/* wdsp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, double precision, reentrant) of the WDSP RXA receive
 * chain that Quisk drives through quisk_wdsp.c / fexchange0():
 *
 *   shift -> resample(in) -> nbp0 (partitioned overlap-save fircore) -> amd/fmd
 *         -> bp1 -> agc (fixed gain) -> panel -> resample(out)
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).  It follows the REFERENCE's algorithms (rotation-recurrence NCO,
 * ring-buffer polyphase resampler, uniformly partitioned overlap-save), not the GPU
 * library's, so the two are algorithmically independent.
 *
 * PARITY UNPINNED by reference execution: every wdsp/ source includes <fftw3.h>
 * (wdsp/comm.h:55), FFTW3 is not vendored and not in this image, and the rules
 * forbid stand-in headers, so wdsp/ cannot be built here.  The restatement is
 * checked instead against (a) independent closed-form / numpy-scipy computations
 * (tests/test_oracle_wdsp.py) and (b) the behaviours probed from the real reference
 * and recorded in SURVEY.md section 8 (latency 2*dsp blocks, first non-zero output
 * index 993, steady-state gain 4.0, passband sign convention).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef WDSP_ORACLE_H
#define WDSP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* wdsp/RXA.h:31-45 */
enum { WO_LSB = 0, WO_USB, WO_DSB, WO_CWL, WO_CWU, WO_FM, WO_AM, WO_DIGU, WO_SPEC, WO_DIGL, WO_SAM, WO_DRM };

typedef struct wo_channel wo_channel;

/* OpenChannel(), wdsp/channel.c:75-103 (type 0 = RXA, state 1, bfo 1) */
wo_channel *wo_open(int in_size, int dsp_size, int in_rate, int dsp_rate, int out_rate,
                    double tdelayup, double tslewup, double tdelaydown, double tslewdown);
void wo_close(wo_channel *c);

/* fexchange0(), wdsp/iobuffs.c:464-516 with the DSP thread (wdsp/main.c:29-66) run synchronously */
void wo_fexchange0(wo_channel *c, const double *in, double *out, int *error);

/* xrxa() on one DSP block without the iobuffs latency/slew: in = dsp_insize, out = dsp_outsize complex */
void wo_xrxa_block(wo_channel *c, const double *in, double *out);
void wo_xrxa_blocks(wo_channel *c, const double *in, double *out, int nblk);

int wo_dsp_insize(const wo_channel *c);
int wo_dsp_outsize(const wo_channel *c);
int wo_out_size(const wo_channel *c);

/* setters: same meaning as the WDSP exports of the same name */
void wo_SetRXAMode(wo_channel *c, int mode);                    /* RXA.c:748-787 */
void wo_RXASetPassband(wo_channel *c, double f_low, double f_high); /* RXA.c:926-932 */
void wo_RXASetNC(wo_channel *c, int nc);                        /* RXA.c:934-946 */
void wo_SetRXAShiftRun(wo_channel *c, int run);                 /* shift.c:110-117 */
void wo_SetRXAShiftFreq(wo_channel *c, double fshift);          /* shift.c:119-127 */
void wo_RXANBPSetRun(wo_channel *c, int run);                   /* nbp.c:528-536 */
void wo_RXANBPSetFreqs(wo_channel *c, double flow, double fhigh); /* nbp.c:538-552 */
void wo_SetRXABandpassRun(wo_channel *c, int run);              /* bandpass.c:381-387 */
void wo_SetRXABandpassFreqs(wo_channel *c, double f_low, double f_high); /* bandpass.c:389-407 */
void wo_SetRXAAGCMode(wo_channel *c, int mode);                 /* wcpAGC.c:369-411 */
void wo_SetRXAAGCFixed(wo_channel *c, double fixed_agc_db);     /* wcpAGC.c:541-548 */
void wo_SetRXAAGCAttack(wo_channel *c, int attack_ms);          /* wcpAGC.c:413-420 */
void wo_SetRXAAGCDecay(wo_channel *c, int decay_ms);            /* wcpAGC.c:422-429 */
void wo_SetRXAAGCHang(wo_channel *c, int hang_ms);              /* wcpAGC.c:431-438 */
void wo_SetRXAAGCTop(wo_channel *c, double max_agc_db);         /* wcpAGC.c:520-527 */
void wo_SetRXAAGCSlope(wo_channel *c, int slope);               /* wcpAGC.c:529-536 */
void wo_SetRXAAGCHangThreshold(wo_channel *c, int threshold);   /* wcpAGC.c:480-487 */
void wo_SetRXAPanelRun(wo_channel *c, int run);                 /* patchpanel.c:123-129 */
void wo_SetRXAPanelGain1(wo_channel *c, double gain);
void wo_SetRXAPanelGain2(wo_channel *c, double gainI, double gainQ);
void wo_SetRXAPanelSelect(wo_channel *c, int select);
void wo_SetRXAPanelCopy(wo_channel *c, int copy);
void wo_SetRXAAMDSBMode(wo_channel *c, int sbmode);             /* amd.c:259-265 */
void wo_SetRXAAMDRun(wo_channel *c, int run);                   /* amd.c:264-277 */
void wo_SetRXASNBARun(wo_channel *c, int run);                  /* snb.c:579-593 */
void wo_SetRXASNBAovrlp(wo_channel *c, int ovrlp);              /* snb.c:595-603 */
void wo_SetRXASNBATuning(wo_channel *c, int which, double v);   /* snb.c:604-658, `which` as in wo_snba_set_tuning */
void wo_SetRXAEMNRRun(wo_channel *c, int run);                  /* emnr.c:1096-1110 */
void wo_SetRXAEMNRgainMethod(wo_channel *c, int method);        /* emnr.c:1112-1118 */
void wo_SetRXAEMNRnpeMethod(wo_channel *c, int method);         /* emnr.c:1120-1126 */
void wo_SetRXAEMNRaeRun(wo_channel *c, int run);                /* emnr.c:1128-1134 */
void wo_SetRXAEMNRaeZetaThresh(wo_channel *c, double v);       /* emnr.c:1145-1174 */
void wo_SetRXAEMNRaePsi(wo_channel *c, double v);
void wo_SetRXAEMNRtrainZetaThresh(wo_channel *c, double v);
void wo_SetRXAEMNRtrainT2(wo_channel *c, double v);
void wo_SetRXAEMNRPosition(wo_channel *c, int position);        /* emnr.c:1136-1143 */
void wo_SetEMNRTables(wo_channel *c, const double *GG, const double *GGS, const double *zeta_hat, const int *zeta_true, double gmin, double gmax,
                      double ximin, double ximax);              /* the run-time data of emnr.c:317-334 */
void wo_SetRXAAMSQRun(wo_channel *c, int run);                  /* amsq.c:216-222 */
void wo_SetRXAAMSQThreshold(wo_channel *c, double threshold);   /* amsq.c:224-232 */
void wo_SetRXAAMSQMaxTail(wo_channel *c, double tail);          /* amsq.c:234-243 */
void wo_SetRXAANFRun(wo_channel *c, int run);                   /* anf.c:175-189 */
void wo_SetRXAANFVals(wo_channel *c, int taps, int delay, double gain, double leakage);     /* anf.c:191-201 */
void wo_SetRXAANFPosition(wo_channel *c, int position);         /* anf.c:231-239 */
void wo_SetRXAANRRun(wo_channel *c, int run);                   /* anr.c:175-189 */
void wo_SetRXAANRVals(wo_channel *c, int taps, int delay, double gain, double leakage);     /* anr.c:191-201 */
void wo_SetRXAANRPosition(wo_channel *c, int position);         /* anr.c:231-238 */
void wo_RXASetMP(wo_channel *c, int mp);                        /* RXA.c:948-958 */
void wo_SetRXAFMLimRun(wo_channel *c, int run);                 /* fmd.c:336-347 */
void wo_SetRXAFMLimGain(wo_channel *c, double gaindB);          /* fmd.c:349-362 */
/* notch database, nbp.c:358-525; make_nbp nbp.c:97-179 */
int wo_RXANBPAddNotch(wo_channel *c, int notch, double fcenter, double fwidth, int active);
int wo_RXANBPDeleteNotch(wo_channel *c, int notch);
int wo_RXANBPEditNotch(wo_channel *c, int notch, double fcenter, double fwidth, int active);
void wo_RXANBPSetTuneFrequency(wo_channel *c, double tunefreq);
void wo_RXANBPSetShiftFrequency(wo_channel *c, double shift);
void wo_RXANBPSetNotchesRun(wo_channel *c, int run);
void wo_RXANBPSetWindow(wo_channel *c, int wintype);
void wo_RXANBPSetAutoIncrease(wo_channel *c, int autoincr);
int wo_make_nbp(int nn, const int *active, const double *center, const double *width, const double *nlow, const double *nhigh,
                double minwidth, int autoincr, double flow, double fhigh, double *bplow, double *bphigh, int *havnotch);
void wo_mp_imp(int N, const double *fir, double *mpfir, int pfactor, int polarity);     /* fir.c:319-368 */
void wo_SetRXAAMDFadeLevel(wo_channel *c, int levelfade);       /* amd.c:267-273 */
void wo_SetRXAFMDeviation(wo_channel *c, double deviation);     /* fmd.c:236-246 */
void wo_SetRXACTCSSFreq(wo_channel *c, double freq);            /* fmd.c:248-258 */
void wo_SetRXACTCSSRun(wo_channel *c, int run);                 /* fmd.c:260-267 */
double wo_GetRXAMeter(wo_channel *c, int mt);                   /* meter.c:133-142 */

/* building blocks, exported for unit tests */
double *wo_fir_bandpass(int N, double f_low, double f_high, double samplerate, int wintype, int rtype, double scale); /* fir.c:187-254 */
/* calc_resample(), resample.c:35-78: returns malloc'd h[ncoef] in phase-major order; fills L, M, ncoef, cpp */
double *wo_calc_resample_taps(int in_rate, int out_rate, double fc, int ncoef_in, double gain, int *L, int *M, int *ncoef, int *cpp);
/* stand-alone fircore (wdsp/firmin.c:290-430) */
typedef struct wo_fircore wo_fircore;
wo_fircore *wo_fircore_create(int size, int nc, const double *impulse);
void wo_fircore_destroy(wo_fircore *a);
void wo_fircore_exec(wo_fircore *a, const double *in, double *out); /* size complex in -> size complex out */
/* stand-alone resampler (wdsp/resample.c:80-157) */
typedef struct wo_resample wo_resample;
wo_resample *wo_resample_create(int in_rate, int out_rate, double fc, int ncoef, double gain);
void wo_resample_destroy(wo_resample *a);
int wo_resample_exec(wo_resample *a, const double *in, int size, double *out);

/* Quisk's re-blocking shim in front of fexchange0 (quisk_wdsp.c:12-91); fn(ctx, in, out, &error) stands for the
 * (*wdsp_fexchange0)(channel, in, out, &error) pointer Quisk is handed through QS.wdsp_set_parameter */
typedef struct wo_shim {
    double *cBuf;       /* circular sample buffer, interleaved complex */
    int sizeBuf, nBuf, in_size, in_use, Windex, Rindex;
} wo_shim;
typedef void (*wo_fexchange0_fn)(void *ctx, double *in, double *out, int *error);
wo_shim *wo_shim_create(void);
void wo_shim_destroy(wo_shim *s);
void wo_shim_set_parameter(wo_shim *s, int in_size, int in_use);
int wo_shim_fexchange0(wo_shim *s, wo_fexchange0_fn fn, void *ctx, double *cSamples, int nSamples);

#ifdef __cplusplus
}
#endif
#endif
