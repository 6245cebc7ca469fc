/* quiskhip.h -- C ABI of libquiskhip.so: MI355X (gfx950) receive DSP for Quisk / WDSP.
 *
 * Plain C, plain pointers and sizes; no C++ or torch types.  Three groups of entry points:
 *
 *  1. qh_rxa_*    batched many-channel RXA engine (device-resident buffers).  The reference has
 *                 no batched form: it runs one DSP thread per channel (wdsp/channel.c:31-35) and
 *                 caps channels at 32 (wdsp/comm.h:117).  This is what bench.py measures.
 *  2. WDSP names  OpenChannel / fexchange0 / SetRXA* ... with the reference's exact signatures,
 *                 so that quisk_wdsp.py's ctypes.CDLL("./wdsp/libwdsp.so") (quisk_wdsp.py:27-41)
 *                 and quisk_wdsp.c's function pointer (quisk_wdsp.c:22,57) bind unchanged.
 *  3. qh_fir_*    batched FIR / decimator primitives equivalent to filter.c (filter.h:39-55).
 *
 * All functions returning int return 0 on success and a negative qh_status on failure;
 * qh_last_error() gives the message.  Nothing here falls back to the CPU: without a usable
 * HIP device every compute entry point fails with QH_ERR_NO_DEVICE.
 */
#ifndef QUISKHIP_H
#define QUISKHIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    QH_OK = 0,
    QH_ERR_NO_DEVICE = -1,
    QH_ERR_INVALID = -2,
    QH_ERR_UNSUPPORTED = -3,
    QH_ERR_HIP = -4
} qh_status;

/* ------------------------------------------------------------------ library */
int qh_version(void);                       /* 100 * major + minor */
const char *qh_last_error(void);            /* thread-local message of the last failure */
int qh_device_count(void);                  /* number of visible HIP devices (0 without a GPU) */

/* RXA demodulator modes, numerically identical to wdsp/RXA.h:31-45 */
enum { QH_LSB = 0, QH_USB, QH_DSB, QH_CWL, QH_CWU, QH_FM, QH_AM, QH_DIGU, QH_SPEC, QH_DIGL, QH_SAM, QH_DRM };

/* ------------------------------------------------------------------ 1. batched RXA engine */
typedef struct qh_rxa qh_rxa;

/* Creates `nch` independent receiver channels, each configured exactly like
 * OpenChannel(ch, *, dsp_size, in_rate, dsp_rate, out_rate, type 0, ...) followed by create_rxa()
 * (wdsp/channel.c:75-103, wdsp/RXA.c:31-490): shift run=1 at 0 Hz, nbp0 -4150..-150 Hz nc 2048,
 * AGC mode 3, panel gain 4.  `stream` is a hipStream_t (NULL = the engine creates its own).
 * in_rate / dsp_rate = 1, 2, 4, 8 or 16 runs the fused overlap-save front stage; any other whole ratio up or down (3, 5,
 * 6 ...; 1/2, 1/4 ...: what pre_main_build's integer divisions size consistently, wdsp/channel.c:39-42) runs xshift and the
 * polyphase resampler of wdsp/resample.c:35-157 as two passes.  out_rate: an integer multiple or fraction of dsp_rate. */
qh_rxa *qh_rxa_create(int device, int nch, int dsp_size, int in_rate, int dsp_rate, int out_rate, void *stream);
void qh_rxa_destroy(qh_rxa *e);

int qh_rxa_nch(const qh_rxa *e);
int qh_rxa_dsp_insize(const qh_rxa *e);     /* complex input samples per DSP block  (wdsp/channel.c:39-42) */
int qh_rxa_dsp_outsize(const qh_rxa *e);    /* complex output samples per DSP block (wdsp/channel.c:44-47) */
void *qh_rxa_stream(const qh_rxa *e);                            /* the hipStream_t the engine launches on */
/* Launch-sequence replay for block-at-a-time callers (the WDSP drop-in switches it on).  While the arguments of
 * qh_rxa_process and every parameter stand still, the launches of a call are captured into hipGraphs -- one per
 * state of the engine's ping-pong history buffers -- and replayed; a setter, a flush or different arguments drop
 * them.  Results are those of the plain path.  Off by default: a caller that passes large batches gains nothing. */
int qh_rxa_set_graph_replay(qh_rxa *e, int on);
long long qh_rxa_graph_launches(const qh_rxa *e);                /* calls served by a replayed graph so far */
/* Tile of the fircore (nbp0 / bp1 / bpsnba / FM audio) stages for nc <= 2048: 4096 points (0 = default), or 8192 points
 * shared by two lane groups: three times the useful outputs per pair of transforms and 22 % fewer instructions per sample, yet
 * slower on MI355X because its barriers hold eight wavefronts; kept selectable for that comparison.
 * nc = 4096 always runs 8192-point tiles, nc = 8192 ... 65536 two ... sixteen 4096-tap partitions on them.  Results agree to rounding; a change rebuilds the filter masks on the next call.
 * qh_rxa_band_tile: the size in use. */
int qh_rxa_set_band_tile(qh_rxa *e, int nfft);
int qh_rxa_band_tile(const qh_rxa *e);

/* Per-channel setters; `ch` indexes the batch, or -1 for every channel.  Same meaning as the WDSP
 * export of the same name (cited in section 2).  They take effect at the next qh_rxa_process call,
 * i.e. on a DSP-block boundary, which is when the reference applies them (csDSP, wdsp/main.c:41). */
int qh_rxa_SetRXAMode(qh_rxa *e, int ch, int mode);
int qh_rxa_RXASetPassband(qh_rxa *e, int ch, double f_low, double f_high);
int qh_rxa_RXASetNC(qh_rxa *e, int ch, int nc);     /* a power of two in [dsp_size, 65536]; above 4096: partitions of 4096 taps (tests/test_gpu_long_nc.py) */
int qh_rxa_SetRXAShiftRun(qh_rxa *e, int ch, int run);
int qh_rxa_SetRXAShiftFreq(qh_rxa *e, int ch, double fshift);
int qh_rxa_RXANBPSetRun(qh_rxa *e, int ch, int run);
int qh_rxa_RXANBPSetFreqs(qh_rxa *e, int ch, double flow, double fhigh);
int qh_rxa_SetRXABandpassRun(qh_rxa *e, int ch, int run);
int qh_rxa_SetRXABandpassFreqs(qh_rxa *e, int ch, double f_low, double f_high);
int qh_rxa_SetRXAAGCMode(qh_rxa *e, int ch, int mode);
int qh_rxa_SetRXAAGCFixed(qh_rxa *e, int ch, double fixed_agc_db);
int qh_rxa_SetRXAAGCAttack(qh_rxa *e, int ch, int attack_ms);
int qh_rxa_SetRXAAGCDecay(qh_rxa *e, int ch, int decay_ms);
int qh_rxa_SetRXAAGCHang(qh_rxa *e, int ch, int hang_ms);
int qh_rxa_SetRXAAGCTop(qh_rxa *e, int ch, double max_agc_db);
int qh_rxa_SetRXAAGCSlope(qh_rxa *e, int ch, int slope);
int qh_rxa_SetRXAAGCHangThreshold(qh_rxa *e, int ch, int hangthreshold);
int qh_rxa_SetRXAPanelGain1(qh_rxa *e, int ch, double gain);
int qh_rxa_SetRXAPanelGain2(qh_rxa *e, int ch, double gainI, double gainQ);
int qh_rxa_SetRXAPanelSelect(qh_rxa *e, int ch, int select);
int qh_rxa_SetRXAPanelCopy(qh_rxa *e, int ch, int copy);
int qh_rxa_SetRXAAMDSBMode(qh_rxa *e, int ch, int sbmode);
int qh_rxa_SetRXAAMDRun(qh_rxa *e, int ch, int run);             /* wdsp/amd.c:264-277; run = 1 on a channel in FM mode (both detectors in a row in the reference) is refused at the next process call */
int qh_rxa_SetRXAFMLimRun(qh_rxa *e, int ch, int run);           /* wdsp/fmd.c:336-347: the FM detector's limiter */
int qh_rxa_SetRXAFMLimGain(qh_rxa *e, int ch, double gaindB);    /* wdsp/fmd.c:349-362 */
/* xemnr, WDSP's spectral noise reduction "NR2" (wdsp/emnr.c; setters :1096-1143): overlap-add STFT 4096 / 1024, noise estimate by
 * minimum statistics (npe 0), speech presence (npe 1) or LambdaDl (npe 2), gain methods 0 Gaussian-amplitude, 1 log-MMSE, 2 gamma tables, 3 trained
 * zeta tables, post-filter aepf; position 0 (before bp1 and the AGC) or 1.  The tables are the data WDSP loads at create time from
 * its files `calculus` (GG, GGS: 241 x 241 doubles each) and `zetaHat.bin` (60 x 60 doubles, 60 x 60 int validity flags, ranges in
 * dB): hand them over once per engine before switching EMNR on.  dsp_size up to 1024. */
int qh_rxa_SetEMNRTables(qh_rxa *e, const double *GG, const double *GGS, const double *zeta_hat, const int *zeta_valid, double gamma_min,
                         double gamma_max, double xi_min, double xi_max);
int qh_rxa_SetRXAEMNRRun(qh_rxa *e, int ch, int run);
int qh_rxa_SetRXAEMNRgainMethod(qh_rxa *e, int ch, int method);
int qh_rxa_SetRXAEMNRnpeMethod(qh_rxa *e, int ch, int method);
int qh_rxa_SetRXAEMNRaeRun(qh_rxa *e, int ch, int run);
int qh_rxa_SetRXAEMNRPosition(qh_rxa *e, int ch, int position);
int qh_rxa_SetRXAEMNRaeZetaThresh(qh_rxa *e, int ch, double v);
int qh_rxa_SetRXAEMNRaePsi(qh_rxa *e, int ch, double v);
int qh_rxa_SetRXAEMNRtrainZetaThresh(qh_rxa *e, int ch, double v);
int qh_rxa_SetRXAEMNRtrainT2(qh_rxa *e, int ch, double v);
/* xamsqcap / xamsq, the AM squelch (wdsp/amsq.c:119-192; create_amsq arguments RXA.c:158-172) */
int qh_rxa_SetRXAAMSQRun(qh_rxa *e, int ch, int run);
/* SNBA, wdsp/snb.c:579-593 (run; also switches bpsnba and bp1 as RXA.c:800-917) and :660-694 (pass band of its output resampler;
 * RXASetPassband calls it).  dsp_rate 12000 / 24000 / 48000, dsp_size <= 1024. */
int qh_rxa_SetRXASNBARun(qh_rxa *e, int ch, int run);
int qh_rxa_SetRXASNBAOutputBandwidth(qh_rxa *e, int ch, double flow, double fhigh);
/* the blanker's tuning, wdsp/snb.c:604-658 (defaults: create_rxa's 64, 2, 8.0, 20.0, 10, 2, 2, 0.5; SetRXASNBAovrlp is not provided) */
int qh_rxa_SetRXASNBAasize(qh_rxa *e, int ch, int size);
int qh_rxa_SetRXASNBAnpasses(qh_rxa *e, int ch, int npasses);
int qh_rxa_SetRXASNBAk1(qh_rxa *e, int ch, double k1);
int qh_rxa_SetRXASNBAk2(qh_rxa *e, int ch, double k2);
int qh_rxa_SetRXASNBAbridge(qh_rxa *e, int ch, int bridge);
int qh_rxa_SetRXASNBApresamps(qh_rxa *e, int ch, int presamps);
int qh_rxa_SetRXASNBApostsamps(qh_rxa *e, int ch, int postsamps);
int qh_rxa_SetRXASNBApmultmin(qh_rxa *e, int ch, double pmultmin);
int qh_rxa_SetRXASNBAovrlp(qh_rxa *e, int ch, int ovrlp);          /* snb.c:595; ch = -1: the frame advance is the engine's */
int qh_rxa_SetRXAAMSQThreshold(qh_rxa *e, int ch, double threshold_db);
int qh_rxa_SetRXAAMSQMaxTail(qh_rxa *e, int ch, double tail_seconds);
/* xanf / xanr (wdsp/anf.c:82-133, anr.c:82-133), setters wdsp/anf.c:175-239 and anr.c:175-238; which position (0 before
 * bp1 and the AGC, 1 after the AGC) also moves bp1, as in the reference.  Taps and delay 1..64. */
int qh_rxa_SetRXAANFRun(qh_rxa *e, int ch, int v);
int qh_rxa_SetRXAANFTaps(qh_rxa *e, int ch, int v);
int qh_rxa_SetRXAANFDelay(qh_rxa *e, int ch, int v);
int qh_rxa_SetRXAANFPosition(qh_rxa *e, int ch, int v);
int qh_rxa_SetRXAANFGain(qh_rxa *e, int ch, double v);
int qh_rxa_SetRXAANFLeakage(qh_rxa *e, int ch, double v);
int qh_rxa_SetRXAANFVals(qh_rxa *e, int ch, int taps, int delay, double gain, double leakage);
int qh_rxa_SetRXAANRRun(qh_rxa *e, int ch, int v);
int qh_rxa_SetRXAANRTaps(qh_rxa *e, int ch, int v);
int qh_rxa_SetRXAANRDelay(qh_rxa *e, int ch, int v);
int qh_rxa_SetRXAANRPosition(qh_rxa *e, int ch, int v);
int qh_rxa_SetRXAANRGain(qh_rxa *e, int ch, double v);
int qh_rxa_SetRXAANRLeakage(qh_rxa *e, int ch, double v);
int qh_rxa_SetRXAANRVals(qh_rxa *e, int ch, int taps, int delay, double gain, double leakage);
/* The notch database and nbp0's notched band-pass (wdsp/nbp.c:358-525, make_nbp :97-179, fir_mbandpass :64-80).
 * *rval receives the reference's return value (0, or -1 for an index out of range). */
int qh_rxa_RXANBPAddNotch(qh_rxa *e, int ch, int notch, double fcenter, double fwidth, int active, int *rval);
int qh_rxa_RXANBPDeleteNotch(qh_rxa *e, int ch, int notch, int *rval);
int qh_rxa_RXANBPEditNotch(qh_rxa *e, int ch, int notch, double fcenter, double fwidth, int active, int *rval);
int qh_rxa_RXANBPGetNotch(qh_rxa *e, int ch, int notch, double *fcenter, double *fwidth, int *active, int *rval);
int qh_rxa_RXANBPGetNumNotches(qh_rxa *e, int ch, int *nnotches);
int qh_rxa_RXANBPGetMinNotchWidth(qh_rxa *e, int ch, double *minwidth);
int qh_rxa_RXANBPSetTuneFrequency(qh_rxa *e, int ch, double tunefreq);
int qh_rxa_RXANBPSetShiftFrequency(qh_rxa *e, int ch, double shift);
int qh_rxa_RXANBPSetNotchesRun(qh_rxa *e, int ch, int run);
int qh_rxa_RXANBPSetWindow(qh_rxa *e, int ch, int wintype);
int qh_rxa_RXANBPSetAutoIncrease(qh_rxa *e, int ch, int autoincr);
int qh_rxa_RXASetMP(qh_rxa *e, int ch, int mp);                  /* wdsp/RXA.c:948-958: minimum-phase filters (mp_imp, fir.c:319); the FM channels of one engine share their filters' design: different flags among them are refused, like different nc */
int qh_rxa_SetRXAAMDFadeLevel(qh_rxa *e, int ch, int levelfade);
int qh_rxa_SetRXAFMDeviation(qh_rxa *e, int ch, double deviation);
int qh_rxa_SetRXACTCSSFreq(qh_rxa *e, int ch, double freq);
int qh_rxa_SetRXACTCSSRun(qh_rxa *e, int ch, int run);

/* Runs xrxa() (wdsp/RXA.c:561-598) over `nblk` consecutive DSP blocks of every channel.
 *   d_in  : device pointer, [nch][in_stride] interleaved complex double, nblk*dsp_insize samples used
 *   d_out : device pointer, [nch][out_stride] interleaved complex double, nblk*dsp_outsize samples written
 * Strides are in complex samples.  Filter/NCO state is carried to the next call exactly as the
 * reference carries it from block to block.  Asynchronous on the engine's stream.
 * The output rows may lie over the input rows (a caller that works in place): the engine then writes the output in a last pass
 * behind every read of the input instead of from its last filter's store -- the same samples within rounding (1e-11), decided by
 * the extent of the rows themselves (first sample of the first row to last sample of the last), not by where the matrices lie. */
int qh_rxa_process(qh_rxa *e, const double *d_in, long long in_stride, double *d_out, long long out_stride, int nblk);

/* Same with host buffers (synchronous; pageable memory; includes the PCIe copies). */
int qh_rxa_process_host(qh_rxa *e, const double *h_in, long long in_stride, double *h_out, long long out_stride, int nblk);

/* Tiles of the time-tiled FM loop that had to be re-run sequentially so far (diagnostics; 0 on carriers, a fraction of a
 * percent of the 256-sample tiles on noise alone). */
long long qh_rxa_pll_repairs(qh_rxa *h);
int qh_rxa_debug_pll(qh_rxa *h, int check_only, int ch, double *out, int max);   /* diagnostics, see qh_engine.hip */
int qh_rxa_debug_agc(qh_rxa *h, int form);                                         /* diagnostics: 0 time tiles for long calls (default), 1 sample by sample, 2 batches of 64 */
int qh_rxa_debug_agc_ends(qh_rxa *h, int slot, double *out, int max);                /* diagnostics, see qh_engine.hip */
long long qh_rxa_agc_repairs(qh_rxa *h);                                            /* wcpAGC time tiles the verify pass re-ran in order */
long long qh_rxa_agc_segments_rerun(qh_rxa *h);                                     /* super-segments of its boundary pass walked again */
int qh_rxa_agc_tiled_channels(qh_rxa *h);                                           /* channels whose xwcpagc took the time tiles in the last call */
int qh_rxa_synchronize(qh_rxa *e);

/* Meters (wdsp/meter.c:75-142).  They cost an extra pass, so they are off until enabled.  mt as wdsp/RXA.h:47-57:
 * 0 S_PK, 1 S_AV, 2 ADC_PK, 3 ADC_AV, 4 AGC_GAIN, 5 AGC_PK, 6 AGC_AV; values in dB, -400 before the first block. */
int qh_rxa_enable_meters(qh_rxa *e, int enable);
int qh_rxa_GetRXAMeter(qh_rxa *e, int ch, int mt, double *value);

/* flush_rxa (wdsp/RXA.c:527-559): zero the NCO phase and every filter history of every channel. */
int qh_rxa_flush(qh_rxa *e);

/* Timing of the kernels of qh_rxa_process with HIP events on the engine's stream.
 * enable != 0 brackets every kernel of later process calls with events; qh_rxa_timing() waits for
 * the last call and returns, for kernel k (0 = shift+resample stage, 1 = bandpass stage, 2 = state
 * bookkeeping), the milliseconds of the last call in ms[k].  Returns the number of entries. */
int qh_rxa_enable_timing(qh_rxa *e, int enable);
int qh_rxa_timing(qh_rxa *e, double *ms, int n);

/* Bytes of device memory the engine holds (state, masks, intermediate buffers). */
long long qh_rxa_device_bytes(const qh_rxa *e);

/* ------------------------------------------------------------------ 2. WDSP drop-in exports */
/* Signatures are the reference's: wdsp/channel.h:62, wdsp/iobuffs.h:89-90, and the PORT functions
 * cited per line.  Channel numbers 0..31 (wdsp/comm.h:117).  Each open channel is a 1-channel
 * qh_rxa engine plus the host-side ring logic of wdsp/iobuffs.c (two-block latency, upslew). */
int GetWDSPVersion(void);                                                        /* wdsp/version.c */
void OpenChannel(int channel, int in_size, int dsp_size, int input_samplerate, int dsp_rate,
                 int output_samplerate, int type, int state, double tdelayup, double tslewup,
                 double tdelaydown, double tslewdown, int bfo);                  /* wdsp/channel.c:75-103 */
void CloseChannel(int channel);                                                  /* wdsp/channel.c:122-128 */
int SetChannelState(int channel, int state, int dmode);                          /* wdsp/channel.c:260-298 */
void fexchange0(int channel, double *in, double *out, int *error);               /* wdsp/iobuffs.c:464-516 */
void fexchange2(int channel, float *Iin, float *Qin, float *Iout, float *Qout, int *error);     /* wdsp/iobuffs.c:518-582 */
void SetRXAMode(int channel, int mode);                                          /* wdsp/RXA.c:748-787 */
void RXASetPassband(int channel, double f_low, double f_high);                   /* wdsp/RXA.c:926-932 */
void RXASetNC(int channel, int nc);                                              /* wdsp/RXA.c:934-946 */
void RXASetMP(int channel, int mp);                                              /* wdsp/RXA.c:948-958 */
void SetRXAShiftRun(int channel, int run);                                       /* wdsp/shift.c:110-117 */
void SetRXAShiftFreq(int channel, double fshift);                                /* wdsp/shift.c:119-127 */
void RXANBPSetRun(int channel, int run);                                         /* wdsp/nbp.c:528-536 */
void RXANBPSetFreqs(int channel, double flow, double fhigh);                     /* wdsp/nbp.c:538-552 */
void SetRXABandpassRun(int channel, int run);                                    /* wdsp/bandpass.c:381-387 */
void SetRXABandpassFreqs(int channel, double f_low, double f_high);              /* wdsp/bandpass.c:389-407 */
void SetRXAAGCMode(int channel, int mode);                                       /* wdsp/wcpAGC.c:369-411 */
void SetRXAAGCFixed(int channel, double fixed_agc);                              /* wdsp/wcpAGC.c:541-548 */
void SetRXAAGCAttack(int channel, int attack);                                   /* wdsp/wcpAGC.c:413-420 */
void SetRXAAGCDecay(int channel, int decay);                                     /* wdsp/wcpAGC.c:422-429 */
void SetRXAAGCHang(int channel, int hang);                                       /* wdsp/wcpAGC.c:431-438 */
void SetRXAAGCTop(int channel, double max_agc);                                  /* wdsp/wcpAGC.c:520-527 */
void SetRXAAGCSlope(int channel, int slope);                                     /* wdsp/wcpAGC.c:529-536 */
void SetRXAAGCHangThreshold(int channel, int hangthreshold);                     /* wdsp/wcpAGC.c:480-487 */
void SetRXAPanelRun(int channel, int run);                                       /* wdsp/patchpanel.c:123-129 */
void SetRXAPanelGain1(int channel, double gain);                                 /* wdsp/patchpanel.c:139-145 */
void SetRXAPanelGain2(int channel, double gainI, double gainQ);                  /* wdsp/patchpanel.c:147-154 */
void SetRXAPanelSelect(int channel, int select);                                 /* wdsp/patchpanel.c:131-137 */
void SetRXAPanelCopy(int channel, int copy);                                     /* wdsp/patchpanel.c:175-181 */
void SetRXAAMDSBMode(int channel, int sbmode);                                   /* wdsp/amd.c:277-283 */
void SetRXAAMDRun(int channel, int run);                                         /* wdsp/amd.c:264-277 */
void SetRXAFMLimRun(int channel, int run);                                       /* wdsp/fmd.c:336-347 */
void SetRXAFMLimGain(int channel, double gaindB);                                /* wdsp/fmd.c:349-362 */
int RXANBPAddNotch(int channel, int notch, double fcenter, double fwidth, int active);          /* wdsp/nbp.c:358-388 */
int RXANBPGetNotch(int channel, int notch, double *fcenter, double *fwidth, int *active);       /* wdsp/nbp.c:390-412 */
int RXANBPDeleteNotch(int channel, int notch);                                   /* wdsp/nbp.c:414-438 */
int RXANBPEditNotch(int channel, int notch, double fcenter, double fwidth, int active);         /* wdsp/nbp.c:440-459 */
void RXANBPGetNumNotches(int channel, int *nnotches);                            /* wdsp/nbp.c:461-469 */
void RXANBPSetTuneFrequency(int channel, double tunefreq);                       /* wdsp/nbp.c:471-481 */
void RXANBPSetShiftFrequency(int channel, double shift);                         /* wdsp/nbp.c:483-493 */
void RXANBPSetNotchesRun(int channel, int run);                                  /* wdsp/nbp.c:495-514 */
void RXANBPSetWindow(int channel, int wintype);                                  /* wdsp/nbp.c:542-560 */
void RXANBPSetAutoIncrease(int channel, int autoincr);                           /* wdsp/nbp.c:600-619 */
void RXANBPGetMinNotchWidth(int channel, double *minwidth);                      /* wdsp/nbp.c:588-597 */
void SetRXAAMDFadeLevel(int channel, int levelfade);                             /* wdsp/amd.c:285-291 */
void SetRXAFMDeviation(int channel, double deviation);                           /* wdsp/fmd.c:236-246 */
void SetRXACTCSSFreq(int channel, double freq);                                  /* wdsp/fmd.c:248-258 */
void SetRXACTCSSRun(int channel, int run);                                       /* wdsp/fmd.c:260-267 */
double GetRXAMeter(int channel, int mt);                                         /* wdsp/meter.c:133-142 */
/* Quisk's re-blocking shim around fexchange0 (quisk_wdsp.c:24-69): any nSamples in, scaled by 1/CLIP32 into
 * in_size blocks, results scaled back; returns the number of samples written to cSamples.  qh_wdsp_set_parameter
 * is the C form of quisk_wdsp_set_parameter (quisk_wdsp.c:71-91; in_size <= 0 / in_use < 0 leave the value).  One deviation: a CHANGE
 * of in_size starts the ring again -- the reference keeps a ring whose length is no longer a multiple of the block and reads past
 * its end (quisk_wdsp.c:44-49,57-60). */
int wdspFexchange0(int channel, double *cSamples, int nSamples);
void qh_wdsp_set_parameter(int channel, int in_size, int in_use);
/* The same hand-off for samples that are already on the GPU (what qh_quisk_process_samples does at quisk.c:2660-2661):
 * d_samples = nSamples interleaved complex doubles in device memory with room for nSamples + in_size; the shim's ring, the
 * double rings of fexchange0 (wdsp/iobuffs.c:464-516), the up- and down-slews and the DSP blocks all stay in device memory
 * and are enqueued on the channel's stream (ordered behind `stream` on entry; `stream` is ordered behind it on return) --
 * no copy to the host, nothing waited for.  Returns the number of samples left in d_samples, like wdspFexchange0.  The same
 * samples as the host-pointer call, bit for bit; a channel may change between the two in mid-stream. */
int qh_wdsp_fexchange0_device(int channel, void *d_samples, int nSamples, void *stream);
/* accepted and ignored: these blocks are run = 0 on the hot path */
/* anf / anr: the leaky-LMS automatic notch filter and noise reduction of the RXA chain (wdsp/anf.c:175-239, anr.c:175-238);
 * up to 64 taps and a delay of up to 64 samples (defaults 64 / 16, RXA.c:285-286,305-306) */
void SetRXAANFRun(int channel, int v);
void SetRXAANFTaps(int channel, int v);
void SetRXAANFDelay(int channel, int v);
void SetRXAANFPosition(int channel, int v);
void SetRXAANFGain(int channel, double v);
void SetRXAANFLeakage(int channel, double v);
void SetRXAANFVals(int channel, int taps, int delay, double gain, double leakage);
void SetRXAANRRun(int channel, int v);
void SetRXAANRTaps(int channel, int v);
void SetRXAANRDelay(int channel, int v);
void SetRXAANRPosition(int channel, int v);
void SetRXAANRGain(int channel, double v);
void SetRXAANRLeakage(int channel, double v);
void SetRXAANRVals(int channel, int taps, int delay, double gain, double leakage);
void SetRXAAMSQRun(int channel, int run);                                        /* wdsp/amsq.c:216-222 */
void SetRXAAMSQThreshold(int channel, double threshold);                         /* wdsp/amsq.c:224-232, dB */
void SetRXAAMSQMaxTail(int channel, double tail);                                /* wdsp/amsq.c:234-243, seconds */
void SetRXAEMNRRun(int channel, int run);                                        /* wdsp/emnr.c:1096-1110; needs the files `calculus` and
                                                                                    `zetaHat.bin` in the working directory or in $QH_WDSP_DATA, as WDSP reads them (emnr.c:212,317) */
void SetRXAEMNRnpeMethod(int channel, int method);                               /* wdsp/emnr.c:1120 */
void SetRXAEMNRaeRun(int channel, int run);                                      /* wdsp/emnr.c:1128 */
void SetRXAEMNRPosition(int channel, int position);                              /* wdsp/emnr.c:1136 */
void SetRXAEMNRaeZetaThresh(int channel, double v);
void SetRXAEMNRaePsi(int channel, double v);
void SetRXAEMNRtrainZetaThresh(int channel, double v);
void SetRXAEMNRtrainT2(int channel, double v);

void SetRXAEMNRgainMethod(int channel, int method);                              /* wdsp/emnr.c:1112: gain methods 0..3 */
void SetRXASNBARun(int channel, int run);                                        /* wdsp/snb.c:579-593 */
void SetRXASNBAOutputBandwidth(int channel, double flow, double fhigh);          /* wdsp/snb.c:660-694 */
void SetRXASNBAasize(int channel, int size);                                     /* wdsp/snb.c:604-658 */
void SetRXASNBAnpasses(int channel, int npasses);
void SetRXASNBAk1(int channel, double k1);
void SetRXASNBAk2(int channel, double k2);
void SetRXASNBAbridge(int channel, int bridge);
void SetRXASNBApresamps(int channel, int presamps);
void SetRXASNBApostsamps(int channel, int postsamps);
void SetRXASNBApmultmin(int channel, double pmultmin);
void SetRXASNBAovrlp(int channel, int ovrlp);

/* Status of the drop-in layer: 0 when the last WDSP-named call succeeded, else a qh_status. */
int qh_wdsp_status(void);
long long qh_wdsp_graph_launches(void);     /* DSP blocks replayed from a captured hipGraph (steady state of fexchange0) */

/* ------------------------------------------------------------------ 3. batched FIR decimator bank */
/* GPU form of quisk_cDecimate / quisk_cCDecimate / quisk_cFilter (filter.c:203-257,372-375; filter.h:47-55)
 * over `nch` independent complex streams that share one filter:
 *     y[m] = sum_k h[k] * x[decim*m + (decim-1-phase) - k],   phase = the reference's decim_index, kept
 * between calls like struct quisk_cFilter keeps it (filter.h:1-10), as is the ntaps-1 sample history.
 * taps_im == NULL: real taps (quisk_cDecimate); else complex taps as quisk_filt_tune() builds them
 * (quisk_cCDecimate).  With qh_hb45_taps() and decim 2 it is quisk_cDecim2HB45 (filter.c:377-417). */
typedef struct qh_fir qh_fir;
enum { QH_F64 = 0, QH_F32 = 1 };            /* sample type: interleaved complex double / complex float */

qh_fir *qh_fir_create(int device, int nch, const double *taps_re, const double *taps_im, int ntaps, int decim,
                      int dtype, void *stream);
void qh_fir_destroy(qh_fir *f);
int qh_fir_reset(qh_fir *f);                                   /* history and phase back to zero */
/* Load state from the host: hist = the ntaps-1 most recent samples per channel, oldest first ([nch][ntaps-1],
 * NULL = zeros); phase = decim_index (0 .. decim-1). */
int qh_fir_set_state(qh_fir *f, const void *hist, int phase);
int qh_fir_out_count(const qh_fir *f, int n_in);               /* outputs the next call with n_in samples produces */
/* d_in [nch][in_stride], d_out [nch][out_stride] device pointers (strides in complex samples); *n_out = outputs
 * per channel (may be 0).  Asynchronous on the filter's stream.  In-place (d_out == d_in) is not supported. */
int qh_fir_process(qh_fir *f, const void *d_in, long long in_stride, int n_in, void *d_out, long long out_stride, int *n_out);
int qh_fir_process_host(qh_fir *f, const void *h_in, long long in_stride, int n_in, void *h_out, long long out_stride, int *n_out);
int qh_fir_synchronize(qh_fir *f);
/* The 43 taps (delays 0..42) of Quisk's 45-tap half-band whose outer taps are zero (filter.c:382-385). */
void qh_hb45_taps(double *taps43);

/* ------------------------------------------------------------------ 3b. fused half-band cascade */
/* `nstage` (1..8) consecutive quisk_cDecim2HB45 stages (filter.c:377-417; chained at quisk.c:1772-1796) over
 * `nch` complex streams in ONE pass over HBM: decimation 2^nstage, state (the input history) carried between
 * calls.  n_in must be a multiple of 2^nstage; n_in / 2^nstage outputs per channel.  dtype QH_F64 / QH_F32.
 * Results equal nstage calls of quisk_cDecim2HB45 to rounding (time-domain sums, not FFT).  In-place (d_out over d_in) is not
 * supported: the time segments of a call run side by side (the drop-in quisk_cDecim2HB45, which works in place like the reference's, goes
 * through rows of its own). */
typedef struct qh_hbc qh_hbc;
qh_hbc *qh_hbc_create(int device, int nch, int nstage, int dtype, void *stream);
void qh_hbc_destroy(qh_hbc *h);
int qh_hbc_reset(qh_hbc *h);
int qh_hbc_process(qh_hbc *h, const void *d_in, long long in_stride, int n_in, void *d_out, long long out_stride);
int qh_hbc_process_host(qh_hbc *h, const void *h_in, long long in_stride, int n_in, void *h_out, long long out_stride);
int qh_hbc_synchronize(qh_hbc *h);

/* ------------------------------------------------------------------ 3c. polyphase rational resampler */
/* quisk_cInterpDecim (filter.c:287-324) for `nch` complex streams, real taps; decim = 1 is quisk_cInterpolate
 * (filter.c:131-165); two real streams ride as the real and imaginary parts (quisk_dInterpolate, filter.c:167-201).
 * Output m sits at upsampled position p = phase + m*decim:  y[m] = interp * sum_k taps[p % interp + k*interp] *
 * x[p / interp - k].  `phase` (the reference's decim_index) and the last ceil(ntaps/interp)-1 inputs carry over. */
typedef struct qh_rat qh_rat;
qh_rat *qh_rat_create(int device, int nch, const double *taps, int ntaps, int interp, int decim, int dtype, void *stream);
void qh_rat_destroy(qh_rat *h);
int qh_rat_reset(qh_rat *h);
/* hist: [nch][ceil(ntaps/interp) - 1] complex samples of the stream's dtype, oldest first (NULL = zeros). */
int qh_rat_set_state(qh_rat *h, const void *hist, int phase);
int qh_rat_phase(const qh_rat *h);
int qh_rat_out_count(const qh_rat *h, int n_in);
int qh_rat_process(qh_rat *h, const void *d_in, long long in_stride, int n_in, void *d_out, long long out_stride, int *n_out);
int qh_rat_process_host(qh_rat *h, const void *h_in, long long in_stride, int n_in, void *h_out, long long out_stride, int *n_out);
int qh_rat_synchronize(qh_rat *h);

/* ------------------------------------------------------------------ 5. batched panadapter */
/* Quisk's spectrum display path for `nch` receivers: the FFT ring producer of quisk_process_samples
 * (quisk.c:2454-2475), record_app's Hanning window (quisk.c:6003-6009) and get_graph job 1
 * (quisk.c:5142-5331: window, complex FFT, RMS S-meter over the passband bins, fftshift, |X| average,
 * box sum per pixel, 20*log10 - 20*(log10 count + log10 N + 31 log10 2), clamp [-200, 0]).
 * Every completed block of fft_size samples is transformed (the reference drops blocks when the GUI is
 * slow).  fft_size: any even size 16 .. 16384 ("FFT size must be an even number", quisk.py:186).  Powers of two from 1024 run
 * the fused kernel; the sizes Quisk itself picks (data_width * fft_mult with data_width = 2^a * y * z, quisk.py:186-194,4179:
 * 4000, 9600 ...) run Bluestein's algorithm on the power-of-two transforms, about three times the work. */
typedef struct qh_pan qh_pan;
qh_pan *qh_pan_create(int device, int nch, int fft_size, int data_width, double sample_rate, void *stream);
void qh_pan_destroy(qh_pan *p);
/* S-meter passband of channel ch (-1 = all): starts at f_start = rx_tune_freq + filter_start_offset (Hz, signed),
 * width = filter_bandwidth (quisk.c:5223-5229). */
int qh_pan_set_smeter_band(qh_pan *p, int ch, double f_start, double bandwidth);
/* Append n raw (pre-tune) IQ samples per channel, device pointer [nch][in_stride] complex double. */
int qh_pan_feed(qh_pan *p, const double *d_in, long long in_stride, int n);
int qh_pan_feed_host(qh_pan *p, const double *h_in, long long in_stride, int n);
int qh_pan_count(const qh_pan *p);          /* FFTs averaged since the last qh_pan_graph */
/* A decimating FIR on the panadapter's read: quisk_process_samples feeds the same cSamples to the FFT ring (quisk.c:2454-2475)
 * and to quisk_cDecimate (filter.c:203-229); with fft_size 16384, decimation 32 and up to 1024 real taps (BASELINE config 3) the
 * panadapter's own transform serves both (panfir16k_kernel, qh_pan.hip) and the stream is read once.  qh_pan_feed_decimate
 * takes whole blocks (n a multiple of fft_size, the panadapter at a block boundary) and writes n / decim samples per channel
 * exactly as quisk_cDecimate(cSamples, n, filter, decim) leaves them, state carried between calls; other shapes are refused
 * (QH_ERR_UNSUPPORTED): qh_fir beside qh_pan_feed does those. */
int qh_pan_attach_fir(qh_pan *p, const double *taps, int ntaps, int decim);
int qh_pan_feed_decimate(qh_pan *p, const double *d_in, long long in_stride, int n, double *d_out, long long out_stride, int *n_out);
/* The refresh branch of get_graph: h_pixels [nch][data_width] dB, h_smeter [nch] dB (either may be NULL),
 * *count = FFTs that were averaged (0: nothing was written, like get_graph returning None). */
int qh_pan_graph(qh_pan *p, double zoom, double deltaf, double *h_pixels, double *h_smeter, int *count);
/* The waterfall row of the same refresh: get_graph followed by watfall_OnGraphData
 * (quisk.c:5372-5421) -- colour index (int)((dB - gain + 40 + 0.69 y_zero) * (y_scale + 10) * 0.10 + 128) clamped to
 * 0..255 through the 256-entry red / green / blue tables of watfall_RgbData (quisk.c:5334); h_rgb [nch][width][3]
 * bytes, pixels past data_width black.  Resets the average like qh_pan_graph. */
int qh_pan_waterfall(qh_pan *p, double zoom, double deltaf, const unsigned char *red, const unsigned char *green,
                     const unsigned char *blue, int y_zero, int y_scale, double gain, int width, unsigned char *h_rgb,
                     double *h_smeter, int *count);
/* watfall_OnGraphData alone for `nrows` dB rows on the host: h_db [nrows][ncols] -> h_rgb [nrows][width][3]. */
int qh_watfall_rows_host(int device, const double *h_db, int nrows, int ncols, const unsigned char *red,
                         const unsigned char *green, const unsigned char *blue, int y_zero, int y_scale, double gain,
                         int width, unsigned char *h_rgb);

/* The bandscope (get_bandscope, quisk.c:4957-5011; 8(f) rank 4) for `nch` ADC streams: blocks of bandscope_size REAL
 * samples (already divided by bandscopeScale, quisk.c:3596) -> Hanning (init_bandscope, quisk.c:2887) -> r2c ->
 * |X[0 .. size/2]| averaged over the blocks; qh_bscope_graph is the refresh branch: copy2pixels (quisk.c:4932-4955:
 * fractional-bin box sums for the view (zoom, deltaf) of 0 .. clock / 2), scale, dB with floor -200, and
 * hermes_adc_level = the largest |sample| since the last call.  bandscope_size 1024 .. 16384, a power of two. */
typedef struct qh_bscope qh_bscope;
qh_bscope *qh_bscope_create(int device, int nch, int bandscope_size, int graph_width, void *stream);
void qh_bscope_destroy(qh_bscope *b);
int qh_bscope_feed(qh_bscope *b, const double *d_in, long long in_stride, int n);         /* [nch][in_stride] doubles */
int qh_bscope_feed_host(qh_bscope *b, const double *h_in, long long in_stride, int n);
int qh_bscope_count(const qh_bscope *b);
int qh_bscope_graph(qh_bscope *b, int clock, double zoom, double deltaf, double *h_pixels, double *h_adc_level, int *count);

/* ------------------------------------------------------------------ 6. Quisk-native receiver bank */
/* The receive path of quisk_process_samples (quisk.c:2289-2742) for `nch` receivers that share the sample rate
 * and the mode: NCO tune (quisk.c:2477-2488) -> quisk_process_decimate (quisk.c:1673-1846, PlanDecimation rates:
 * 48000 * 2^a 3^b 5^c) -> quisk_process_demodulate (quisk.c:1848-2068: CWL 0, CWU 1, LSB 2, USB 3, AM 4, FM 5, the
 * rx_mode_type values of quisk.h:55-70) with the Rx filter of set_filters -> mono to both channels.  Output: 48 ksps
 * complex (d + I*d).  process_agc, squelch, auto-notch and the noise blanker are not applied.
 * The seven coefficient tables are those of filters.h (98, 147, 245, 50, 36, 186, 309 values). */
typedef struct qh_qrx qh_qrx;
qh_qrx *qh_qrx_create(int device, int nch, int sample_rate, int mode, const double *quiskFilt48dec24Coefs,
                      const double *quiskFilt144D3Coefs, const double *quiskFilt240D5CoefsSharp,
                      const double *quiskAudio24p4Coefs, const double *quiskAudio24p6Coefs,
                      const double *quiskLpFilt48Coefs, const double *quiskAudioFmHpCoefs, void *stream);
void qh_qrx_destroy(qh_qrx *r);
int qh_qrx_filter_rate(const qh_qrx *r);                                   /* get_filter_rate, quisk.c:2787-2859 */
int qh_qrx_set_tune(qh_qrx *r, int ch, int rx_tune_freq);                   /* set_tune, quisk.c:4702; ch -1 = all */
int qh_qrx_set_tune_all(qh_qrx *r, const int *rx_tune_freq);                /* rx_tune_freq[nch]: every receiver's set_tune in ONE launch per table (bit-identical to nch calls) */
int qh_qrx_set_filters(qh_qrx *r, int ch, const double *filtI, const double *filtQ, int size);  /* set_filters, quisk.c:4551 */
int qh_qrx_out_count(const qh_qrx *r, int n_in);                            /* 48 ksps samples the next call returns */
/* Every rate and mode of the path: the filters.h tables by name (the last six may be NULL when the sample rate
 * does not need them: quiskFilt300D5Coefs for rates that land on 50-60 ksps and take the 6/5 x 4/5 stage,
 * quisk.c:1834-1838; the SDR-IQ tables for 53/111/133/185/370/740/1333 ksps, quisk.c:1732-1768).  `bandwidth` is
 * set_filters' third argument (quisk.c:4581): DGT-U/L and FDV-U/L filter at decim_rate/8 below 3000 Hz and at
 * decim_rate otherwise (quisk.c:2089), DGT-IQ is unfiltered from 19000 Hz up (quisk.c:2143).  Modes: rx_mode_type
 * (quisk.h:55-70) 0..13 except EXT (6, a user plugin); IMD takes the SSB path as in the reference; FDV-U/L stop at
 * the audio that the reference hands to the codec.  Output rate = qh_qrx_decim_rate() (48000 except SDR-IQ rates);
 * DGT-IQ output is the filtered IQ stream, every other mode (d, d). */
typedef struct qh_qrx_tables {
    const double *f48dec24, *f144d3, *f240d5, *audio24p4, *audio24p6, *lp48, *fmhp;         /* 98 147 245 50 36 186 309 */
    const double *f300d5, *sdriq53, *sdriq111, *sdriq133, *sdriq167, *sdriq185;             /* 125 55 114 136 174 189 */
} qh_qrx_tables;
qh_qrx *qh_qrx_create_ex(int device, int nch, int sample_rate, int mode, int bandwidth, const qh_qrx_tables *tables, void *stream);
int qh_qrx_decim_rate(const qh_qrx *r);
/* d_in [nch][in_stride] complex double at sample_rate, d_out [nch][out_stride] at 48 ksps; any n_in. */
int qh_qrx_process(qh_qrx *r, const double *d_in, long long in_stride, int n_in, double *d_out, long long out_stride, int *n_out);
int qh_qrx_process_host(qh_qrx *r, const double *h_in, long long in_stride, int n_in, double *h_out, long long out_stride, int *n_out);
int qh_qrx_synchronize(qh_qrx *r);

/* ------------------------------------------------------------------ 7. wire-format sample ingest */
/* Quisk's sample sources left-justify 1-4 byte integer IQ in an int32 (full scale +-2^31) and, for UDP sources,
 * scale by rx_udp_gain_correct: quisk_read_rx_udp (quisk.c:3378-3392), add_rx_samples (quisk.c:2923-2952), the
 * Hermes / HPSDR frames of read_rx_udp10 (quisk.c:3745-3760).  This describes such a byte stream so that the GPU
 * reads it as it arrived: sample g of channel c starts at
 *   c*chan_stride + first_offset + (g / records_per_frame)*frame_stride + (g % records_per_frame)*record_stride
 * (records_per_frame 0 = one endless frame) and is two parts of sample_bytes bytes each.  Results are bit-exact
 * with the reference's conversion (integer -> double -> one multiply). */
typedef struct qh_iq_format {
    int sample_bytes;               /* 1..4 bytes per part */
    int big_endian;                 /* byte order of a part */
    int q_first;                    /* 1: the first part is the imaginary one (Hermes, quisk.c:3748-3750) */
    int records_per_frame;
    long long first_offset, record_stride, frame_stride;
    double gain;                    /* applied to the left-justified int32 (rx_udp_gain_correct; 1/2^31 for WDSP scale) */
} qh_iq_format;
void qh_iq_format_le24(qh_iq_format *f, double gain);               /* quisk_read_rx_udp, quisk.c:3378-3392 */
void qh_iq_format_hermes(qh_iq_format *f, int nrx, double gain);    /* read_rx_udp10 frames, quisk.c:3745-3760; channel r: chan_stride 6 */
/* d_src: packed bytes on the device; d_dst [nch][dst_stride] complex of `dtype`. */
int qh_unpack_iq(int device, void *stream, const void *d_src, long long src_bytes, const qh_iq_format *fmt, int nch,
                 long long chan_stride, int n, void *d_dst, long long dst_stride, int dtype);
/* qh_rxa_process with the decode fused into the first kernel's load: 6 bytes per sample cross HBM instead of 16
 * and no complex-double copy of the input exists.  nblk blocks of dsp_insize samples per channel. */
int qh_rxa_process_packed(qh_rxa *e, const void *d_src, long long src_bytes, const qh_iq_format *fmt, long long chan_stride,
                          double *d_out, long long out_stride, int nblk);
/* The same two from host memory (upload, run, download, synchronize). */
int qh_unpack_iq_host(int device, const void *h_src, long long src_bytes, const qh_iq_format *fmt, int nch, long long chan_stride,
                      int n, void *h_dst, long long dst_stride, int dtype);
int qh_rxa_process_packed_host(qh_rxa *e, const void *h_src, long long src_bytes, const qh_iq_format *fmt, long long chan_stride,
                               double *h_out, long long out_stride, int nblk);

/* read_rx_udp17 (quisk.c:3821-3999), the two-stream UDP source: packets of 2 header bytes (sequence number; status, bit 1 =
 * ADC overrange) + 6-byte records, 24-bit little-endian I then Q left-justified in an int32 and scaled by rx_udp_gain_correct.
 * The LSB of I sorts every sample into the receiver's stream (clear: d_ch0, what the function returns in cSamples0) or the
 * panadapter's (set: d_ch1, conjugated when the spectrum is inverted, the DC estimate (dc_re, dc_im) removed); on the
 * panadapter stream a clear LSB of Q marks the first sample of a scan's first block (d_marks: slots of d_ch1).  The flag bits
 * stay in the sample values, as in the reference.  d_counts[4] = samples on channel 0, on channel 1, marks, packets with
 * the overrange bit; d_dc_sum[2] = sum of the channel-1 samples ahead of the DC removal: the caller keeps the estimate (the
 * reference renews it from that sum once a second of wall time, quisk.c:3947-3952).  Buffers: npackets * (packet_bytes - 2) / 6
 * entries each.  Bit-exact with the reference's loop. */
int qh_unpack_udp17(int device, void *stream, const void *d_src, int npackets, int packet_bytes, double gain, int invert_spectrum,
                    double dc_re, double dc_im, void *d_ch0, void *d_ch1, int *d_marks, long long *d_counts, double *d_dc_sum);
int qh_unpack_udp17_host(int device, const void *h_src, int npackets, int packet_bytes, double gain, int invert_spectrum, double dc_re,
                         double dc_im, void *h_ch0, void *h_ch1, int *h_marks, long long *h_counts, double *h_dc_sum);

/* ------------------------------------------------------------------ 7b. audio egress */
/* The narrowing Quisk's sound back ends apply to the complex doubles a receive chain returns, done in the store of the
 * chain's last kernel so that 4 (Int16 stereo) instead of 16 bytes per output sample cross HBM:
 *   QH_AUDIO_I16   (short)(int)(volume * x / 65536)            sound_alsa.c:344-348, sound_pulseaudio.c:694-695
 *   QH_AUDIO_I24   three low bytes (LE) of (int)(volume * x / 256)                       sound_alsa.c:360-375
 *   QH_AUDIO_I32   (int)(volume * x)                           sound_alsa.c:386-390
 *   QH_AUDIO_F32   (float)(volume * x / CLIP32)                sound_pulseaudio.c:684-685, sound_portaudio.c:120-123
 * x = real / imaginary part at Quisk's +-2^31 scale.  prescale multiplies first (0 or 1: none): WDSP-scale chains (+-1.0)
 * pass 2147483647.0, the factor of quisk_wdsp.c:67.  A frame has num_channels slots; the real part goes to slot channel_I,
 * the imaginary part to slot channel_Q (struct sound_dev, quisk.h).  (int) truncates toward zero; it saturates where C
 * leaves the conversion undefined.  Bit-exact with the reference's expression for every in-range value. */
enum { QH_AUDIO_I16 = 1, QH_AUDIO_I24 = 2, QH_AUDIO_I32 = 3, QH_AUDIO_F32 = 4 };
typedef struct qh_audio_format {
    int kind, num_channels, channel_I, channel_Q;
    double volume, prescale;
} qh_audio_format;
/* qh_rxa_process with audio frames as output: d_out [nch][out_stride_bytes], nblk * dsp_outsize frames per channel. */
int qh_rxa_process_audio(qh_rxa *e, const double *d_in, long long in_stride, void *d_out, long long out_stride_bytes, int nblk,
                         const qh_audio_format *fmt);
/* Stand-alone: d_src [nch][src_stride] complex double on the device -> frames. */
int qh_audio_pack(int device, void *stream, const double *d_src, long long src_stride, int nch, int n, const qh_audio_format *fmt,
                  void *d_dst, long long dst_stride_bytes);

/* ------------------------------------------------------------------ 8. Quisk's audio AGC */
/* process_agc (quisk.c:2162-2287) for `nch` streams of complex double at `sample_rate` (the playback rate):
 * 15 ms look-ahead FIFO, gain ramp on overload, exponential release towards min(release gain, headroom).
 * max_out 0.7 and release_time 1.0 s are Quisk's values (quisk.c:2321,192); is_cpx selects |z| (EXT, DGT-IQ) or
 * |Re z| (every other mode, quisk.c:2686-2702).  As in the reference, the FIRST call only sets the state up and
 * leaves its samples untouched (quisk.c:2173-2190).  In place. */
typedef struct qh_qagc qh_qagc;
qh_qagc *qh_qagc_create(int device, int nch, int sample_rate, double max_out, double release_time, int is_cpx, void *stream);
void qh_qagc_destroy(qh_qagc *a);
int qh_qagc_set_gain(qh_qagc *a, int ch, double release_gain);          /* set_agc(d), quisk.c:4543; default 80 */
int qh_qagc_set_cpx(qh_qagc *a, int is_cpx);                            /* process_agc's is_cpx argument for the calls to come */
int qh_qagc_reset(qh_qagc *a);
int qh_qagc_process(qh_qagc *a, void *d_buf, long long stride, int n);
/* the same from one device buffer into another */
int qh_qagc_process2(qh_qagc *a, const void *d_src, long long src_stride, void *d_dst, long long dst_stride, int n);
/* diagnostics: 0 = the two regimes of the state machine as instruction chains (default), 1 = the whole machine sample by sample
   (bit-identical, ~9 times slower) */
int qh_qagc_debug_form(qh_qagc *a, int form);
int qh_qagc_process_host(qh_qagc *a, void *h_buf, long long stride, int n);
/* The receiver bank with process_agc on its output, as quisk_process_samples has it; off by default. */
int qh_qrx_set_agc(qh_qrx *r, int on, double release_gain);
/* FM / DGT-FM banks: set_squelch(d) (quisk.c:4721); the block is zeroed while the mean |cx| (dB re full scale, over >= 2400
 * samples, evaluated per call: quisk.c:2076-2085) is below `level`.  Default -999: never. */
int qh_qrx_set_squelch(qh_qrx *r, int ch, double level);
/* one quisk_process_samples call handed over in pieces: 1 ahead of the first piece, 0 behind the last (the FM squelch's level is looked at once
 * per call, quisk.c:2076-2085, whether a threshold is set or not) */
int qh_qrx_squelch_pieces(qh_qrx *r, int pieces);
/* CW / SSB / AM banks: set_ssb_squelch(enabled, level) (quisk.c:4729): spectral-flatness squelch on 512-sample blocks of the
 * audio at the filter rate (ssb_squelch, quisk.c:1086-1180) plus the 512-sample audio delay that goes with it (d_delay). */
int qh_qrx_set_ssb_squelch(qh_qrx *r, int enabled, int level);

/* ------------------------------------------------------------------ 10. Quisk's noise blanker */
/* NoiseBlanker (quisk.c:680-784) for `nch` fp64 complex streams at the receiver's INPUT rate,
 * where quisk_process_samples runs it (quisk.c:2448-2449: before the panadapter ring and the tune).  While on, the
 * output lags the input by qh_nb_delay() = 3 * (int)(sample_rate * 500e-6 + 0.5) samples (the reference's delay
 * line); level 0 passes samples through undelayed and freezes the delay line, as the reference does.  Levels 1..3 =
 * threshold 6.0 / 4.0 / 2.5 times the mean magnitude (set_noise_blanker, quisk.c:4605).  d_in and d_out must be
 * different buffers.  Sample rates up to about 3.5 MHz (the 500 us window has to fit one LDS tile). */
typedef struct qh_nb qh_nb;
qh_nb *qh_nb_create(int device, int nch, int sample_rate, void *stream);
void qh_nb_destroy(qh_nb *b);
int qh_nb_delay(const qh_nb *b);
int qh_nb_set_level(qh_nb *b, int level);
int qh_nb_reset(qh_nb *b);
int qh_nb_process(qh_nb *b, const void *d_in, long long in_stride, void *d_out, long long out_stride, int n);
int qh_nb_process_host(qh_nb *b, const void *h_in, long long in_stride, void *h_out, long long out_stride, int n);
int qh_nb_synchronize(qh_nb *b);
/* dAutoNotch (quisk.c:786-963; 8(f) rank 3) inside the receiver bank, where the mode calls it (on the real audio after
 * the Rx filter; after the interpolators for FM; DGT-IQ has none): set_auto_notch(i) (quisk.c:4596) -- stores the flag
 * and starts the notch over -- with rit_freq as set_sidetone passes it (quisk.c:4712): the CW modes keep the notch off
 * the sidetone.  Off by default. */
int qh_qrx_set_auto_notch(qh_qrx *r, int on, int rit_freq);
/* The receiver bank with the blanker in front of its tune, as quisk_process_samples has it; 0 = off (default). */
int qh_qrx_set_noise_blanker(qh_qrx *r, int level);

/* ------------------------------------------------------------------ 11. WDSP display engine (analyzer) */
/* wdsp/analyzer.c for a bank of `ndisp` displays that share one configuration: Spectrum0's sample
 * rings, the windowed transform every size - overlap samples, clip / flip / stitch (Celiminate, eliminate, stitch,
 * analyzer.c:179-279,555-600), the five detectors or the bin interpolation (detector, :282-461), the five averaging modes
 * (avenger, :463-553), the calibration spline (SetCalibration / build_interpolants / interpolate, :747-882,1380-1410) and
 * GetPixels (:1315).  The reference's dispatcher and worker threads become: every frame that is complete when a feed call
 * returns has been computed, in order.  fft sizes: powers of two from 512 to 8192 * 64; one LO per sub-span (dMAX_NUM_FFT = 1,
 * comm.h:125); SnapSpectrum is not built.  Samples are (I, Q) doubles; they are kept as floats like the reference's dINREAL
 * rings (comm.h:128-132).  Setters take the reference's arguments (analyzer.c:999-1017,1582-1676). */
typedef struct qh_ana qh_ana;
qh_ana *qh_ana_create(int device, int ndisp, int max_size, int max_stitch, void *stream);
void qh_ana_destroy(qh_ana *a);
int qh_ana_set_analyzer(qh_ana *a, int n_pixout, int n_fft, int typ, const int *flp, int sz, int bf_sz, int win_type, double pi, int ovrlp, int clp,
                        double fscLin, double fscHin, int n_pix, int n_stch, int calset, double fmin, double fmax, int max_w);
int qh_ana_set_calibration(qh_ana *a, int set_num, int n_points, const double *cal /* n_points rows of (frequency, value) */);
int qh_ana_set_detector_mode(qh_ana *a, int pixout, int mode);
int qh_ana_set_average_mode(qh_ana *a, int pixout, int mode);
int qh_ana_set_num_average(qh_ana *a, int pixout, int num);
int qh_ana_set_av_backmult(qh_ana *a, int pixout, double mult);
int qh_ana_set_sample_rate(qh_ana *a, int rate);
int qh_ana_set_norm_onehz(qh_ana *a, int pixout, int norm);
double qh_ana_get_enb(qh_ana *a);
int qh_ana_reset_pixel_buffers(qh_ana *a);
/* n = k * buff_size samples for sub-span ss of every display: d_iq[ndisp][disp_stride] on the device (qh_ana_feed) or in host
 * memory (qh_ana_feed_host; swap_iq = 1 reads Spectrum0's (Q, I) pair order).  *frames = pixel rows published by this call. */
int qh_ana_feed(qh_ana *a, int ss, const void *d_iq, long long disp_stride, int n, int *frames);
int qh_ana_feed_host(qh_ana *a, int ss, const double *h_iq, long long disp_stride, int n, int swap_iq, int *frames);
/* GetPixels for one display of the bank: *flag = 1 and num_pixels floats (dB) if a row has been published since the last read */
int qh_ana_get_pixels(qh_ana *a, int disp, int pixout, float *pix, int *flag);
/* every row of the last feed call, [ndisp][frames][num_pixels] floats: device pointer / host copy */
int qh_ana_rows(qh_ana *a, int pixout, const float **d_rows, int *frames, int *num_pixels);
int qh_ana_rows_host(qh_ana *a, int pixout, float *out, int max_frames, int *frames);
/* SnapSpectrum (analyzer.c:1337-1367): the complex transform of the next frame of (display, sub-span), 2 * size doubles, fft-shifted
 * (bins size/2 .. size-1, then 0 .. size/2-1, analyzer.c:710-711).  arm, feed, take; or arm and wait while another thread feeds. */
int qh_ana_snap_arm(qh_ana *a, int disp, int ss);
int qh_ana_snap_take(qh_ana *a, double *snap_buff, int *flag);
int qh_ana_snap_wait(qh_ana *a, double *snap_buff, int timeout_ms, int *flag);
void *qh_ana_stream(qh_ana *a);
long long qh_ana_frames(qh_ana *a);
int qh_ana_buff_size(qh_ana *a);
int qh_ana_num_pixels(qh_ana *a);
/* WDSP's own names and signatures (wdsp/analyzer.h:100-193, wdsp.h), one display per id 0..63, host pointers */
void XCreateAnalyzer(int disp, int *success, int m_size, int m_LO, int m_stitch, char *app_data_path);
void DestroyAnalyzer(int disp);
void SetAnalyzer(int disp, int n_pixout, int n_fft, int typ, int *flp, int sz, int bf_sz, int win_type, double pi, int ovrlp, int clp, double fscLin,
                 double fscHin, int n_pix, int n_stch, int calset, double fmin, double fmax, int max_w);
void SetCalibration(int disp, int set_num, int n_points, double (*cal)[2]);
void Spectrum0(int run, int disp, int ss, int LO, double *pbuff);
void Spectrum2(int run, int disp, int ss, int LO, float *pbuff);
void Spectrum(int disp, int ss, int LO, float *pI, float *pQ);
void OpenBuffer(int disp, int ss, int LO, void **Ipointer, void **Qpointer);
void CloseBuffer(int disp, int ss, int LO);
void GetPixels(int disp, int pixout, float *pix, int *flag);
void SnapSpectrum(int disp, int ss, int LO, double *snap_buff);
void SnapSpectrumTimeout(int disp, int ss, int LO, double *snap_buff, unsigned int timeout, int *flag);
void ResetPixelBuffers(int disp);
void SetDisplayDetectorMode(int disp, int pixout, int mode);
void SetDisplayAverageMode(int disp, int pixout, int mode);
void SetDisplayNumAverage(int disp, int pixout, int num);
void SetDisplayAvBackmult(int disp, int pixout, double mult);
void SetDisplaySampleRate(int disp, int rate);
void SetDisplayNormOneHz(int disp, int pixout, int norm);
double GetDisplayENB(int disp);

/* ------------------------------------------------------------------ 9. Quisk native block API, one receiver */
/* The shape of quisk.c's own receive API: a process-wide receiver, parameters through setters, samples through
 * `int quisk_process_samples(complex double *cSamples, int nSamples)` (quisk.h:375, quisk.c:2289) -- in place, returns
 * the output count at the playback rate, nSamples <= 0 returned unchanged.  process_agc is on as in the reference
 * (default release gain 80, quisk.c:191).  qh_quisk_open takes quisk_sound_state.sample_rate and .playback_rate (open_sound,
 * quisk.c:4106; 48000 x 1, 2, 4 or 8 -- the ratios quisk.c:2663-2682 interpolates), the filters.h tables and record_app's
 * fft_size / data_width (0, 0 = no panadapter).  The whole of quisk_process_samples (quisk.c:2289-2742) runs behind
 * qh_quisk_process_samples; the block is uploaded once and every step is a kernel on one stream (qh_quisk_rx_compat.cpp). */
int qh_quisk_open(int sample_rate, int playback_rate, const qh_qrx_tables *tables, int fft_size, int data_width);
void qh_quisk_close(void);
void qh_quisk_set_tune(int rx_tune_freq);                   /* set_tune, quisk.c:4702 */
void qh_quisk_set_rx_mode(int mode);                        /* set_rx_mode, quisk.c:4621 */
int qh_quisk_set_filters(const double *filtI, const double *filtQ, int size, int bandwidth);       /* set_filters, quisk.c:4551 */
void qh_quisk_set_agc(double level);                        /* set_agc, quisk.c:4543 */
void qh_quisk_set_noise_blanker(int level);                 /* set_noise_blanker, quisk.c:4605 */
void qh_quisk_set_auto_notch(int on, int rit_freq);         /* set_auto_notch, quisk.c:4596: the flag and the notch's restart.  rit_freq is IGNORED (kept for callers of earlier
                                                                 versions): the RIT is the one qh_quisk_set_sidetone set, the reference's global (quisk.c:4712) -- it also tunes the split
                                                                 receiver (quisk.c:2538) */
int qh_quisk_get_filter_rate(void);                         /* get_filter_rate(-1, 0), quisk.c:2787 */
int qh_quisk_process_samples(double *cSamples, int nSamples);       /* quisk_process_samples, quisk.c:2289.  In place, like the reference: cSamples is the caller's
                                                                         block buffer (SAMP_BUFFER_SIZE = 66000 samples in Quisk) and must hold the OUTPUT too -- up to (nSamples / decimation
                                                                         + one WDSP block when the shim is in use) x playback_rate / 48000 samples */
int qh_quisk_get_graph(double zoom, double deltaf, double *pixels, double *smeter);                /* get_graph, quisk.c:5142 */
/* get_filter (quisk.c:5481-5568): the Rx filter's response as the "RX Filter" screen draws it -- a multitone through the copy of the
 * cRxFilterOut loop that function carries, record_app's window, a data_width-point transform; db[data_width], negative frequencies
 * first, floor -140.  All sizeFilter taps of set_filters (up to MAX_FILTER_SIZE).  Returns data_width (0: error). */
int qh_quisk_get_filter(double *db);
/* The rest of quisk_process_samples' orchestration (quisk.c:2289-2742): a second receiver bank whose audio shares the
 * stereo output with the first -- split Rx/Tx (the same samples at tx_tune + rit, split modes 1..4 of quisk.c:2548-2590) or
 * the played sub-receiver (its own samples, frequency, mode and nFilter-1 filter; play methods 0..2 of quisk.c:2601-2620) --
 * with one AGC per output channel (Agc1 / Agc2, quisk.c:2690-2698); the key-down replacement of the block by sidetone or
 * silence, the TxRxSilenceMsec of silence after it and the 5 ms key-up ramp (quisk.c:2368-2433,2729-2738); kill_audio;
 * invert_spectrum. */
void qh_quisk_set_tx_tune(int tx_tune_freq);                               /* set_tune's second argument, quisk.c:4702 */
void qh_quisk_set_split_rxtx(int split);                                   /* set_split_rxtx, quisk.c:4694 */
void qh_quisk_set_multirx_play_channel(int ch);                            /* quisk.c:4856 */
void qh_quisk_set_multirx_play_method(int method);                         /* quisk.c:4846 */
void qh_quisk_set_multirx_freq(int index, int freq);                       /* quisk.c:4826 */
void qh_quisk_set_multirx_mode(int index, int mode);                       /* quisk.c:4836 */
int qh_quisk_multirx_samples(int index, const double *cSamples, int nSamples);     /* multirx_cSamples[index] for the next block */
int qh_quisk_set_filters2(const double *filtI, const double *filtQ, int size, int bandwidth);      /* set_filters(..., nFilter 1) */
void qh_quisk_set_key_state(int key_down, int cw_key_down, int active_sidetone, int is_fdx);
void qh_quisk_set_sidetone(double volume, int rit_freq, int playback_rate, int txrx_silence_msec);     /* set_sidetone, quisk.c:4710 */
void qh_quisk_set_kill_audio(int kill);
void qh_quisk_invert_spectrum(int invert);                                 /* quisk.c:4535 */
/* The tail and the side paths of quisk_process_samples: set_filters for any nFilter (0 main / split, 1 played sub-receiver,
 * 2 sub-receiver 1's digital output; one global sizeFilter as in quisk.c:4591); the squelches; add_tone (quisk.c:3203) and
 * AddTestTone (quisk.c:1258-1303); measure_frequency (quisk.c:3181) and measure_freq (quisk.c:5579-5649); sub-receiver 1
 * demodulated to a digital output device (quisk.c:2630-2651: quisk_multirx_count, the device's driver flag, the audio that
 * play_sound_interface is handed).  cFracDecim (quisk.c:622-665), the wdspFexchange0 hand-off (channel 0, when
 * qh_wdsp_set_parameter switched it on) and the HB45 interpolation to the playback rate need no setter. */
int qh_quisk_set_filters_n(const double *filtI, const double *filtQ, int size, int bandwidth, int nFilter);   /* set_filters, quisk.c:4551 */
void qh_quisk_set_squelch(double level);                                   /* set_squelch (FM), quisk.c:4721 */
void qh_quisk_set_ssb_squelch(int enabled, int level);                     /* set_ssb_squelch, quisk.c:4729 */
int qh_quisk_squelch_flags(void);                                          /* bit 0 squelch_real, bit 1 squelch_imag of the last block */
/* quisk_process_samples has no error return (failures are QuiskPrintf + counters, quisk.c:55): a call whose device chain failed returns
 * 0 samples, qh_last_error() says why, and this counter says that it happened (0 samples alone also means "no output yet") */
long long qh_quisk_error_count(void);
/* Mode EXT (quisk.c:2490-2493): the tuned block goes to the user's quisk_extern_demod (extdemod.c:13 -- `complex double *` in place,
 * returns the play-sample count <= nSamples) and from there straight to process_agc.  The reference links the function in; here it
 * is registered (a binding passes &quisk_extern_demod).  It is host code: the block makes one round trip.  Without one, mode EXT
 * fails loudly. */
void qh_quisk_set_extern_demod(int (*fn)(double *cSamples, int nSamples, double decim));
void qh_quisk_add_tone(int freq);                                          /* add_tone, quisk.c:3203 */
double qh_quisk_measure_frequency(int mode);                               /* measure_frequency, quisk.c:3181 */
void qh_quisk_set_multirx_count(int n);                                    /* quisk_multirx_count */
void qh_quisk_set_sub_rx1_output(int on);                                  /* quiskPlaybackDevices[QUISK_INDEX_SUB_RX1]->driver */
int qh_quisk_sub_rx1_audio(double *cSamples, int cap);                     /* the block play_sound_interface got, quisk.c:2651 */
/* ------------------------------------------------------------------ 9b. quisk_process_samples for a bank of receivers */
/* The whole of quisk_process_samples (quisk.c:2289-2742) for `nch` receivers side by side -- the batched form SURVEY.md 8(b) asks
 * for beside the drop-in symbols: AddTestTone / inversion (quisk.c:2438-2446), NoiseBlanker (:2448), the FFT ring feed on the same
 * samples (:2454-2475), tune + quisk_process_decimate + quisk_process_demodulate (:2477-2530), cFracDecim (:2654), the HB45
 * interpolation to the playback rate (:2663-2682), process_agc (:2686-2702, always on as in the reference) and kill_audio / the
 * squelches (:2712-2728).  Per receiver: tune frequency, Rx filter, FM squelch level; the rest are the bank's, as they are
 * process-wide globals in the reference.  The operator's side of the function (key-down replacement, key-up envelope, split and
 * sub-receiver channels, measure_freq, the WDSP hand-off) stays with the one-receiver API of group 9, which runs the same kernels
 * with nch = 1 (qh_ps_kernels.hpp).  fft_size / data_width 0, 0 = no panadapter.  stream NULL = a stream of the bank's own. */
typedef struct qh_qps qh_qps;
qh_qps *qh_qps_create(int device, int nch, int sample_rate, int playback_rate, int mode, int bandwidth, const qh_qrx_tables *tables,
                      int fft_size, int data_width, void *stream);
void qh_qps_destroy(qh_qps *h);
int qh_qps_set_tune(qh_qps *h, int ch, int rx_tune_freq);                  /* set_tune, quisk.c:4702; ch -1 = all */
int qh_qps_set_tune_all(qh_qps *h, const int *rx_tune_freq);               /* rx_tune_freq[nch], one launch per table for the whole bank */
int qh_qps_set_filters(qh_qps *h, int ch, const double *filtI, const double *filtQ, int size);     /* set_filters, quisk.c:4551 */
int qh_qps_set_agc(qh_qps *h, double level);                               /* set_agc, quisk.c:4543 (agcReleaseGain, default 80) */
int qh_qps_set_noise_blanker(qh_qps *h, int level);                        /* set_noise_blanker, quisk.c:4605 */
int qh_qps_set_auto_notch(qh_qps *h, int on, int rit_freq);                /* set_auto_notch, quisk.c:4596 */
int qh_qps_invert_spectrum(qh_qps *h, int invert);                         /* quisk.c:4535 */
int qh_qps_set_kill_audio(qh_qps *h, int kill);
int qh_qps_add_tone(qh_qps *h, int freq);                                  /* add_tone, quisk.c:3203; 0 = off */
int qh_qps_set_squelch(qh_qps *h, int ch, double level);                   /* set_squelch, quisk.c:4721: looked at by the FM demodulator alone -- accepted and without effect in a bank of another mode, like the reference's */
int qh_qps_set_ssb_squelch(qh_qps *h, int enabled, int level);             /* set_ssb_squelch, quisk.c:4729: CW, SSB and AM banks; accepted and without effect in an FM bank */
/* long calls run as `pieces` time pieces, process_agc of one beside the filters of the next (0 = chosen by call length) */
int qh_qps_set_pieces(qh_qps *h, int pieces);
int qh_qps_set_pipelined(qh_qps *h, int on);                               /* streaming callers: a call returns with its AGC still running and the next call's filters start beside it;
                                                                              output rows are complete after qh_qps_synchronize.  Same samples. */
int qh_qps_filter_rate(qh_qps *h);                                         /* get_filter_rate, quisk.c:2787 */
int qh_qps_decim_rate(qh_qps *h);
int qh_qps_out_capacity(qh_qps *h, int n_in);                              /* the most playback samples a call of n_in returns: out_stride >= this */
/* device rows [nch][stride] of complex doubles; *n_out = playback samples per receiver; asynchronous on the bank's stream */
int qh_qps_process(qh_qps *h, const double *d_in, long long in_stride, int n, double *d_out, long long out_stride, int *n_out);
int qh_qps_process_host(qh_qps *h, const double *h_in, long long in_stride, int n, double *h_out, long long out_stride, int *n_out);
int qh_qps_synchronize(qh_qps *h);
int qh_qps_squelch_flags(qh_qps *h, int *flags);                           /* squelch_real of every receiver after the last call */
int qh_qps_get_graph(qh_qps *h, double zoom, double deltaf, double *pixels, double *smeter, int *count);   /* get_graph, quisk.c:5142 */

/* Plumbing between the block API and the engines it chains (device-resident; used by qh_quisk_rx_compat.cpp): the bank
 * leaves the squelch to its caller and says where the flag lives; ssb_squelch's one-for-all-banks `plan` static
 * (quisk.c:1091,1104) and the one bandwidth it reads for all banks (filter_bandwidth[0], quisk.c:1120); the tuning oscillator's phase in 2^-64 turns (one vector per purpose in the reference,
 * quisk.c:2308-2311); measure_freq's window and averaged spectrum on the panadapter engine; the shim's state. */
int qh_qrx_set_mute_deferred(qh_qrx *r, int on);
const int *qh_qrx_squelch_flag(qh_qrx *r, int ch);
int qh_qrx_ssb_squelch_planned(qh_qrx *r, int set);
int qh_qrx_set_ssb_squelch_bandwidth(qh_qrx *r, int bandwidth);       /* filter_bandwidth[0]: what ssb_squelch reads in every bank (quisk.c:1120) */
int qh_qrx_get_nco_phase(qh_qrx *r, int ch, unsigned long long *phase);
int qh_qrx_set_nco_phase(qh_qrx *r, int ch, unsigned long long phase);
int qh_pan_set_window(qh_pan *p, const double *window);
int qh_pan_read_avg(qh_pan *p, double *h_avg, int reset);
int qh_pan_drop_partial(qh_pan *p);
int qh_wdsp_shim_in_size(int channel);

/* ------------------------------------------------------------------ 4. filter.h drop-in exports */
/* The reference's own names and struct layouts (filter.h:1-55) so that quisk.c links against this library
 * instead of filter.o.  `double *` stands for `complex double *` (same ABI: interleaved re, im).  Each call
 * takes the filter state from the caller's struct (circular history, ptcSamp, decim_index / toggle), runs the
 * block on the GPU and writes the state back in the reference's format, so calls may be mixed freely with the
 * reference's own functions on the same struct.  Errors (no device ...) return 0 samples and set
 * qh_last_error(); nothing is computed on the CPU. */
struct quisk_cFilter {                      /* filter.h:1-10 */
    double *dCoefs;
    double *cpxCoefs;                       /* complex double * */
    int nBuf;
    int nTaps;
    int decim_index;
    double *cSamples;                       /* complex double *, nTaps entries */
    double *ptcSamp;                        /* next write position inside cSamples */
    double *cBuf;
};
struct quisk_cHB45Filter {                  /* filter.h:23-29 */
    double *cBuf;
    int nBuf;
    int toggle;
    double samples[2 * 22];                 /* complex double samples[22] */
    double center[2 * 11];                  /* complex double center[11]  */
};
void quisk_filt_cInit(struct quisk_cFilter *filter, double *coefs, int taps);               /* filter.c:9-20 */
void quisk_filt_tune(struct quisk_cFilter *filter, double freq, int ssb_upper);             /* filter.c:58-81 */
int quisk_cDecimate(double *cSamples, int count, struct quisk_cFilter *filter, int decim);  /* filter.c:203-229 */
int quisk_cCDecimate(double *cSamples, int count, struct quisk_cFilter *filter, int decim); /* filter.c:231-257 */
int quisk_cFilter(double *cSamples, int count, struct quisk_cFilter *filter);               /* filter.c:372-375 */
int quisk_cDecim2HB45(double *cSamples, int count, struct quisk_cHB45Filter *filter);       /* filter.c:377-417 */
/* struct quisk_dFilter (filter.h:12-21) has the layout of struct quisk_cFilter; its history ring holds doubles. */
struct quisk_dHB45Filter {                  /* filter.h:31-37 */
    double *dBuf;
    int nBuf;
    int toggle;
    double samples[22];
    double center[11];
};
void quisk_filt_dInit(struct quisk_cFilter *filter, double *coefs, int taps);               /* filter.c:22-33 */
void quisk_filt_differInit(struct quisk_cFilter *filter, int taps);                         /* filter.c:35-56 */
int quisk_cInterpolate(double *cSamples, int count, struct quisk_cFilter *filter, int interp);      /* filter.c:131-165 */
int quisk_dInterpolate(double *dSamples, int count, struct quisk_cFilter *filter, int interp);      /* filter.c:167-201 */
int quisk_cInterpDecim(double *cSamples, int count, struct quisk_cFilter *filter, int interp, int decim);  /* filter.c:287-324 */
int quisk_dDecimate(double *dSamples, int count, struct quisk_cFilter *filter, int decim);  /* filter.c:259-285 */
int quisk_dFilter(double *dSamples, int count, struct quisk_cFilter *filter);               /* filter.c:347-370 */
double quisk_dD_out(double sample, struct quisk_cFilter *filter);                           /* filter.c:326-345 */
int quisk_cInterp2HB45(double *cSamples, int count, struct quisk_cHB45Filter *filter);      /* filter.c:455-488 */
int quisk_dInterp2HB45(double *dSamples, int count, struct quisk_dHB45Filter *filter);      /* filter.c:420-453 */
/* quisk_dC_out (filter.c:83-104) returns `complex double` by value.  The library exports it under the reference's own name
 * (microphone.c:469 links against it unchanged); ISO C++ has no spelling for the type, so the declaration is for C translation
 * units, and qh_quisk_dC_out is the same computation with the result written through a pointer. */
#ifndef __cplusplus
double _Complex quisk_dC_out(double sample, struct quisk_cFilter *filter);
#endif
void qh_quisk_dC_out(double sample, struct quisk_cFilter *filter, double *out_re_im);

#ifdef __cplusplus
}
#endif
#endif
