/* analyzer_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 * Synchronous restatement of wdsp/analyzer.c (PARITY UNPINNED by reference execution: wdsp needs <fftw3.h>). */
#ifndef ANALYZER_ORACLE_H
#define ANALYZER_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* wdsp/comm.h:123-139 */
#define AO_MAX_STITCH 4
#define AO_MAX_PIXELS 16384
#define AO_MAX_AVERAGE 60
#define AO_SAMP_BUFF_MULT 2
#define AO_NUM_PIXEL_BUFFS 3
#define AO_MAX_N 100
#define AO_MAX_CAL_SETS 2
#define AO_MAX_PIXOUTS 4

typedef struct ao_disp ao_disp;

ao_disp *ao_create(int max_size, int max_stitch);
void ao_destroy(ao_disp *a);
void ao_set_analyzer(ao_disp *a, int n_pixout, int typ, int flip, int sz, int bf_sz, int win_type, double pi, int ovrlp, int clp,
                     double fscLin, double fscHin, int n_pix, int n_stch, int calset, double fmin, double fmax, int max_w);
void ao_set_calibration(ao_disp *a, int set, int n_points, double *cal);
void ao_spectrum0(ao_disp *a, int run, int ss, const double *pbuff);
void ao_spectrum(ao_disp *a, int ss, const float *pI, const float *pQ);
int ao_get_pixels(ao_disp *a, int pixout, float *pix);
void ao_reset_pixel_buffers(ao_disp *a);
void ao_set_detector_mode(ao_disp *a, int pixout, int mode);
void ao_set_average_mode(ao_disp *a, int pixout, int mode);
void ao_set_num_average(ao_disp *a, int pixout, int num);
void ao_set_av_backmult(ao_disp *a, int pixout, double mult);
void ao_set_sample_rate(ao_disp *a, int rate);
void ao_set_norm_onehz(ao_disp *a, int pixout, int norm);
double ao_get_enb(ao_disp *a);
long ao_frames(ao_disp *a);
const double *ao_window_ptr(ao_disp *a);
const double *ao_cd_ptr(ao_disp *a);

void ao_window(int type, int size, double PiAlpha, double *w, double *inv_coherent_gain, double *inherent_power_gain, double *inv_enb);
void ao_detector(int det_type, int m, int num_pixels, double pix_per_bin, double bin_per_pix, const double *bins, double *pixels,
                 double inv_enb, double fsclipL, double fsclipH, double det_offset);

/* SnapSpectrum (analyzer.c:1337-1346) without the wait: the next frame of sub-span ss goes to buf (2 * size doubles, fft-shifted) */
void ao_snap(ao_disp *a, int ss, double *buf);
int ao_snap_taken(const ao_disp *a);

#ifdef __cplusplus
}
#endif
#endif
