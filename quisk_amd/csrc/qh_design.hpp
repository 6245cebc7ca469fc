// qh_design.hpp -- host-side filter design and table generation for the HIP receive chain.
// These run at parameter-change time only (the reference does the same work in
// calc_resample / fir_bandpass / calc_fircore on the setter's thread).
#pragma once
#include <complex>
#include <utility>
#include <vector>

namespace qh {

using cd = std::complex<double>;

// WDSP windowed-sinc bandpass, wdsp/fir.c:187-254.
//   rtype 0: N real taps returned as complex with zero imaginary part
//   rtype 1: N complex taps  c*cos(w n) - j*c*sin(w n)
// wintype 0 = 4-term Blackman-Harris, 1 = 7-term (fir.c:220-236).
std::vector<cd> fir_bandpass(int N, double f_low, double f_high, double samplerate, int wintype, int rtype, double scale);

// WDSP resampler prototype, wdsp/resample.c:35-72: taps in natural (time) order, real.
// Returns L, M and the tap count; `gain` as in create_resample.
struct ResamplerDesign { int L, M, ncoef, cpp; std::vector<double> h; };
ResamplerDesign design_resampler(int in_rate, int out_rate, double fc, int ncoef, double gain);

// WDSP FM de-emphasis curve by frequency sampling, wdsp/fcurve.c:29-145 + fir.c:129-185 (even nc).
std::vector<cd> fc_impulse(int nc, double f0, double f1, double g0, double g1, int curve, double samplerate,
                           double scale, int ctfmode, int wintype);

// Unnormalised forward DFT (power of two), double data with long double twiddles.
void host_fft(std::vector<cd> &x, int sign);

// Frequency-domain mask for qh::osfir_kernel: FFT_NFFT(h zero padded) / NFFT.
// WDSP notch database entry and the notched band-pass design, wdsp/nbp.c:64-179
struct Notch { double fcenter, fwidth; int active; };
std::vector<std::pair<double, double>> make_nbp(const std::vector<Notch> &notches, double minwidth, int autoincr, double flow,
                                                double fhigh, bool *havnotch);
std::vector<cd> fir_mbandpass(int N, const std::vector<std::pair<double, double>> &bands, double rate, double scale, int wintype);
std::vector<cd> mp_imp(const std::vector<cd> &fir, int pfactor, int polarity);     // wdsp/fir.c:319-368
std::vector<cd> make_mask(const std::vector<cd> &h, int nfft);

// Concatenated per-pass twiddle tables for qh::FftRR<N> (see the Plan table in qh_fft.hpp).
std::vector<cd> fft_twiddle_table(int n);

}  // namespace qh
