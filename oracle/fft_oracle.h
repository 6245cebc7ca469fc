/* fft_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * Plain radix-2 complex FFT in double precision used by the CPU restatement of the
 * reference's FFT-based blocks.  The reference calls FFTW3 (a system library that is
 * not vendored in /root/reference and is absent from this image): unnormalised DFT,
 * sign -1 forward / +1 backward (wdsp/firmin.c:313-318, quisk.c:5999).  Any correct
 * DFT is a valid stand-in for those semantics; this one is checked against numpy.fft
 * in tests/test_oracle_fft.py.
 */
#ifndef FFT_ORACLE_H
#define FFT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* In-place complex DFT of n points (radix-2 for powers of two, the O(n^2) definition otherwise), interleaved re/im doubles.
 * sign = -1: forward (FFTW_FORWARD), sign = +1: backward (FFTW_BACKWARD), unnormalised. */
void fo_fft(double *x, int n, int sign);

/* Out-of-place convenience. */
void fo_fft_oop(const double *in, double *out, int n, int sign);

#ifdef __cplusplus
}
#endif
#endif
