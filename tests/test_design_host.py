"""The host-side filter design of the library (quisk_amd/csrc/qh_design.cpp: plain C++, compiled here with g++) against the oracle's
restatement of wdsp/fir.c's fir_bandpass and wdsp/resample.c's calc_resample: the taps every band-pass, resampler and de-emphasis
stage of the GPU chain is made from.  CPU only."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = r'''
#include <cstring>
#include "qh_design.hpp"
extern "C" void t_fir_bandpass(int N, double lo, double hi, double fs, int wintype, int rtype, double scale, double *out)
{ auto h = qh::fir_bandpass(N, lo, hi, fs, wintype, rtype, scale); std::memcpy(out, h.data(), h.size() * 16); }
extern "C" int t_resampler(int in_rate, int out_rate, double fc, int ncoef, double gain, double *out, int cap, int *LM)
{ auto d = qh::design_resampler(in_rate, out_rate, fc, ncoef, gain); LM[0] = d.L; LM[1] = d.M; LM[2] = d.ncoef; LM[3] = d.cpp;
  if (d.ncoef <= cap) std::memcpy(out, d.h.data(), (size_t)d.ncoef * 8); return d.ncoef; }
extern "C" void t_fc_impulse(int nc, double f0, double f1, double g0, int curve, double fs, double scale, int ctfmode, int wintype, double *out)
{ auto h = qh::fc_impulse(nc, f0, f1, g0, 0.0, curve, fs, scale, ctfmode, wintype); std::memcpy(out, h.data(), h.size() * 16); }
'''


@pytest.fixture(scope="module")
def design(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("design")
    (d / "shim.cpp").write_text(SHIM)
    so = d / "libdesign.so"
    csrc = os.path.join(ROOT, "quisk_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", csrc, str(d / "shim.cpp"), os.path.join(csrc, "qh_design.cpp"), "-o", str(so)], check=True)
    return C.CDLL(str(so))


@pytest.mark.parametrize("N,lo,hi,fs", [(2048, 300.0, 3000.0, 48000.0), (2048, -3000.0, -300.0, 48000.0), (256, -4000.0, 4000.0, 48000.0),
                                        (561, -0.1171875, 0.1171875, 1.0), (141, 200.0 / 48000, 5400.0 / 48000, 1.0), (4096, 240.0, 3300.0, 48000.0)])
@pytest.mark.parametrize("wintype", [0, 1])
@pytest.mark.parametrize("rtype", [0, 1])
def test_fir_bandpass_is_the_reference_design(design, oracle, N, lo, hi, fs, wintype, rtype):
    out = np.zeros(N, dtype=np.complex128)
    design.t_fir_bandpass(C.c_int(N), C.c_double(lo), C.c_double(hi), C.c_double(fs), C.c_int(wintype), C.c_int(rtype), C.c_double(1.0 / 512),
                          out.ctypes.data_as(C.c_void_p))
    want = oracle.fir_bandpass(N, lo, hi, fs, wintype, rtype, 1.0 / 512)
    if rtype == 0:
        assert np.all(out.imag == 0.0)
        out = out.real
    scale = np.abs(want).max()
    assert scale > 0 and np.abs(out - want).max() <= 4e-16 * scale         # the same expressions; the last bit of a product at most


@pytest.mark.parametrize("rates", [(192000, 48000), (48000, 192000), (48000, 12000), (12000, 48000), (96000, 48000)])
def test_resampler_prototype_is_calc_resample(design, oracle, rates):
    want, L, M, ncoef, cpp = oracle.resample_taps(rates[0], rates[1], 0.0, 0, 1.0)
    out = np.zeros(ncoef + 8)
    lm = (C.c_int * 4)()
    n = design.t_resampler(C.c_int(rates[0]), C.c_int(rates[1]), C.c_double(0.0), C.c_int(0), C.c_double(1.0), out.ctypes.data_as(C.c_void_p),
                           C.c_int(out.size), lm)
    assert (n, lm[0], lm[1], lm[2], lm[3]) == (ncoef, L, M, ncoef, cpp)
    # the oracle returns the phase-major table h[L][cpp] (resample.c:69-72); the design keeps time order
    nat = want.reshape(L, cpp).T.reshape(-1) if want.size == ncoef else want
    assert np.abs(out[:ncoef] - nat).max() <= 4e-16 * np.abs(nat).max()


@pytest.mark.parametrize("curve,ctfmode", [(1, 0), (1, 1), (0, 0)])
def test_fc_impulse_is_even_linear_phase_and_follows_the_curve(design, curve, ctfmode):
    """The de-emphasis FIR (wdsp/fcurve.c:29-145 through fir_fsamp): symmetric taps, and a response that follows the 6 dB / octave line
    between f0 and f1.  (Its use inside xfmd is pinned against the oracle's chain in the -m gpu FM tests.)"""
    nc, f0, f1, fs = 2048, 300.0, 3000.0, 48000.0
    out = np.zeros(nc, dtype=np.complex128)
    design.t_fc_impulse(C.c_int(nc), C.c_double(f0), C.c_double(f1), C.c_double(20.0), C.c_int(curve), C.c_double(fs), C.c_double(1.0),
                        C.c_int(ctfmode), C.c_int(0), out.ctypes.data_as(C.c_void_p))
    h = out.real
    assert np.all(out.imag == 0.0) and np.abs(h - h[::-1]).max() <= 1e-18 + 1e-15 * np.abs(h).max()
    H = np.abs(np.fft.fft(h, 16 * nc))
    f = np.fft.fftfreq(16 * nc, 1.0 / fs)
    pick = lambda fr: H[np.argmin(np.abs(f - fr))]
    ratio = pick(1200.0) / pick(600.0)                   # one octave inside the band
    assert abs(ratio - (0.5 if curve == 1 else 2.0)) < 0.05
