"""oracle/snba_oracle.c (WDSP's SNBA restated: wdsp/snb.c + wdsp/lmath.c) against independent numpy statements of its pieces
and against what the block is for.  The reference itself cannot be built here (wdsp needs <fftw3.h>): parity unpinned, so these
checks are what stands behind the oracle.  CPU only."""
import ctypes as C

import numpy as np

from oracle import pyoracle as po


def _lib():
    L = po.lib()
    L.wo_snba_create.restype = C.c_void_p; L.wo_snba_create.argtypes = [C.c_int, C.c_int]
    L.wo_snba_free.argtypes = [C.c_void_p]
    L.wo_snba_frame.argtypes = [C.c_void_p, C.c_void_p]
    L.wo_snba_asolve.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.wo_snba_median.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    return L


def test_asolve_is_the_autocorrelation_method_lpc():
    L = _lib()
    rng = np.random.default_rng(5)
    n, p = 256, 64
    buf = rng.standard_normal(2 * n)
    buf = np.convolve(buf, [1.0, -1.6, 0.9])[:2 * n].copy()          # something with a spectrum
    a = np.zeros(p)
    L.wo_snba_asolve(n, p, buf.ctypes.data + n * 8, a.ctypes.data)
    x = buf[n:]
    r = np.array([np.dot(x, buf[n - i:2 * n - i]) for i in range(p + 1)])       # x[j] * x[j - i], history included (lmath.c:104-105)
    R = np.array([[r[abs(i - j)] for j in range(p)] for i in range(p)])
    want = np.linalg.solve(R, r[1:])
    assert np.max(np.abs(a - want)) < 1e-8 * np.max(np.abs(want))


def test_median_is_the_element_of_rank_n_over_2():
    L = _lib()
    rng = np.random.default_rng(6)
    for n in (192, 193, 5, 2):
        v = rng.random(n)
        v[::7] = v[0]                                                 # ties
        w = v.copy()
        m = C.c_double()
        L.wo_snba_median(n, w.ctypes.data, C.byref(m))
        assert m.value == np.sort(v)[n // 2]


def test_a_frame_with_clicks_is_repaired_to_the_noise_floor():
    L = _lib()
    d = L.wo_snba_create(48000, 256)
    rng = np.random.default_rng(1)
    t = np.arange(512) / 12000.0
    clean = np.cos(2 * np.pi * 1000 * t) + 0.5 * np.cos(2 * np.pi * 1700 * t + 1) + 0.001 * rng.standard_normal(512)
    x = clean.copy()
    x[256 + 100] += 10; x[256 + 101] -= 7; x[256 + 180] += 3
    hit = x.copy()
    L.wo_snba_frame(d, x.ctypes.data + 256 * 8)
    assert np.array_equal(x[:256], hit[:256])                         # the history in front of the frame is read, not written
    assert np.max(np.abs(x[256:] - clean[256:])) < 0.01               # clicks of 3 .. 10 down to a few times the noise
    untouched = np.ones(256, bool); untouched[95:107] = False; untouched[175:187] = False
    assert np.array_equal(x[256:][untouched], hit[256:][untouched])   # and nothing else was changed
    y = clean.copy()
    L.wo_snba_frame(d, y.ctypes.data + 256 * 8)
    assert np.array_equal(y, clean)                                   # a clean frame passes bit for bit
    L.wo_snba_free(d)


def test_the_chain_with_snba_is_transparent_for_a_clean_carrier_and_takes_the_clicks_out():
    n = 1024 * 120
    t = np.arange(n) / 48000.0
    rng = np.random.default_rng(3)
    clean = np.exp(-2j * np.pi * 1000 * t) + 0.5 * np.exp(-2j * np.pi * 1700 * t) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    hit = clean.copy()
    k = rng.integers(0, n, 20)
    hit[k] += 100 * (rng.standard_normal(20) + 1j * rng.standard_normal(20))

    def run(sig, snba):
        c = po.WdspChannel(1024, 256, 48000, 48000, 48000)
        c.SetRXAMode(1); c.RXASetPassband(150.0, 4150.0); c.SetRXAAGCMode(0)
        if snba:
            c.SetRXASNBARun(1)
        return c.xrxa(sig)

    d = 8000
    rms = lambda v: float(np.sqrt(np.mean(np.abs(v[d:]) ** 2)))
    yc0, yc1, y0, y1 = run(clean, 0), run(clean, 1), run(hit, 0), run(hit, 1)
    assert abs(rms(yc1) / rms(yc0) - 1.0) < 1e-3                      # bpsnba + both resamplers + bp1 (gain 2 on the real part): unity
    assert np.all(yc1.imag[d:] != 0.0)                                # bp1 rebuilt the analytic signal behind the real-valued blanker
    assert np.abs(y0 - yc0)[d:].max() > 3.0 * np.abs(y1 - yc1)[d:].max()
    assert rms(y0 - yc0) > 2.0 * rms(y1 - yc1)
