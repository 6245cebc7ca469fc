"""The CPU restatement of wdsp/analyzer.c (oracle/analyzer_oracle.c) against numpy recomputation: PARITY UNPINNED by reference
execution (wdsp needs <fftw3.h>), so every block is rebuilt here from its definition."""
import numpy as np
import pytest

from oracle import pyoracle as po

RATE = 48000


def _mlog10(v):
    """wdsp/meterlog10.c: log10(2) * (exponent + log2(1 + mantissa truncated to 11 bits))"""
    v = np.asarray(v, dtype=np.float64)
    bits = v.view(np.uint64)
    e = ((bits >> np.uint64(52)) & np.uint64(2047)).astype(np.int64) - 1023
    m = ((bits >> np.uint64(41)) & np.uint64(2047)).astype(np.float64)
    return 0.301029995663981 * (e + np.log2(1.0 + m / 2048.0))


def _display(size=1024, bf=256, overlap=512, npix=256, typ=1, win=2, clip=0, fL=0.0, fH=0.0, flip=0, stitch=1, pixout=1, pi=0.0, max_size=None):
    a = po.OracleAnalyzer(max_size or size, stitch)
    a.SetDisplaySampleRate(RATE)
    a.SetAnalyzer(pixout, 1, typ, [flip], size, bf, win, pi, overlap, clip, fL, fH, npix, stitch, 0, 0.0, 0.0, 2 * size)
    return a


def _feed(a, x, ss=0):
    """x complex; Spectrum0 takes (Q, I) pairs; returns the rows GetPixels handed out, in order"""
    rows = []
    bf = a.buff_size
    for b in range(len(x) // bf):
        blk = x[b * bf:(b + 1) * bf]
        buf = np.empty(2 * bf)
        buf[0::2] = blk.imag
        buf[1::2] = blk.real
        a.Spectrum0(1, ss, 0, buf)
        pix, flag = a.GetPixels(0)
        if flag:
            rows.append(pix)
    return rows


@pytest.mark.parametrize("wtype", range(7))
def test_windows_are_the_textbook_ones_at_unit_coherent_gain(wtype):
    n = 1024
    a = _display(size=n, win=wtype, pi=9.0)
    w = a.window()
    k = np.arange(n)
    arg = 2 * np.pi * k / (n - 1)
    want = {0: np.ones(n),
            1: 0.35875 - 0.48829 * np.cos(arg) + 0.14128 * np.cos(2 * arg) - 0.01168 * np.cos(3 * arg),
            2: np.hanning(n), 4: np.hamming(n),
            3: 0.21557895 - 0.41663158 * np.cos(arg) + 0.277263158 * np.cos(2 * arg) - 0.083578947 * np.cos(3 * arg) + 0.006947368 * np.cos(4 * arg),
            5: np.kaiser(n, 9.0)}.get(wtype)
    if wtype == 6:          # 7-term Blackman-Harris given as a polynomial in cos: compare with its harmonic form
        c = np.cos(arg)
        want = sum(co * c ** p for p, co in enumerate([6.3964424114390378e-02, -2.3993864599352804e-01, 3.5015956323820469e-01,
                                                        -2.4774111897080783e-01, 8.5438256055858031e-02, -1.2320203369293225e-02,
                                                        4.3778825791773474e-04]))
    want = want * (n / want.sum())
    tol = 2e-7 if wtype == 5 else 1e-12         # the reference's I0 is the 1e-7 polynomial of Abramowitz & Stegun
    assert np.max(np.abs(w - want)) < tol * max(1.0, want.max())
    assert abs(w.mean() - 1.0) < 1e-12
    enb = n * np.sum(want ** 2) / np.sum(want) ** 2
    assert abs(a.GetDisplayENB() - enb) < 1e-6 * enb


def test_tone_level_position_and_frame_schedule():
    size, bf, ov, npix = 4096, 1024, 2048, 1024
    a = _display(size, bf, ov, npix)
    t = np.arange(16 * bf)
    x = 0.1 * np.exp(2j * np.pi * 3000.0 / RATE * t)
    got = []
    for b in range(16):
        blk = x[b * bf:(b + 1) * bf]
        buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
        a.Spectrum0(1, 0, 0, buf)
        got.append(a.GetPixels(0)[1])
    # first frame when 4096 samples are in, then one every size - overlap = 2048 samples
    assert got == [0, 0, 0, 1] + [0, 1] * 6
    rows = _feed(_display(size, bf, ov, npix), x)
    for r in rows:
        assert abs(r.max() - (-20.0)) < 0.01            # 0.1 amplitude, unit coherent gain: -20 dB
        assert r.argmax() == npix // 2 + 3000 * npix // RATE


def test_complex_frame_against_numpy_all_detectors():
    size, bf, npix = 2048, 512, 300
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(4 * size) + 1j * rng.standard_normal(4 * size)) * 0.05
    x += 0.3 * np.exp(2j * np.pi * 0.123 * np.arange(x.size))
    xf = x.real.astype(np.float32).astype(np.float64) + 1j * x.imag.astype(np.float32).astype(np.float64)    # dINREAL is float
    for clip, fL, fH, flip in ((0, 0.0, 0.0, 0), (40, 0.0, 0.0, 0), (10, 25.5, 13.25, 0), (10, 25.5, 13.25, 1)):
        for det in range(5):
            a = _display(size, bf, 0, npix, clip=clip, fL=fL, fH=fH, flip=flip)
            a.SetDisplayDetectorMode(0, det)
            rows = _feed(a, x)
            assert len(rows) == 4
            w = a.window()
            inv_enb = 1.0 / a.GetDisplayENB()
            for f, row in enumerate(rows):
                X = np.fft.fft(w * xf[f * size:(f + 1) * size])
                P = np.abs(X) ** 2
                if flip:
                    P = P[::-1]                         # bin i is read at out_size - 1 - i
                # display order: bins size/2 + 1 + clip + fscL ... size - 1, then 0 ... size/2 - clip - fscH - 1
                bins = np.concatenate([P[size // 2 + 1 + clip + int(fL):], P[:size // 2 - clip - int(fH)]])
                m = bins.size
                span = size - 1 - 2 * clip
                ppb = npix / (span - fL - fH - 1.0)
                off = -ppb * (fL - np.floor(fL))
                imin = 0 if fL == np.floor(fL) else 1
                ilim = m if fH == np.floor(fH) else m - 1
                idx = np.arange(imin, ilim)
                pc = np.minimum((off + idx * ppb).astype(np.int64), npix - 1)
                want = np.zeros(npix)
                if det == 1:
                    want = po.analyzer_detector(1, bins, npix, ppb, 0.0, inv_enb, fL, fH, off)      # checked by its own test below
                else:
                    for p in range(npix):
                        sel = bins[idx[pc == p]]
                        if det == 0: want[p] = sel.max()
                        elif det == 2: want[p] = sel.mean() * inv_enb
                        elif det == 3: want[p] = sel[len(sel) - 1 - len(sel) // 2] * inv_enb
                        else: want[p] = np.sqrt(np.mean(sel ** 2)) * inv_enb
                db = 10.0 * _mlog10(want / size ** 2 + 1e-60)
                assert np.max(np.abs(row - db.astype(np.float32))) < 2.2e-3, (clip, fL, det, f)     # one mlog10 table step
                assert np.mean(np.abs(row - db.astype(np.float32)) > 1e-5) < 0.02


def test_rosenfell_alternates_minima_and_maxima_of_noisy_pixels():
    rng = np.random.default_rng(5)
    m, npix = 4000, 100
    bins = rng.random(m) + 1.0
    ppb = npix / (m - 1.0)
    out = po.analyzer_detector(1, bins, npix, ppb, 0.0, 1.0, 0.0, 0.0, 0.0)
    pc = np.minimum((np.arange(m) * ppb).astype(np.int64), npix - 1)
    for p in range(1, npix - 1):
        sel = bins[pc == p]
        if p & 1:
            prev = bins[pc == p - 1].max()
            assert out[p] == max(prev, sel.max())
        else:
            assert out[p] == sel.min()
    ramp = np.arange(m, dtype=np.float64)          # monotone: never "rose and fell", every pixel is its maximum
    out = po.analyzer_detector(1, ramp, npix, ppb, 0.0, 1.0, 0.0, 0.0, 0.0)
    assert all(out[p] == ramp[pc == p].max() for p in range(npix))


def test_more_pixels_than_bins_is_linear_interpolation():
    size, npix = 512, 1500
    a = _display(size, 256, 0, npix)
    x = 0.2 * np.exp(2j * np.pi * 0.031 * np.arange(size)) + 0.01
    (row,) = _feed(a, x)
    xf = x.real.astype(np.float32).astype(np.float64) + 1j * x.imag.astype(np.float32).astype(np.float64)
    P = np.abs(np.fft.fft(a.window() * xf)) ** 2
    bins = np.concatenate([P[size // 2 + 1:], P[:size // 2]])
    bpp = (size - 1 - 1.0) / (npix - 1.0)
    pos = np.arange(npix) * bpp
    want = np.interp(pos, np.arange(bins.size), bins)
    db = 10.0 * _mlog10(want / size ** 2 + 1e-60)
    assert np.max(np.abs(row - db.astype(np.float32))) < 2.2e-3


def test_real_input_uses_half_the_spectrum_and_four_times_the_scale():
    size, npix = 1024, 200
    a = _display(size, 256, 0, npix, typ=0)
    t = np.arange(size)
    x = 0.5 * np.cos(2 * np.pi * 100 / size * t) + 0j
    (row,) = _feed(a, x)
    # a real cosine of amplitude 0.5 holds 0.25 per side; scale 4 / N^2 reports the full amplitude: 20 log10(0.5)
    assert abs(row.max() - 20 * np.log10(0.5)) < 0.01
    assert row.argmax() == int(100 * npix / (size // 2 - 1.0))


def test_averaging_modes_follow_their_recurrences():
    size, bf, npix = 512, 512, 128
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(12 * size) + 1j * rng.standard_normal(12 * size)) * 0.1
    base = _feed(_display(size, bf, 0, npix), x)
    lin = [10.0 ** (np.asarray(r, dtype=np.float64) / 10.0) for r in base]        # scale * t, to table accuracy
    for mode in (-1, 1, 2, 3):
        a = _display(size, bf, 0, npix)
        a.SetDisplayAverageMode(0, mode)
        a.SetDisplayAvBackmult(0, 0.8)
        a.SetDisplayNumAverage(0, 4)
        rows = _feed(a, x)
        acc = {1: 1e-12 / size ** 2 * np.ones(npix), 3: -160.0 * np.ones(npix)}.get(mode, np.zeros(npix))
        hist = []
        for f, r in enumerate(rows):
            if mode == -1: acc = np.maximum(acc, lin[f]); want = 10 * np.log10(acc)
            elif mode == 1: acc = 0.8 * acc + 0.2 * lin[f]; want = 10 * np.log10(acc)
            elif mode == 3: acc = 0.8 * acc + 0.2 * 10 * np.log10(lin[f]); want = acc
            else:
                hist.append(lin[f]); want = 10 * np.log10(np.mean(hist[-4:], axis=0))
            assert np.max(np.abs(r - want)) < 0.01, (mode, f)


def test_two_stitched_sub_spans_and_one_hz_normalisation():
    size, bf, npix, clip = 1024, 256, 400, 100
    a = _display(size, bf, 0, npix, clip=clip, stitch=2)
    a.SetDisplayNormOneHz(0, 1)
    t = np.arange(size)
    lo = 0.1 * np.exp(2j * np.pi * 51 / size * t)            # on bin centres: no scalloping loss
    hi = 0.2 * np.exp(-2j * np.pi * 72 / size * t)
    for b in range(size // bf):
        for ss, sig in ((0, lo), (1, hi)):
            blk = sig[b * bf:(b + 1) * bf]
            buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
            a.Spectrum0(1, ss, 0, buf)
    row, flag = a.GetPixels(0)
    assert flag == 1 and a.frames() == 1
    norm = 10 * np.log10(1.0 / (RATE / size))
    half = npix // 2
    assert abs(row[:half].max() - (-20.0 + norm)) < 0.05 and abs(row[half:].max() - (20 * np.log10(0.2) + norm)) < 0.05
    per = size - 1 - 2 * clip                               # bins per sub-span
    assert abs(row[:half].argmax() - (per // 2 + 51) * npix / (2 * per)) <= 1
    assert abs(half + row[half:].argmax() - (per + per // 2 - 72) * npix / (2 * per)) <= 1


def test_calibration_spline_goes_through_its_points():
    size, npix = 512, 101
    a = po.OracleAnalyzer(size)
    a.SetDisplaySampleRate(RATE)
    table = np.array([[0.0, 1.0], [3000.0, 1.5], [1000.0, 1.2], [2000.0, 0.9], [4000.0, 1.1], [4000.0, 1.1]])
    a.SetCalibration(0, table)
    a.SetAnalyzer(1, 1, 1, [0], size, 256, 2, 0.0, 0, 0, 0.0, 0.0, npix, 1, 0, 0.0, 4000.0, 2 * size)
    cd = a.cd()
    f = np.linspace(0.0, 4000.0, npix)
    for fr, v in ((0.0, 1.0), (1000.0, 1.2), (2000.0, 0.9), (3000.0, 1.5), (4000.0, 1.1)):
        assert abs(cd[np.argmin(np.abs(f - fr))] - v * v) < 1e-9
    assert np.all(np.abs(np.diff(np.sqrt(cd), 2)) < 0.01)          # smooth between them
