"""WDSP's display engine on the GPU (quisk_amd/csrc/qh_analyzer.hip) against the CPU restatement of wdsp/analyzer.c.

Pixels are float dB from mlog10, a table of the top 11 mantissa bits: its steps are 10 log10(1 + 1/2048) = 0.0021 dB, and a
last-bit difference of the transform can move a value across a step.  The gates: every pixel within one step, at most 2 % of
the pixels more than 1e-4 dB apart.  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RATE = 192000
STEP = 0.0022


def _signal(n, seed, tones=((0.0371, 0.3), (-0.211, 0.05), (0.4031, 0.01))):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.003
    t = np.arange(n)
    for f, a in tones:
        x += a * np.exp(2j * np.pi * f * t)
    return x


def _compare(got, want, what):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    d = np.abs(got - want)
    assert d.max() < STEP, (what, d.max(), int(d.argmax()))
    assert np.mean(d > 1e-4) < 0.02, (what, np.mean(d > 1e-4))


def _oracle_rows(oracle, cfg, xs, setup=None):
    """xs: [nss][n] complex.  Every row GetPixels hands out while the streams are fed buffer by buffer, ss in turn."""
    a = oracle.OracleAnalyzer(cfg["max_size"], cfg["stitch"])
    a.SetDisplaySampleRate(RATE)
    if setup:
        setup(a)
    a.SetAnalyzer(*cfg["args"])
    bf = cfg["bf"]
    rows = [[] for _ in range(cfg["pixout"])]
    for b in range(xs.shape[1] // bf):
        for ss in range(cfg["stitch"]):
            blk = xs[ss, b * bf:(b + 1) * bf]
            buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
            a.Spectrum0(1, ss, 0, buf)
        for o in range(cfg["pixout"]):
            pix, flag = a.GetPixels(o)
            if flag:
                rows[o].append(pix)
    return [np.array(r) for r in rows]


def _cfg(size, bf, overlap, npix, typ=1, win=2, clip=0, fL=0.0, fH=0.0, flip=0, stitch=1, pixout=1, pi=0.0, calset=0, fmin=0.0, fmax=0.0):
    return {"max_size": size, "stitch": stitch, "bf": bf, "pixout": pixout,
            "args": (pixout, 1, typ, [flip], size, bf, win, pi, overlap, clip, fL, fH, npix, stitch, calset, fmin, fmax, 2 * size)}


@pytest.mark.parametrize("size", [512, 1024, 2048, 4096, 8192, 16384, 65536])
def test_every_transform_size_peak_detector(qh, oracle, size):
    bf = size // 4
    cfg = _cfg(size, bf, size // 2, 1000)
    x = _signal(4 * size, size)[None, :]
    want = _oracle_rows(oracle, cfg, x)[0]
    g = qh.AnalyzerBank(1, size)
    g.SetDisplaySampleRate(RATE)
    g.SetAnalyzer(*cfg["args"])
    frames = g.feed_host(0, x)
    assert frames == want.shape[0] == 7
    _compare(g.rows_host(0)[0], want, size)


@pytest.mark.parametrize("det", range(5))
@pytest.mark.parametrize("geom", [dict(), dict(clip=100, fL=37.5, fH=12.25), dict(clip=100, fL=37.5, fH=12.25, flip=1), dict(typ=0, clip=20, fL=3.0)])
def test_detectors_clip_flip_and_real_input(qh, oracle, det, geom):
    size, bf, npix = 4096, 1024, 777
    cfg = _cfg(size, bf, 1024, npix, **geom)
    x = _signal(5 * size, 11 + det)[None, :]
    want = _oracle_rows(oracle, cfg, x, setup=lambda a: a.SetDisplayDetectorMode(0, det))[0]
    g = qh.AnalyzerBank(1, size)
    g.SetDisplaySampleRate(RATE)
    g.SetDisplayDetectorMode(0, det)
    g.SetAnalyzer(*cfg["args"])
    got = []
    for k in range(0, x.shape[1], 2 * bf):              # two buffers per call
        if g.feed_host(0, x[:, k:k + 2 * bf]):
            got.append(g.rows_host(0)[0])
    _compare(np.concatenate(got), want, (det, geom))


@pytest.mark.parametrize("det", [0, 2, 3])
def test_more_pixels_than_bins(qh, oracle, det):
    size, bf, npix = 1024, 512, 3000
    cfg = _cfg(size, bf, 0, npix, clip=30, fL=10.75, fH=2.5)
    x = _signal(3 * size, 5)[None, :]
    want = _oracle_rows(oracle, cfg, x, setup=lambda a: a.SetDisplayDetectorMode(0, det))[0]
    g = qh.AnalyzerBank(1, size)
    g.SetDisplaySampleRate(RATE)
    g.SetDisplayDetectorMode(0, det)
    g.SetAnalyzer(*cfg["args"])
    assert g.feed_host(0, x) == 3
    _compare(g.rows_host(0)[0], want, det)


@pytest.mark.parametrize("mode", [-1, 0, 1, 2, 3])
def test_averaging_modes_across_calls_and_four_outputs(qh, oracle, mode):
    size, bf, npix = 2048, 512, 640
    cfg = _cfg(size, bf, 1536, npix, pixout=4, win=6)
    x = _signal(40 * bf, 21)[None, :]

    def setup(a):
        for o, (det, norm) in enumerate(((0, 0), (2, 1), (0, 1), (4, 0))):
            a.SetDisplayDetectorMode(o, det)
            a.SetDisplayAverageMode(o, mode)
            a.SetDisplayAvBackmult(o, 0.7 + 0.05 * o)
            a.SetDisplayNumAverage(o, 3 + 2 * o)
            a.SetDisplayNormOneHz(o, norm)
    want = _oracle_rows(oracle, cfg, x, setup=setup)
    g = qh.AnalyzerBank(1, size)
    g.SetDisplaySampleRate(RATE)
    setup(g)
    g.SetAnalyzer(*cfg["args"])
    got = [[] for _ in range(4)]
    pos = 0
    for nb in (1, 3, 7, 2, 11, 16):                     # buffers per call: 0, 1, several rows per call
        if g.feed_host(0, x[:, pos * bf:(pos + nb) * bf]):
            for o in range(4):
                got[o].append(g.rows_host(o)[0])
        pos += nb
    assert pos == 40
    for o in range(4):
        _compare(np.concatenate(got[o]), want[o], (mode, o))


def test_stitched_sub_spans_in_a_bank_of_displays(qh, oracle):
    size, bf, npix, ndisp = 1024, 256, 500, 3
    cfg = _cfg(size, bf, 512, npix, clip=64, fL=100.5, fH=7.0, stitch=2, win=1)
    xs = np.stack([np.stack([_signal(6 * size, 100 + 10 * d + ss) for ss in range(2)]) for d in range(ndisp)])    # [disp][ss][n]
    g = qh.AnalyzerBank(ndisp, size, max_stitch=2)
    g.SetDisplaySampleRate(RATE)
    g.SetAnalyzer(*cfg["args"])
    got = []
    for k in range(0, xs.shape[2], 3 * bf):
        assert g.feed_host(0, xs[:, 0, k:k + 3 * bf]) == 0              # nothing is published before the second sub-span has its share
        if g.feed_host(1, xs[:, 1, k:k + 3 * bf]):
            got.append(g.rows_host(0))
    got = np.concatenate(got, axis=1)
    for d in range(ndisp):
        _compare(got[d], _oracle_rows(oracle, cfg, xs[d])[0], d)
    pix, flag = g.GetPixels(1, 0)
    assert flag == 1 and np.array_equal(pix, got[1][-1])
    assert g.GetPixels(1, 0)[1] == 0 and g.GetPixels(2, 0)[1] == 1      # read once per display


def test_calibration_and_kaiser_window(qh, oracle):
    size, bf, npix = 4096, 1024, 800
    table = np.array([[0.0, 1.0], [30000.0, 1.6], [10000.0, 1.25], [20000.0, 0.8], [48000.0, 1.1], [60000.0, 0.7], [96000.0, 1.0]])
    cfg = _cfg(size, bf, 0, npix, win=5, pi=11.5, calset=1, fmin=0.0, fmax=96000.0)
    x = _signal(2 * size, 8)[None, :]
    want = _oracle_rows(oracle, cfg, x, setup=lambda a: a.SetCalibration(1, table))[0]
    g = qh.AnalyzerBank(1, size)
    g.SetDisplaySampleRate(RATE)
    g.SetCalibration(1, table)
    g.SetAnalyzer(*cfg["args"])
    assert g.feed_host(0, x) == 2
    _compare(g.rows_host(0)[0], want, "cal")
    o = oracle.OracleAnalyzer(size); o.SetDisplaySampleRate(RATE); o.SetAnalyzer(*cfg["args"])
    assert abs(g.GetDisplayENB() - o.GetDisplayENB()) < 1e-12 * o.GetDisplayENB()


def test_wdsp_named_exports_buffer_by_buffer(qh, oracle):
    """XCreateAnalyzer / SetAnalyzer / Spectrum0 / Spectrum / Spectrum2 / OpenBuffer+CloseBuffer / GetPixels: same flags, same rows"""
    size, bf, npix = 2048, 512, 400
    cfg = _cfg(size, bf, 1024, npix)
    x = _signal(12 * bf, 77)
    a = oracle.OracleAnalyzer(size)
    a.SetDisplaySampleRate(RATE)
    a.SetDisplayAverageMode(0, 1); a.SetDisplayAvBackmult(0, 0.5)
    a.SetAnalyzer(*cfg["args"])
    g = qh.WdspDisplay(7, size)
    try:
        g.SetDisplaySampleRate(RATE)
        g.SetDisplayAverageMode(0, 1); g.SetDisplayAvBackmult(0, 0.5)
        g.SetAnalyzer(*cfg["args"])
        nrows = 0
        for b in range(12):
            blk = x[b * bf:(b + 1) * bf]
            buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
            a.Spectrum0(1, 0, 0, buf)
            i32, q32 = blk.real.astype(np.float32), blk.imag.astype(np.float32)
            if b % 4 == 0: g.Spectrum0(1, 0, 0, buf)
            elif b % 4 == 1: g.Spectrum(0, 0, i32, q32)
            elif b % 4 == 2: g.Spectrum2(1, 0, 0, buf.astype(np.float32))
            else: g.OpenCloseBuffer(0, 0, i32, q32)
            want, wflag = a.GetPixels(0)
            got, gflag = g.GetPixels(0)
            assert gflag == wflag, b
            if wflag:
                _compare(got, want, b)
                nrows += 1
        assert nrows == 5
    finally:
        g.close()


def test_writeahead_skip_reset_and_reconfiguration(qh, oracle):
    """The corners of the ring bookkeeping: a max_writeahead smaller than a frame (the reference then keeps skipping samples,
    analyzer.c:1428-1434, and never fills one), ResetPixelBuffers in mid-stream, and further SetAnalyzer calls with another
    size / pixel count / detector on the same display."""
    size, bf, npix = 2048, 512, 300
    starve = (1, 1, 1, [0], size, bf, 1, 0.0, 0, 0, 0.0, 0.0, npix, 1, 0, 0.0, 0.0, 1024)
    normal = starve[:-1] + (4 * size,)
    x = _signal(48 * bf, 31)

    def fresh_oracle(args):
        o = oracle.OracleAnalyzer(8192)
        o.SetDisplaySampleRate(RATE)
        o.SetDisplayAverageMode(0, 2); o.SetDisplayNumAverage(0, 3)
        o.SetAnalyzer(*args)
        return o
    a = fresh_oracle(starve)
    g = qh.WdspDisplay(9, 8192)
    try:
        g.SetDisplaySampleRate(RATE)
        g.SetDisplayAverageMode(0, 2); g.SetDisplayNumAverage(0, 3)
        g.SetAnalyzer(*starve)
        rows = []

        def push(b):
            blk = x[b * bf:(b + 1) * bf]
            buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
            a.Spectrum0(1, 0, 0, buf); g.Spectrum0(1, 0, 0, buf)
            want, wflag = a.GetPixels(0)
            got, gflag = g.GetPixels(0)
            assert gflag == wflag, b
            if wflag:
                _compare(got, want, b)
                rows.append(b)
        for b in range(8):
            push(b)
        assert rows == []                               # 1024 samples of write-ahead never make a 2048-sample frame
        a.SetAnalyzer(*normal); g.SetAnalyzer(*normal)
        for b in range(8, 20):
            push(b)
        assert len(rows) == 3
        a.ResetPixelBuffers(); g.ResetPixelBuffers()    # (window averaging keeps its running sum across the reset, as in the reference)
        for b in range(20, 30):
            push(b)
        assert len(rows) == 5
        # another transform size, pixel count and detector on the same display
        args2 = (1, 1, 1, [0], 4096, bf, 4, 0.0, 2048, 16, 5.5, 0.0, 777, 1, 0, 0.0, 0.0, 8192)
        a.SetDisplayDetectorMode(0, 2); g.SetDisplayDetectorMode(0, 2)
        a.SetDisplayAverageMode(0, 1); g.SetDisplayAverageMode(0, 1)
        a.SetDisplayAvBackmult(0, 0.6); g.SetDisplayAvBackmult(0, 0.6)
        a.SetAnalyzer(*args2); g.SetAnalyzer(*args2)
        for b in range(30, 48):
            push(b)
        assert len(rows) == 5 + 3
    finally:
        g.close()


def test_snap_spectrum_hands_out_the_next_frames_transform(qh, oracle):
    """SnapSpectrum (analyzer.c:1337-1346, served in Cspectra :708-713): the complex transform of the NEXT frame of (display, sub-span),
    fft-shifted -- armed, then taken after the feed call that made the frame; nothing while no frame has been made; through the
    WDSP-named entry points with a feeding thread (the call blocks like the reference's) and with its time-out."""
    import ctypes as C
    import threading
    size, bf, npix, ndisp = 4096, 1024, 600, 3
    cfg = _cfg(size, bf, size // 2, npix, stitch=2)
    xs = np.stack([np.stack([_signal(6 * size, 40 + 5 * d + s) for s in range(2)]) for d in range(ndisp)])      # [disp][ss][n]
    g = qh.AnalyzerBank(ndisp, size, 2)
    g.SetDisplaySampleRate(RATE)
    g.SetAnalyzer(*cfg["args"])
    a = oracle.OracleAnalyzer(size, 2)
    a.SetDisplaySampleRate(RATE)
    a.SetAnalyzer(*cfg["args"])
    d, ss = 1, 1

    def feed_both(b0, b1):
        for b in range(b0, b1):
            for s in range(2):
                g.feed_host(s, np.ascontiguousarray(xs[:, s, b * bf:(b + 1) * bf]))
                blk = xs[d, s, b * bf:(b + 1) * bf]
                buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
                a.Spectrum0(1, s, 0, buf)
    feed_both(0, 5)                                       # frames have been made; nothing was asked for
    assert g.SnapSpectrum_take(size) is None
    g.SnapSpectrum_arm(d, ss)
    want = a.SnapSpectrum(ss)
    assert g.SnapSpectrum_take(size) is None              # armed, no new frame yet
    feed_both(5, 9)
    got = g.SnapSpectrum_take(size)
    assert got is not None and a.snap_taken() == 1
    w = want.view(np.complex128)
    assert np.abs(w).max() > 1.0
    assert np.abs(got - w).max() < 1e-11 * np.abs(w).max()
    assert g.SnapSpectrum_take(size) is None              # handed out once
    with pytest.raises(qh.QuiskHipError):
        g.SnapSpectrum_arm(ndisp, 0)
    # the WDSP-named calls: one display per id, SnapSpectrum blocks until another thread's Spectrum0 completes a frame
    lib = qh.load()
    ok = C.c_int(0)
    lib.XCreateAnalyzer(7, C.byref(ok), size, 1, 1, None)
    assert ok.value == 0
    flp = (C.c_int * 1)(0)
    lib.SetAnalyzer.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] + [C.c_int] * 3 + [C.c_double] + [C.c_int] * 2 + [C.c_double] * 2 + [C.c_int] * 3 + [C.c_double] * 2 + [C.c_int]
    lib.SetAnalyzer(7, 1, 1, 1, flp, size, bf, 2, 0.0, size // 2, 0, 0.0, 0.0, npix, 1, 0, 0.0, 0.0, 2 * size)
    lib.Spectrum0.argtypes = [C.c_int] * 4 + [C.c_void_p]
    lib.SnapSpectrum.argtypes = [C.c_int] * 3 + [C.c_void_p]
    lib.SnapSpectrumTimeout.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_uint, C.POINTER(C.c_int)]
    snap = np.zeros(2 * size)
    flag = C.c_int(5)
    lib.SnapSpectrumTimeout(7, 0, 0, snap.ctypes.data, 50, C.byref(flag))
    assert flag.value == 0                                # nobody feeds: the time-out, and the request is withdrawn
    a1 = oracle.OracleAnalyzer(size, 1)
    a1.SetDisplaySampleRate(RATE)
    a1.SetAnalyzer(1, 1, 1, [0], size, bf, 2, 0.0, size // 2, 0, 0.0, 0.0, npix, 1, 0, 0.0, 0.0, 2 * size)
    want1 = a1.SnapSpectrum(0)
    x1 = xs[0, 0]

    def feeder():
        import time
        time.sleep(0.2)                                   # SnapSpectrum is waiting by now (if not, the snap is armed at the latest by then)
        for b in range(6):
            blk = x1[b * bf:(b + 1) * bf]
            buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
            lib.Spectrum0(1, 7, 0, 0, buf.ctypes.data)
    th = threading.Thread(target=feeder)
    th.start()
    lib.SnapSpectrum(7, 0, 0, snap.ctypes.data)          # returns with the first frame the feeder completes
    th.join()
    for b in range(6):
        blk = x1[b * bf:(b + 1) * bf]
        buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
        a1.Spectrum0(1, 0, 0, buf)
    w1 = want1.view(np.complex128)
    assert a1.snap_taken() == 1 and np.abs(snap.view(np.complex128) - w1).max() < 1e-11 * np.abs(w1).max()
    lib.DestroyAnalyzer(7)
