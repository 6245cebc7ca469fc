"""Seeded walks over WDSP's display engine through its own names (XCreateAnalyzer / SetAnalyzer / Spectrum0 / Spectrum / Spectrum2 /
OpenBuffer+CloseBuffer / GetPixels / SetDisplay* / ResetPixelBuffers, wdsp/analyzer.c) against the restatement fed the same buffers:
geometry (transform size, buffer size, overlap, window, clip and span clip, pixel count, flip, real or complex input, write-ahead)
drawn at the start and again in mid-stream, and between buffers the detector, the averaging mode and its constants, the 1 Hz
normalisation and ResetPixelBuffers on any of up to four pixel outputs.  What is carried -- the input ring's indices, the averages'
sums and frame rings, the pixel buffers' read / write slots -- is the state a batched form gets wrong first.  Gates as in
tests/test_gpu_analyzer.py: the same flags from GetPixels, every pixel within one step of mlog10's table.  -m gpu."""
import os

import numpy as np
import pytest

from test_gpu_analyzer import RATE, _compare, _signal

pytestmark = pytest.mark.gpu
MAX_SIZE = 8192


def _geometry(rng, bf=None):
    size = int(rng.choice([512, 1024, 2048, 4096, 8192]))
    if bf is None:
        bf = int(rng.choice([size // 8, size // 4, size // 2, size]))
    typ = int(rng.integers(0, 4) != 0)                   # real input now and then
    out_size = size if typ else size // 2 + 1
    clip = int(rng.choice([0, 0, 7, out_size // 16]))
    span = out_size - 1 - 2 * clip
    fL, fH = float(rng.choice([0.0, 0.0, 3.25, span / 9.0])), float(rng.choice([0.0, 0.0, 5.5, span / 7.0]))
    overlap = int(rng.choice([0, size // 4, size // 2, size - size // 8]))
    npix = int(rng.choice([64, 300, 777, 1000, 2048, 3000]))
    win = int(rng.integers(0, 7))
    pi = float(rng.choice([0.0, 8.0, 14.0]))
    flip = int(rng.integers(0, 2))
    pixout = int(rng.choice([1, 1, 2, 4]))
    # (the reference's input ring holds dSAMP_BUFF_MULT = 2 times max_size samples, comm.h:134; a write-ahead beyond it lets a sub-span
    # that waits for its set be overwritten -- garbage in the reference itself)
    max_w = min(int(rng.choice([2, 4])) * max(size, bf) + bf, 2 * MAX_SIZE - bf)
    return (pixout, 1, typ, [flip], size, bf, win, pi, overlap, clip, fL, fH, npix, 1, 0, 0.0, 0.0, max_w)


def _setter(rng, pixout):
    o = int(rng.integers(0, pixout))
    k = int(rng.integers(0, 6))
    if k == 0: return ("SetDisplayDetectorMode", o, int(rng.integers(0, 5)))
    if k == 1: return ("SetDisplayAverageMode", o, int(rng.integers(-1, 4)))
    if k == 2: return ("SetDisplayAvBackmult", o, float(rng.choice([0.3, 0.7, 0.9, 0.97])))
    if k == 3: return ("SetDisplayNumAverage", o, int(rng.choice([1, 2, 3, 5, 8, 30])))
    if k == 4: return ("SetDisplayNormOneHz", o, int(rng.integers(0, 2)))
    return ("ResetPixelBuffers",)


@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_walk_over_the_display_engine(qh, oracle, seed):
    rng = np.random.default_rng(52000 + seed)
    args = _geometry(rng)
    a = oracle.OracleAnalyzer(MAX_SIZE)
    g = qh.WdspDisplay(20 + seed % 8, MAX_SIZE)
    log = [("SetAnalyzer", args)]
    try:
        for t in (a, g):
            t.SetDisplaySampleRate(RATE)
            t.SetAnalyzer(*args)
        x = _signal(64 * MAX_SIZE, seed)
        pos, rows = 0, 0
        for b in range(int(rng.integers(60, 120))):
            if b and rng.integers(0, 3) == 0:
                for _ in range(int(rng.integers(1, 3))):
                    s = _setter(rng, args[0])
                    log.append((b, s))
                    for t in (a, g):
                        getattr(t, s[0])(*s[1:])
            if b and rng.integers(0, 25) == 0:           # another geometry in mid-stream; the buffer size with it now and then
                args = _geometry(rng, bf=None if rng.integers(0, 2) else args[5])
                log.append((b, ("SetAnalyzer", args)))
                for t in (a, g):
                    t.SetAnalyzer(*args)
            bf = args[5]
            if pos + bf > x.size:
                break
            blk = x[pos:pos + bf]
            pos += bf
            buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
            i32, q32 = blk.real.astype(np.float32), blk.imag.astype(np.float32)
            way = int(rng.integers(0, 4))
            if way == 0:
                a.Spectrum0(1, 0, 0, buf); g.Spectrum0(1, 0, 0, buf)
            elif way == 1:
                a.Spectrum(0, 0, i32, q32); g.Spectrum(0, 0, i32, q32)
            elif way == 2:                               # Spectrum2 takes floats: the restatement gets what they hold
                f32 = buf.astype(np.float32)
                a.Spectrum0(1, 0, 0, f32.astype(np.float64)); g.Spectrum2(1, 0, 0, f32)
            else:
                a.Spectrum(0, 0, i32, q32); g.OpenCloseBuffer(0, 0, i32, q32)
            for o in range(args[0]):
                if rng.integers(0, 5) == 0:
                    continue                             # not every output is read every time: the pixel buffers' slots move on without the reader
                want, wflag = a.GetPixels(o)
                got, gflag = g.GetPixels(o)
                assert gflag == wflag, (seed, b, o, gflag, wflag, log)
                if wflag:
                    _compare(got, want, (seed, b, o, log))
                    rows += 1
        assert rows > 0, (seed, rows, log)               # (a large transform fed small buffers makes few frames)
    finally:
        g.close()


@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_random_walk_over_a_bank_of_displays(qh, oracle, seed):
    """The batched form (qh_ana_*): three displays of one configuration, one or two stitched sub-spans, SEVERAL buffers per call (so a
    call makes no, one or many frames, and the averages run on inside it), the setters and new geometries between calls.  The
    restatement is fed buffer by buffer in the same order, one per display; a buffer that completes several frames shows only its
    last row through GetPixels, so rows are matched by the restatement's frame count."""
    rng = np.random.default_rng(53000 + seed)
    ndisp, stitch = 3, int(rng.choice([1, 1, 2]))
    def geometry(bf=None):
        g = list(_geometry(rng, bf))
        g[13] = stitch
        return tuple(g)
    args = geometry()
    g = qh.AnalyzerBank(ndisp, MAX_SIZE, max_stitch=2)
    refs = [oracle.OracleAnalyzer(MAX_SIZE, 2) for _ in range(ndisp)]
    log = [("SetAnalyzer", args)]
    for t in [g] + refs:
        t.SetDisplaySampleRate(RATE)
        t.SetAnalyzer(*args)
    xs = np.stack([np.stack([_signal(48 * MAX_SIZE, 1000 * seed + 10 * d + s) for s in range(2)]) for d in range(ndisp)])     # [disp][ss][n]
    pos, rows = 0, 0
    for call in range(int(rng.integers(25, 50))):
        if call and rng.integers(0, 2) == 0:
            for _ in range(int(rng.integers(1, 3))):
                s = _setter(rng, args[0])
                log.append((call, s))
                for t in [g] + refs:
                    getattr(t, s[0])(*s[1:])
        if call and rng.integers(0, 12) == 0:
            args = geometry(bf=None if rng.integers(0, 2) else args[5])
            log.append((call, ("SetAnalyzer", args)))
            for t in [g] + refs:
                t.SetAnalyzer(*args)
        bf, pixout = args[5], args[0]
        nb = int(rng.choice([1, 1, 2, 3, 5, 9]))
        if pos + nb * bf > xs.shape[2]:
            break
        made, parts = 0, [[] for _ in range(pixout)]
        for s in range(stitch):                                          # (a call on the first sub-span can publish too: frames the other one had waiting)
            k = g.feed_host(s, np.ascontiguousarray(xs[:, s, pos:pos + nb * bf]))
            if k:
                for o in range(pixout):
                    r = g.rows_host(o)
                    assert r.shape[1] == k, (seed, call, s, o, r.shape, k)
                    parts[o].append(r)
            made += k
        got = [np.concatenate(p, axis=1) for p in parts] if made else None
        if os.environ.get("QH_TRACE"):
            print("call %d: %d buffers of %d per sub-span, %d rows" % (call, nb, bf, made), flush=True)
        for d in range(ndisp):
            a = refs[d]
            frames0, done = a.L.ao_frames(a.h), 0
            for s in range(stitch):                                     # (the same order of buffers as the bank's calls: sub-span by sub-span)
                for b in range(nb):
                    blk = xs[d, s, pos + b * bf:pos + (b + 1) * bf]
                    buf = np.empty(2 * bf); buf[0::2] = blk.imag; buf[1::2] = blk.real
                    a.Spectrum0(1, s, 0, buf)
                    now = a.L.ao_frames(a.h) - frames0
                    for o in range(pixout):
                        want, flag = a.GetPixels(o)
                        assert flag == int(now > done), (seed, call, d, s, b, o, flag, now, done)
                        if flag:
                            assert now <= made, (seed, call, d, s, b, now, made, log)
                            _compare(got[o][d][now - 1], want, (seed, call, d, s, b, o, log))
                            rows += 1
                    done = now
            assert done == made, (seed, call, d, done, made, log)
        pos += nb * bf
    assert rows > 0, (seed, log)
    g.close()
