"""oracle/quisk_oracle.c qo_bscope_* / qo_copy2pixels (quisk.c:4932-5011) against numpy.  PARITY UNPINNED by reference
execution (quisk.c needs <fftw3.h>); these pin the restatement."""
import numpy as np


def test_bandscope_matches_numpy(oracle):
    N, W, clock = 4096, 600, 122880000
    rng = np.random.default_rng(0)
    b = oracle.OracleBandscope(N, W)
    assert b.get(clock) is None
    j = np.arange(N) - N // 2
    win = 0.5 + 0.5 * np.cos(2 * np.pi * j / N)
    avg = np.zeros(N // 2 + 2)
    mx = 0.0
    for k in range(3):
        x = rng.uniform(-0.3, 0.3, N) + 0.5 * np.cos(2 * np.pi * 700.3 * np.arange(N) / N)
        b.block(x)
        avg[:N // 2 + 1] += np.abs(np.fft.rfft(x * win))
        mx = max(mx, np.abs(x).max())
    pix, adc, cnt = b.get(clock, 1.0, 0.0)
    assert cnt == 3 and adc == mx
    L = N // 2 + 1
    edges = L * np.arange(W + 1) / W
    want = np.empty(W)
    cs = np.concatenate([[0.0], np.cumsum(avg)])
    area = lambda d: cs[np.floor(d).astype(int)] + (d - np.floor(d)) * avg[np.minimum(np.floor(d).astype(int), L)]
    want = (area(edges[1:]) - area(edges[:-1])) / (L / W) / 3 / N
    assert np.abs(pix - 20 * np.log10(want)).max() < 1e-9
    assert abs(np.argmax(pix) - 700.3 / L * W) < 1.5
    assert b.get(clock) is None


def test_copy2pixels_zoom_is_a_sub_range(oracle):
    import ctypes as C
    L = oracle.lib()
    L.qo_copy2pixels.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    L.qo_copy2pixels.restype = None
    fft = np.arange(1, 1026, dtype=np.float64)
    fft_p = np.concatenate([fft, [0.0]])
    full = np.empty(100); L.qo_copy2pixels(full.ctypes.data, 100, fft_p.ctypes.data, 1025, 1.0, 0.0, 61.44e6)
    assert abs(full.sum() - fft.sum()) < 1e-6                      # the pixels tile the spectrum
    z = np.empty(100); L.qo_copy2pixels(z.ctypes.data, 100, fft_p.ctypes.data, 1025, 0.5, 0.0, 61.44e6)
    lo, hi = 1025 * 0.25, 1025 * 0.75
    assert abs(z.sum() - (hi * (hi + 1) / 2 - lo * (lo + 1) / 2)) < 0.5 * 1025     # the middle half, up to edge bins
