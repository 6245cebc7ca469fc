"""Independent checks of the panadapter restatement (quisk.c get_graph) with numpy."""
import numpy as np


def test_graph_against_numpy(oracle):
    N, W, fs = 4096, 1024, 192000.0
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(N * 5 + 77) + 1j * rng.standard_normal(N * 5 + 77)) * 2.0 ** 24
    x += 2.0 ** 28 * np.exp(2j * np.pi * 20000.0 / fs * np.arange(x.size))
    g = oracle.OracleGraph(N, W, fs)
    g.set_smeter_band(18650.0, 2700.0)
    assert g.feed(x[:1000]) == 0 and g.get() is None          # no complete block yet: get_graph returns None
    assert g.feed(x[1000:]) == 5
    pix, sm, cnt = g.get(1.0, 0.0)
    assert cnt == 5 and g.get() is None                        # the average was reset
    j = np.arange(N) - N // 2
    win = 0.5 + 0.5 * np.cos(2 * np.pi * j / N)
    avg = np.zeros(N)
    meter = 0.0
    d2 = 2700.0 * N / fs
    i0 = int(18650.0 * N / fs + 0.5)
    n = int(np.floor(d2) + 0.01)
    for b in range(5):
        X = np.fft.fft(x[b * N:(b + 1) * N] * win)
        avg += np.abs(np.fft.fftshift(X))
        meter += np.sum(np.abs(X[i0:i0 + n]) ** 2) + np.abs(X[i0 + n]) ** 2 * (d2 - n)
    want = 20 * np.log10(avg.reshape(W, N // W).sum(axis=1)) - 20 * (np.log10(5) + np.log10(N) + 31 * np.log10(2))
    want = np.clip(want, -200, 0)
    assert np.abs(pix - want).max() < 1e-9
    assert abs(sm - (10 * np.log10(meter / (2147483647.0 * N) ** 2 / 5) + 4.25969)) < 1e-9
    # the tone sits at +20 kHz: pixel (0.5 + 20000/fs) * W
    assert abs(int(np.argmax(pix)) - int((0.5 + 20000.0 / fs) * W)) <= 1
