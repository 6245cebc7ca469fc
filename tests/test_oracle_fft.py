import numpy as np


def test_fft_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in (2, 8, 512, 4096, 16384):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        f = oracle.fft(x, -1)
        assert np.abs(f - np.fft.fft(x)).max() < 1e-11 * np.sqrt(n) * 4
        b = oracle.fft(x, +1)
        assert np.abs(b - np.fft.ifft(x) * n).max() < 1e-11 * np.sqrt(n) * 4
