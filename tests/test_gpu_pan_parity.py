"""Batched panadapter (C ABI group 5) against the get_graph restatement.  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(nch, n, fs, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((nch, n)) + 1j * rng.standard_normal((nch, n))) * 2.0 ** 20
    t = np.arange(n)
    for c in range(nch):
        x[c] += 2.0 ** 27 * np.exp(2j * np.pi * ((-30000.0 + 9000.0 * c) / fs) * t)
    return x


@pytest.mark.parametrize("fft_size,data_width,fs", [(1024, 1024, 48000.0), (2048, 512, 96000.0), (4096, 1024, 192000.0),
                                                     (8192, 1024, 192000.0), (16384, 1024, 1536000.0)])
def test_graph_matches_oracle(qh, oracle, fft_size, data_width, fs):
    """Ragged feeds (partial blocks carried between calls), several channels, S-meter band per channel."""
    nch = 3
    n = fft_size * 6 + 1234
    x = make(nch, n, fs, fft_size)
    p = qh.Panadapter(nch, fft_size, data_width, fs)
    refs = [oracle.OracleGraph(fft_size, data_width, fs) for _ in range(nch)]
    for c in range(nch):
        p.set_smeter_band(c, -31000.0 + 9000.0 * c, 2700.0)
        refs[c].set_smeter_band(-31000.0 + 9000.0 * c, 2700.0)
    assert p.get_graph() is None
    cuts = [0, 100, fft_size + 7, 3 * fft_size, 3 * fft_size + 1, n]
    for a, b in zip(cuts, cuts[1:]):
        p.feed_host(x[:, a:b])
        for c in range(nch):
            refs[c].feed(x[c, a:b])
    pix, sm, cnt = p.get_graph(1.0, 0.0)
    assert cnt == n // fft_size
    for c in range(nch):
        rp, rs, rc = refs[c].get(1.0, 0.0)
        assert rc == cnt
        assert np.abs(pix[c] - rp).max() < 1e-8           # dB; relative 1e-9 in amplitude
        assert abs(sm[c] - rs) < 1e-8
    assert p.get_graph() is None                            # reset like the reference


def test_zoom_and_offset_including_in_place_quirk(qh, oracle):
    """Zoomed in far enough that the reference's in-place pixel loop re-reads sums it has already written
    (quisk.c:5291-5300): the GPU path reproduces that order."""
    fft_size, data_width, fs = 4096, 1024, 192000.0
    x = make(1, fft_size * 3, fs, 5)
    for zoom, deltaf in ((0.5, 10000.0), (0.1, -20000.0), (0.05, 30000.0)):
        p = qh.Panadapter(1, fft_size, data_width, fs)
        r = oracle.OracleGraph(fft_size, data_width, fs)
        p.feed_host(x); r.feed(x[0])
        pix, _, _ = p.get_graph(zoom, deltaf)
        rp, _, _ = r.get(zoom, deltaf)
        assert np.abs(pix[0] - rp).max() < 1e-8


def test_config3_shape_pan_plus_fir(qh, oracle):
    """BASELINE config 3's per-channel work on the same raw stream: 16384-point panadapter + 1023-tap /32."""
    fs, nch = 1536000.0, 4
    n = 16384 * 4
    x = make(nch, n, fs, 9)
    k = np.arange(1023) - 511
    taps = np.sinc(k / 32.0) / 32.0 * np.blackman(1023)
    p = qh.Panadapter(nch, 16384, 1024, fs)
    bank = qh.FirBank(nch, taps, 32)
    p.feed_host(x)
    y = bank.process_host(x)
    pix, sm, cnt = p.get_graph()
    assert cnt == 4 and y.shape == (nch, n // 32)
    for c in (0, nch - 1):
        r = oracle.OracleGraph(16384, 1024, fs); r.feed(x[c])
        assert np.abs(pix[c] - r.get()[0]).max() < 1e-8
        want = oracle.OracleFir(taps).cDecimate(x[c], 32)
        assert np.sqrt(np.sum(np.abs(y[c] - want) ** 2) / np.sum(np.abs(want) ** 2)) < 1e-12


@pytest.mark.parametrize("fft_size,data_width,fs", [(4000, 1000, 192000.0), (2400, 1200, 48000.0), (9000, 1000, 960000.0), (672, 336, 48000.0),
                                                    (15360, 960, 192000.0), (1008 * 6, 1008, 96000.0)])
def test_quisk_sizes_that_are_not_powers_of_two(qh, oracle, fft_size, data_width, fs):
    """fft_size = data_width * fft_mult with data_width = 2^a y z (quisk.py:186-194, 4179): Bluestein on the power-of-two
    transforms; the oracle evaluates the DFT by its definition."""
    nch = 2
    n = fft_size * 3 + 321
    x = make(nch, n, fs, fft_size)
    p = qh.Panadapter(nch, fft_size, data_width, fs)
    refs = [oracle.OracleGraph(fft_size, data_width, fs) for _ in range(nch)]
    for c in range(nch):
        p.set_smeter_band(c, 9000.0 + 1000.0 * c, 2700.0)
        refs[c].set_smeter_band(9000.0 + 1000.0 * c, 2700.0)
    for a, b in ((0, 77), (77, fft_size + 5), (fft_size + 5, n)):
        p.feed_host(x[:, a:b])
        for c in range(nch):
            refs[c].feed(x[c, a:b])
    pix, sm, cnt = p.get_graph(1.0, 0.0)
    assert cnt == 3
    for c in range(nch):
        rp, rs, rc = refs[c].get(1.0, 0.0)
        assert rc == 3
        assert np.abs(pix[c] - rp).max() < 1e-8 and abs(sm[c] - rs) < 1e-8
    assert p.get_graph() is None
    with pytest.raises(qh.QuiskHipError):
        qh.Panadapter(1, 4001, 1000, fs)                    # "FFT size must be an even number", quisk.py:186


def test_block_ranges_that_come_out_empty(qh, oracle):
    """64 channels x 16384 points split a call's blocks into four ranges per channel; five blocks leave the fourth range empty
    (2 + 2 + 1 + 0).  Its partial sums must count as zero, not as what an earlier, longer call left in the buffer."""
    fs, nch, N = 1536000.0, 64, 16384
    x = make(nch, N * 8, fs, 21)
    p = qh.Panadapter(nch, N, 1024, fs)
    p.feed_host(x)                                  # eight blocks: every range of every channel holds sums
    assert p.get_graph()[2] == 8
    p.feed_host(x[:, :5 * N])
    pix, sm, cnt = p.get_graph()
    assert cnt == 5
    for c in (0, 17, 63):
        r = oracle.OracleGraph(N, 1024, fs); r.feed(x[c, :5 * N])
        rp, rs, _ = r.get()
        assert np.abs(pix[c] - rp).max() < 1e-8 and abs(sm[c] - rs) < 1e-8
