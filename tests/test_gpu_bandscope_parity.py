"""qh_bscope_* (get_bandscope + copy2pixels, quisk.c:4932-5011) against the C restatement.
fp64: |dB difference| <= 1e-8 wherever the pixel is above the -200 floor.  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,W,zoom,deltaf", [(16384, 1000, 1.0, 0.0), (4096, 777, 0.25, 3.0e6), (1024, 300, 0.5, -7.0e6), (8192, 1200, 1.0, 0.0),
                                             (24 * 512, 1000, 1.0, 0.0), (3000, 500, 0.5, 1.0e6)])
def test_bandscope_matches_oracle(qh, oracle, N, W, zoom, deltaf):
    clock, nch, nblk = 122880000, 3, 5
    rng = np.random.default_rng(N)
    t = np.arange(N * nblk)
    x = np.stack([rng.uniform(-0.01, 0.01, t.size) + 0.4 * np.cos(2 * np.pi * (0.11 + 0.07 * c) * t) + 0.2 * np.cos(2 * np.pi * 0.31 * t + c)
                  for c in range(nch)])
    b = qh.Bandscope(nch, N, W)
    assert b.get_bandscope(clock) is None
    b.feed_host(x[:, :N + 100])                                    # ragged feeds: the block fill carries over
    b.feed_host(x[:, N + 100:])
    assert b.count() == nblk
    pix, adc, cnt = b.get_bandscope(clock, zoom, deltaf)
    assert cnt == nblk and b.get_bandscope(clock) is None
    for c in range(nch):
        o = oracle.OracleBandscope(N, W)
        for k in range(nblk):
            o.block(x[c, k * N:(k + 1) * N])
        rp, radc, rc = o.get(clock, zoom, deltaf)
        assert rc == nblk and adc[c] == radc
        assert np.abs(pix[c] - rp).max() < 1e-8
        assert rp.max() > -60.0
    # the average starts over
    b.feed_host(x[:, :N])
    pix2, adc2, cnt2 = b.get_bandscope(clock, zoom, deltaf)
    o = oracle.OracleBandscope(N, W)
    o.block(x[0, :N])
    rp, radc, rc = o.get(clock, zoom, deltaf)
    assert cnt2 == 1 and adc2[0] == radc and np.abs(pix2[0] - rp).max() < 1e-8


def test_silence_hits_the_floor_and_bad_sizes_fail(qh):
    b = qh.Bandscope(1, 2048, 100)
    b.feed_host(np.zeros((1, 2048)))
    pix, adc, cnt = b.get_bandscope(122880000)
    assert cnt == 1 and adc[0] == 0.0 and np.all(pix == -200.0)
    with pytest.raises(qh.QuiskHipError):
        qh.Bandscope(1, 3001, 100)                          # odd sizes have no Nyquist bin to put at N / 2
