"""The waterfall row (watfall_OnGraphData, quisk.c:5372-5421) on the GPU against the C
restatement: byte-exact for the mapping itself; through qh_pan_waterfall the dB row comes from the panadapter, whose
fp64 tolerance (1e-8 dB) can move a value across a colour-index boundary, so there the gate is "at most one palette
step, in at most a handful of pixels".  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _palette(seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, 256, dtype=np.uint8) for _ in range(3)]


@pytest.mark.parametrize("ncols,width,y_zero,y_scale,gain", [(1000, 1000, 0, 0, 0.0), (800, 1024, 37, 55, -12.5), (1200, 700, 100, 100, 30.0),
                                                             (0, 16, 5, 5, 0.0)])
def test_rows_are_byte_exact(qh, oracle, ncols, width, y_zero, y_scale, gain):
    from quisk_amd import pan
    rng = np.random.default_rng(ncols + width)
    db = rng.uniform(-200.0, 0.0, size=(5, ncols))
    if ncols:
        db[0, :7] = [-200.0, 0.0, -40.0 - 0.69 * y_zero + gain, 1e6, -1e6, -127.99999, 12.7]     # (beyond +-2^31 colour steps the C cast is undefined)
    red, green, blue = _palette(width)
    got = pan.waterfall_rows(db, red, green, blue, y_zero, y_scale, gain, width)
    want = np.stack([oracle.watfall_row(db[r], red, green, blue, y_zero, y_scale, gain, width) for r in range(5)])
    assert got.dtype == np.uint8 and got.shape == (5, width, 3)
    assert np.array_equal(got, want)
    if ncols < width:
        assert not np.any(got[:, ncols:, :])


def test_panadapter_waterfall_row(qh, oracle):
    fs, N, W, nch = 96000, 4096, 800, 3
    rng = np.random.default_rng(2)
    t = np.arange(N * 6)
    x = np.stack([2.0 ** 20 * np.exp(2j * np.pi * ((5000.0 + 3000 * c) / fs) * t) +
                  2.0 ** 12 * (rng.standard_normal(t.size) + 1j * rng.standard_normal(t.size)) for c in range(nch)])
    red = np.arange(256, dtype=np.uint8); green = (255 - red).astype(np.uint8); blue = ((red.astype(int) * 7) % 256).astype(np.uint8)
    p = qh.Panadapter(nch, N, W, float(fs))
    p.feed_host(x)
    rgb, sm, cnt = p.waterfall_row(red, green, blue, 20, 30, -5.0, 900)
    assert cnt == 6 and rgb.shape == (nch, 900, 3) and not np.any(rgb[:, W:, :])
    assert p.waterfall_row(red, green, blue, 20, 30, -5.0, 900) is None
    for c in range(nch):
        g = oracle.OracleGraph(N, W, float(fs))
        g.feed(x[c])
        pix, rs, rc = g.get(1.0, 0.0)
        want = oracle.watfall_row(pix, red, green, blue, 20, 30, -5.0, 900)
        step = np.abs(rgb[c, :, 0].astype(int) - want[:, 0].astype(int))         # red = the colour index itself
        assert step.max() <= 1 and np.count_nonzero(step) <= 4
        same = step == 0
        assert np.array_equal(rgb[c][same], want[same])
        assert abs(sm[c] - rs) < 1e-8
