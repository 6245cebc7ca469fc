"""Quisk's audio AGC (process_agc, quisk.c:2162-2287) on the GPU against its restatement.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import rxfilter

pytestmark = pytest.mark.gpu


def audio(seed, n):
    """speech-like level changes: quiet, loud burst (overload), quiet, a second louder burst inside a ramp"""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    env = np.full(n, 2.0 ** 18)
    env[n // 5: n // 5 + 3000] = 2.0 ** 29
    env[n // 2: n // 2 + 400] = 2.0 ** 27
    env[n // 2 + 150: n // 2 + 250] = 2.0 ** 30
    env[3 * n // 4:] = 2.0 ** 22
    x = env * np.sin(2 * np.pi * (700.0 + 13 * seed) / 48000 * t) + 2.0 ** 12 * rng.standard_normal(n)
    return x + 1j * x


@pytest.mark.parametrize("is_cpx", [False, True])
def test_agc_alone_on_identical_input(qh, oracle, is_cpx):
    nch, n = 4, 48000
    x = np.stack([audio(c, n) for c in range(nch)])
    if is_cpx:
        x = x.real + 1j * np.roll(x.real, 7, axis=1)
    cuts = [0, 100, 101, 5000, 5064, 20000, n]
    g = qh.QuiskAgc(nch, 48000, is_cpx=is_cpx)
    for c in range(nch):
        g.set_agc(c, 80.0 + 10 * c)
    y = np.concatenate([g.process_host(x[:, a:b]) for a, b in zip(cuts, cuts[1:])], axis=1)
    assert np.array_equal(y[:, :100], x[:, :100])               # the first call only initialises (quisk.c:2173-2190)
    for c in range(nch):
        o = oracle.OracleQuiskAgc(48000)
        want = np.concatenate([o.process(x[c, a:b], is_cpx, 80.0 + 10 * c) for a, b in zip(cuts, cuts[1:])])
        level = np.abs(want[101:]) if is_cpx else np.abs(want[101:].real)
        assert level.max() <= 0.7 * 2.0 ** 31 * 1.0001                     # the limiter holds
        assert rel_rms(y[c], want) < 1e-12, (c, rel_rms(y[c], want))


def test_receiver_bank_with_agc(qh, oracle):
    """USB at 192 k with process_agc on the output, like quisk_process_samples; blocks as the sound thread sends them."""
    fs, nch = 192000, 2
    tabs = rxfilter.coefficient_tables()
    n = fs
    t = np.arange(n)
    bank = qh.QuiskRxBank(nch, fs, 3)
    bank.set_agc(True, 80.0)
    fI, fQ = rxfilter.make_filter_coef(bank.get_filter_rate(), None, 2700, rxfilter.get_filter_center("USB", 2700))
    x, refs = [], []
    for c in range(nch):
        amp = np.where((t > n // 3) & (t < n // 3 + 20000), 2.0 ** 27, 2.0 ** 18)
        x.append(amp * np.exp(2j * np.pi * ((10000.0 + 900.0 + 50 * c) / fs * t % 1.0)))
        bank.set_tune(c, 10000); bank.set_filters(c, fI, fQ)
        r = oracle.OracleQuiskRx(fs, tabs)
        r.set_mode(3); r.set_tune(10000); r.set_filters(fI, fQ); r.set_agc(True, 80.0)
        refs.append(r)
    x = np.stack(x)
    blk = 4096
    y = np.concatenate([bank.process_host(x[:, k:k + blk]) for k in range(0, n, blk)], axis=1)
    for c in range(nch):
        want = np.concatenate([refs[c].process(x[c, k:k + blk]) for k in range(0, n, blk)])
        assert y.shape[1] == want.size
        assert np.abs(want).max() > 2.0 ** 28 and np.abs(want.real).max() <= 0.7 * 2.0 ** 31 * 1.0001
        assert rel_rms(y[c], want) < 1e-8, (c, rel_rms(y[c], want))
