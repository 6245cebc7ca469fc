"""qh_qrx_set_auto_notch (dAutoNotch, quisk.c:786-963) inside the receiver bank against the
restatement inside the oracle receiver: the notch must pick the same bins at the same blocks (its decisions are
arg-max / hysteresis, not arithmetic), after which the gate is the receiver chain's fp64 tolerance.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import rxfilter

pytestmark = pytest.mark.gpu

NAMES = {rxfilter.CWU: "CWU", rxfilter.USB: "USB", rxfilter.LSB: "LSB", rxfilter.AM: "AM", rxfilter.FM: "FM"}


def _rf(fs, n, tune, seed, carriers):
    t = np.arange(n)
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 2.0 ** 14
    for f, a in carriers:
        x = x + a * np.exp(2j * np.pi * (((tune + f) / fs) * t % 1.0))
    return x


@pytest.mark.parametrize("mode,bw,carriers,rit", [
    (rxfilter.USB, 2700, [(1000.0, 2.0 ** 22), (2100.0, 2.0 ** 21)], 0),
    (rxfilter.LSB, 2700, [(-800.0, 2.0 ** 22)], 0),
    (rxfilter.CWU, 500, [(700.0, 2.0 ** 22), (900.0, 2.0 ** 21)], 700),
    (rxfilter.AM, 6000, [(0.0, 2.0 ** 23), (1500.0, 2.0 ** 21)], 0),
])
def test_bank_with_auto_notch_matches_the_oracle_receiver(qh, oracle, mode, bw, carriers, rit):
    fs, blk, nblk, tune = 96000, 12000, 40, 7000
    tabs = rxfilter.coefficient_tables()
    frate = rxfilter.get_filter_rate(fs, mode, bw)
    fI, fQ = rxfilter.make_filter_coef(frate, None, bw, rxfilter.get_filter_center(NAMES[mode], bw))
    bank = qh.QuiskRxBank(2, fs, mode, bandwidth=bw)
    bank.set_filters(-1, fI, fQ)
    refs = []
    xs = []
    for c in range(2):
        bank.set_tune(c, tune + 50 * c)
        r = oracle.OracleQuiskRx(fs, tabs)
        r.set_mode(mode); r.set_tune(tune + 50 * c); r.set_filters(fI, fQ); r.set_bandwidth(bw)
        r.set_auto_notch(True, rit)
        refs.append(r)
        xs.append(_rf(fs, blk * nblk, tune + 50 * c, 30 + c, carriers))
    bank.set_auto_notch(True, rit)
    x = np.stack(xs)
    ys, rs = [], [[], []]
    for k in range(0, blk * nblk, blk):
        ys.append(bank.process_host(x[:, k:k + blk]))
        for c in range(2):
            rs[c].append(refs[c].process(x[c, k:k + blk]))
    y = np.concatenate(ys, axis=1)
    ref = np.stack([np.concatenate(r) for r in rs])
    assert y.shape == ref.shape
    # the notch engaged: late audio is well below what the receiver gives without it
    plain = oracle.OracleQuiskRx(fs, tabs)
    plain.set_mode(mode); plain.set_tune(tune); plain.set_filters(fI, fQ); plain.set_bandwidth(bw)
    yp = np.concatenate([plain.process(x[0, k:k + blk]) for k in range(0, blk * nblk, blk)])
    tail = slice(-48000, None)
    if mode != rxfilter.CWU:
        assert np.std(ref[0, tail].real) < 0.3 * np.std(yp[tail].real)
    assert rel_rms(y, ref) < 1e-8


def test_fm_notch_runs_after_the_interpolators_on_both_parts(qh, oracle):
    fs, blk, nblk, tune = 96000, 12000, 24, 5000
    tabs = rxfilter.coefficient_tables()
    fI, fQ = rxfilter.make_filter_coef(48000, None, 12000, 0)
    t = np.arange(blk * nblk)
    # FM carrier with a steady 1 kHz audio tone: the notch takes the tone out of the audio
    x = 2.0 ** 24 * np.exp(1j * (2 * np.pi * ((tune / fs) * t % 1.0) + 1.5 * np.sin(2 * np.pi * 1000.0 * t / fs)))
    x = x + (np.random.default_rng(4).standard_normal(t.size) + 0j) * 2.0 ** 12
    bank = qh.QuiskRxBank(1, fs, rxfilter.FM, bandwidth=12000)
    bank.set_filters(-1, fI, fQ); bank.set_tune(0, tune); bank.set_auto_notch(True)
    r = oracle.OracleQuiskRx(fs, tabs)
    r.set_mode(rxfilter.FM); r.set_tune(tune); r.set_filters(fI, fQ); r.set_bandwidth(12000); r.set_auto_notch(True)
    y = np.concatenate([bank.process_host(x[None, k:k + blk])[0] for k in range(0, t.size, blk)])
    ref = np.concatenate([r.process(x[k:k + blk]) for k in range(0, t.size, blk)])
    assert y.shape == ref.shape and np.array_equal(y.real, y.imag)
    assert rel_rms(y, ref) < 1e-7


def test_set_auto_notch_restarts_and_switches_off(qh, oracle):
    fs, blk, tune = 48000, 6000, 3000
    tabs = rxfilter.coefficient_tables()
    fI, fQ = rxfilter.make_filter_coef(12000, None, 2700, rxfilter.get_filter_center("USB", 2700))
    x = _rf(fs, blk * 30, tune, 8, [(1200.0, 2.0 ** 22)])
    bank = qh.QuiskRxBank(1, fs, rxfilter.USB)
    bank.set_filters(-1, fI, fQ); bank.set_tune(0, tune)
    r = oracle.OracleQuiskRx(fs, tabs)
    r.set_mode(rxfilter.USB); r.set_tune(tune); r.set_filters(fI, fQ)
    ys, rs = [], []
    for b, k in enumerate(range(0, x.size, blk)):
        if b == 3: bank.set_auto_notch(True); r.set_auto_notch(True)
        if b == 14: bank.set_auto_notch(True); r.set_auto_notch(True)       # again: the notch starts over
        if b == 22: bank.set_auto_notch(False); r.set_auto_notch(False)
        ys.append(bank.process_host(x[None, k:k + blk])[0]); rs.append(r.process(x[k:k + blk]))
    assert rel_rms(np.concatenate(ys), np.concatenate(rs)) < 1e-8
    with pytest.raises(qh.QuiskHipError):
        qh.QuiskRxBank(1, fs, rxfilter.DGT_IQ).set_auto_notch(True)


def test_block_api_setter(qh, oracle):
    fs, blk, tune = 96000, 12000, 4000
    api = qh.quiskapi
    api.open(fs)
    tabs = rxfilter.coefficient_tables()
    fI, fQ = rxfilter.make_filter_coef(12000, None, 2700, rxfilter.get_filter_center("USB", 2700))
    api.set_rx_mode(rxfilter.USB); api.set_tune(tune); api.set_filters(fI, fQ, 2700); api.set_auto_notch(1)
    r = oracle.OracleQuiskRx(fs, tabs)
    r.set_mode(rxfilter.USB); r.set_tune(tune); r.set_filters(fI, fQ); r.set_agc(True, 80.0); r.set_auto_notch(True)
    x = _rf(fs, blk * 24, tune, 12, [(1300.0, 2.0 ** 22)])
    outs, refs = [], []
    for k in range(0, x.size, blk):
        buf = x[k:k + blk].copy()
        n = api.process_samples(buf, blk)
        outs.append(buf[:n].copy()); refs.append(r.process(x[k:k + blk]))
    api.set_auto_notch(0)
    api.close()
    assert rel_rms(np.concatenate(outs), np.concatenate(refs)) < 1e-8
