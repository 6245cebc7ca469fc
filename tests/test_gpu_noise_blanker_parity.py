"""qh_nb_* (Quisk's NoiseBlanker, quisk.c:680-784) against oracle/quisk_rx_oracle.c.
The GPU evaluates the blanker as a sliding-window function of the input (pulse bits from a prefix sum of |x|, gains from
the bits) instead of stepping the reference's state machine; the multiplications are the reference's, in its order, so
the gate is bit-exact output wherever the pulse verdicts agree -- and they must agree on these inputs.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import rxfilter, synth

pytestmark = pytest.mark.gpu


def _oracle_run(oracle, rate, level, x, cuts):
    outs = []
    for c in range(x.shape[0]):
        nb = oracle.OracleNoiseBlanker(rate, level)
        outs.append(np.concatenate([nb.process(x[c, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]))
    return np.stack(outs)


@pytest.mark.parametrize("rate,level", [(48000, 1), (96000, 2), (192000, 3), (192000, 1), (1536000, 2)])
def test_matches_oracle_bit_exact(qh, oracle, rate, level):
    n = 40000 if rate < 1000000 else 120000
    x = synth.impulsive_input(3, n, seed=rate + level)
    cuts = [0, 1, 300, 5000, 5001, 23456, n]                 # ragged calls, one of them a single sample
    ref = _oracle_run(oracle, rate, level, x, cuts)
    nb = qh.NoiseBlanker(3, rate, level)
    assert nb.delay == 3 * int(rate * 500e-6 + 0.5)
    y = np.concatenate([nb.process_host(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])], axis=1)
    assert np.count_nonzero(ref == 0) > 3 * (nb.delay + 20)     # pulses were found
    # -0.0 where the reference multiplies by 0 / hw, +0.0 where it stores 0: equal as numbers
    assert np.array_equal(y, ref)


def test_many_tiles_and_channels(qh, oracle):
    # several tiles per channel (tile = 4096 outputs at 192 k), pulses straddling tile edges
    rate, n, nch = 192000, 70000, 5
    x = synth.impulsive_input(nch, n, seed=99)
    for c in range(nch):
        for edge in (4096, 8192, 8192 + 288, 40960 - 96):
            x[c, edge - 1 - c:edge + 1] += 9e7
    ref = _oracle_run(oracle, rate, 2, x, [0, n])
    y = qh.NoiseBlanker(nch, rate, 2).process_host(x)
    assert np.array_equal(y, ref)


def test_level_changes_and_off_periods_follow_the_reference(qh, oracle):
    rate = 192000
    x = synth.impulsive_input(1, 60000, seed=5)
    nb, o = qh.NoiseBlanker(1, rate, 1), oracle.OracleNoiseBlanker(rate, 1)
    ys, rs = [], []
    plan = [(0, 9000, 1), (9000, 9100, 3), (9100, 20000, 3), (20000, 26000, 0), (26000, 26050, 2), (26050, 60000, 2)]
    for a, b, level in plan:
        nb.set_level(level); o.set_level(level)
        ys.append(nb.process_host(x[:, a:b])[0]); rs.append(o.process(x[0, a:b]))
    y, r = np.concatenate(ys), np.concatenate(rs)
    assert np.array_equal(y[20000:26000], x[0, 20000:26000])        # off: undelayed passthrough
    assert np.array_equal(y, r)


def test_quiet_input_is_a_pure_delay_and_reset_clears_it(qh):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, 20000)) + 1j * rng.standard_normal((2, 20000))) * 1e5
    nb = qh.NoiseBlanker(2, 192000, 2)
    y = nb.process_host(x)
    d = nb.delay
    assert np.array_equal(y[:, 3 * d:], x[:, 2 * d:-d])
    nb.reset()
    assert np.array_equal(nb.process_host(x), y)


def test_receiver_bank_runs_the_blanker_before_the_tune(qh, oracle):
    t = rxfilter.coefficient_tables()
    x = synth.impulsive_input(2, 96000, seed=11)
    fI, fQ = rxfilter.make_filter_coef(12000, None, 2700, rxfilter.get_filter_center("USB", 2700))
    bank = qh.QuiskRxBank(2, 96000, rxfilter.USB)
    bank.set_noise_blanker(2)
    ys, rs = [], [[], []]
    refs = [oracle.OracleQuiskRx(96000, t) for _ in range(2)]
    for c, r in enumerate(refs):
        r.set_mode(rxfilter.USB); r.set_tune(5000 + 100 * c); r.set_filters(fI, fQ); r.set_noise_blanker(2)
        bank.set_tune(c, 5000 + 100 * c)
    bank.set_filters(-1, fI, fQ)
    for k in range(0, 96000, 24000):
        ys.append(bank.process_host(x[:, k:k + 24000]))
        for c, r in enumerate(refs):
            rs[c].append(r.process(x[c, k:k + 24000]))
    y = np.concatenate(ys, axis=1)
    ref = np.stack([np.concatenate(r) for r in rs])
    assert y.shape == ref.shape
    assert rel_rms(y, ref) < 1e-9          # tolerance of the receiver chain (overlap-save FIRs), test_gpu_quisk_rx_parity.py


def test_block_api_setter_blanks_the_panadapter_input_too(qh, oracle):
    fs, blk = 96000, 12000
    api = qh.quiskapi
    api.open(fs, fft_size=2048, data_width=512)
    t = rxfilter.coefficient_tables()
    x = synth.impulsive_input(1, 48000, seed=21, scale=2.0 ** 20)[0]
    fI, fQ = rxfilter.make_filter_coef(12000, None, 2700, rxfilter.get_filter_center("USB", 2700))
    api.set_rx_mode(rxfilter.USB); api.set_tune(3000); api.set_filters(fI, fQ, 2700); api.set_noise_blanker(1)
    r = oracle.OracleQuiskRx(fs, t)
    r.set_mode(rxfilter.USB); r.set_tune(3000); r.set_filters(fI, fQ); r.set_noise_blanker(1); r.set_agc(True, 80.0)
    nb = oracle.OracleNoiseBlanker(fs, 1)
    g = oracle.OracleGraph(2048, 512, float(fs))
    outs, refs = [], []
    for k in range(0, 48000, blk):
        buf = x[k:k + blk].copy()
        n = api.process_samples(buf, blk)
        outs.append(buf[:n].copy())
        refs.append(r.process(x[k:k + blk]))
        g.feed(nb.process(x[k:k + blk]))                    # the FFT ring is filled after NoiseBlanker, quisk.c:2449-2475
    y, ref = np.concatenate(outs), np.concatenate(refs)
    pix, sm, cnt = api.get_graph(1.0, 0.0)
    rp, rs, rc = g.get(1.0, 0.0)
    api.set_noise_blanker(0)
    api.close()
    assert y.shape == ref.shape and rel_rms(y, ref) < 1e-8
    assert cnt == rc and np.abs(pix - rp).max() < 1e-8 and abs(sm - rs) < 1e-8
