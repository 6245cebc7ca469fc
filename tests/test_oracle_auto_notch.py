"""oracle/quisk_rx_oracle.c qo_notch_* (dAutoNotch, quisk.c:786-963) against what the algorithm must do.
PARITY UNPINNED by reference execution (quisk.c needs <fftw3.h>); these pin the restatement's behaviour."""
import numpy as np


def _tone_db(sig, f, rate):
    S = np.abs(np.fft.rfft(sig[-8192:] * np.hanning(8192)))
    k = int(round(f * 8192 / rate))
    return 20 * np.log10(S[k - 3:k + 4].max())


def _signal(rate, n, tones):
    t = np.arange(n)
    x = np.random.default_rng(0).standard_normal(n) * 1e6
    for f, a, ph in tones:
        x = x + a * np.cos(2 * np.pi * f * t / rate + ph)
    return x


def test_two_carriers_are_notched_and_the_rest_passes(oracle):
    rate, n = 12000, 12000 * 4
    x = _signal(rate, n, [(1000.0, 3e7, 0.0), (2100.0, 2e7, 1.0)])
    o = oracle.OracleAutoNotch(True)
    y = np.concatenate([o.process(x[k:k + 1000], 0, rate) for k in range(0, n, 1000)])
    assert _tone_db(x, 1000, rate) - _tone_db(y, 1000, rate) > 50
    assert _tone_db(x, 2100, rate) - _tone_db(y, 2100, rate) > 50
    assert abs(_tone_db(x, 500, rate) - _tone_db(y, 500, rate)) < 3          # noise floor untouched ("empirical" gain 2048 / 102 / 16)


def test_cw_sidetone_is_left_alone(oracle):
    rate, n = 6000, 6000 * 6
    x = _signal(rate, n, [(700.0, 3e7, 0.0), (1500.0, 2e7, 1.0)])
    o = oracle.OracleAutoNotch(True)
    y = np.concatenate([o.process(x[k:k + 750], 700, rate) for k in range(0, n, 750)])
    assert abs(_tone_db(x, 700, rate) - _tone_db(y, 700, rate)) < 3
    assert _tone_db(x, 1500, rate) - _tone_db(y, 1500, rate) > 40


def test_off_is_a_passthrough_and_first_block_is_the_zero_delay_line(oracle):
    x = _signal(12000, 4000, [(1000.0, 3e7, 0.0)])
    assert np.array_equal(oracle.OracleAutoNotch(False).process(x, 0, 12000), x)
    y = oracle.OracleAutoNotch(True).process(x, 0, 12000)
    assert not np.any(y[:1538])                      # data_out starts at zero: the first NOTCH_DATA_OUTPUT_SIZE outputs
    assert np.any(y[1538:])


def test_block_boundaries_do_not_matter(oracle):
    x = _signal(24000, 30000, [(3000.0, 3e7, 0.3)])
    a = oracle.OracleAutoNotch(True).process(x, 0, 24000)
    o = oracle.OracleAutoNotch(True)
    b = np.concatenate([o.process(x[:1], 0, 24000), o.process(x[1:1538], 0, 24000), o.process(x[1538:20000], 0, 24000),
                        o.process(x[20000:], 0, 24000)])
    assert np.array_equal(a, b)
