"""Known answers for the Quisk-native receive restatement (SURVEY.md appendix A probes of the real reference)."""
import numpy as np

from quisk_amd import rxfilter


def test_usb_tone_and_rate(oracle):
    t = rxfilter.coefficient_tables()
    r = oracle.OracleQuiskRx(192000, t)
    r.set_mode(rxfilter.USB)
    r.set_tune(10000)
    r.set_filters(*rxfilter.make_filter_coef(12000, None, 2700, rxfilter.get_filter_center("USB", 2700)))
    n = 192000
    y = r.process(2.0 ** 24 * np.exp(2j * np.pi * 11000.0 / 192000 * np.arange(n)))
    assert y.size == n // 4                                     # 4:1 rate change
    assert np.array_equal(y.real, y.imag)                       # I == Q "stereo"
    Y = np.fft.rfft(y[-8192:].real * np.hanning(8192))
    assert abs(np.argmax(np.abs(Y)) * 48000 / 8192 - 1000.0) < 6.0   # RF tone at tune + 1000 Hz -> 1 kHz audio


def test_rational_stage_rate_250k(oracle):
    """250 k -> /5 = 50 k -> x6/5 x4/5 = 48 k (quisk.c:1834-1838): 250000 in, 48000 out, tone preserved."""
    t = rxfilter.coefficient_tables()
    r = oracle.OracleQuiskRx(250000, t)
    r.set_mode(rxfilter.USB)
    r.set_tune(20000)
    r.set_filters(*rxfilter.make_filter_coef(12000, None, 2700, rxfilter.get_filter_center("USB", 2700)))
    x = 2.0 ** 24 * np.exp(2j * np.pi * 21000.0 / 250000 * np.arange(250000))
    # blocks as the sound thread delivers them: one call may not exceed SAMP_BUFFER_SIZE * 0.8 outputs (filter.c:315)
    y = np.concatenate([r.process(x[k:k + 25000]) for k in range(0, 250000, 25000)])
    assert r.decim_srate() == 48000 and r.filter_srate() == 12000
    assert abs(y.size - 48000) <= 1
    Y = np.fft.rfft(y[-8192:].real * np.hanning(8192))
    assert abs(np.argmax(np.abs(Y)) * 48000 / 8192 - 1000.0) < 6.0


def test_sdriq_rate_and_dgt_modes(oracle):
    t = rxfilter.coefficient_tables()
    r = oracle.OracleQuiskRx(111111, t)                         # SDR-IQ: /2 only, audio at 55555 sps (quisk.c:1740-1743)
    r.set_mode(rxfilter.USB)
    y = np.concatenate([r.process(np.ones(11104, dtype=complex)) for _ in range(10)])
    assert r.decim_srate() == 55555 and y.size == 111040 // 2
    r = oracle.OracleQuiskRx(48000, t)
    r.set_mode(9)                                               # DGT-IQ, wide: samples pass through untouched
    r.set_bandwidth(20000)
    x = np.random.default_rng(0).standard_normal(500) + 1j * np.random.default_rng(1).standard_normal(500)
    assert np.array_equal(r.process(x), x)
    r = oracle.OracleQuiskRx(48000, t)
    r.set_mode(7)                                               # DGT-U narrow filters at 6 ksps, wide at 48 ksps
    r.set_bandwidth(500)
    r.process(x)
    assert r.filter_srate() == 6000
    r.set_bandwidth(3000)
    r.process(x)
    assert r.filter_srate() == 48000
