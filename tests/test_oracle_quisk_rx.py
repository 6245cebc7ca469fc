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


def test_rates_that_do_not_plan_are_refused(oracle):
    import pytest
    with pytest.raises(ValueError):
        oracle.OracleQuiskRx(250000, rxfilter.coefficient_tables())
