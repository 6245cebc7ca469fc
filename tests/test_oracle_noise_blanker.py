"""oracle/quisk_rx_oracle.c qo_nb_* (NoiseBlanker, quisk.c:680-784) against what the algorithm must do by construction.
PARITY UNPINNED by reference execution (quisk.c needs <fftw3.h>); these pin the restatement's behaviour instead."""
import numpy as np

from quisk_amd import rxfilter, synth


def test_quiet_signal_is_only_delayed(oracle):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)) * 1e5
    for rate, level in ((48000, 1), (192000, 2)):
        nb = oracle.OracleNoiseBlanker(rate, level)
        assert nb.delay == 3 * int(rate * 500e-6 + 0.5)
        y = nb.process(x)
        assert not np.any(y[:nb.delay])
        # the first samples ARE pulses against the empty window's mean (quisk.c:744: mag <= save_sum / save_size * limit);
        # once the window has filled, Rayleigh noise never reaches 4x its mean and the blanker is a pure delay
        w = 2 * nb.delay
        assert np.array_equal(y[nb.delay + w:], x[w:x.size - nb.delay])
        assert y[nb.delay] == 0


def test_level_zero_is_a_passthrough(oracle):
    x = synth.impulsive_input(1, 4000)[0]
    nb = oracle.OracleNoiseBlanker(48000, 0)
    assert np.array_equal(nb.process(x), x)


def test_single_pulse_taper_zero_and_ramp(oracle):
    rate = 48000
    hw, S = 24, 72
    x = np.full(1000, 1000.0 + 0j)
    x[500] = 1e6
    y = oracle.OracleNoiseBlanker(rate, 1).process(x)
    g = (y[S:] / x[:x.size - S]).real                     # gain per input sample
    assert np.all(g[200:500 - hw + 1] == 1.0)           # (before 200: the start-up transient, see the first test)
    # quisk.c:751-756: the hw samples up to the pulse are multiplied by j / hw, j = 0 at the pulse
    assert np.array_equal(g[500 - hw + 1:501], (np.arange(hw - 1, -1, -1) / hw))
    assert g[501] == 0.0                                    # state 1: zero until the pulses stop
    # quisk.c:758-762: win_index / hw for win_index = 1 .. hw - 1
    assert np.array_equal(g[502:502 + hw - 1], np.arange(1, hw) / hw)
    assert np.all(g[502 + hw - 1:] == 1.0)


def test_block_boundaries_do_not_matter(oracle):
    x = synth.impulsive_input(1, 30000)[0]
    a = oracle.OracleNoiseBlanker(192000, 3).process(x)
    nb = oracle.OracleNoiseBlanker(192000, 3)
    b = np.concatenate([nb.process(x[:1]), nb.process(x[1:777]), nb.process(x[777:20000]), nb.process(x[20000:])])
    assert np.array_equal(a, b)
    assert np.count_nonzero(a == 0) > nb.delay + 50         # pulses were found and blanked


def test_receiver_oracle_runs_the_blanker_before_the_tune(oracle):
    t = rxfilter.coefficient_tables()
    x = synth.impulsive_input(1, 96000, scale=1e6)[0]
    nb = oracle.OracleNoiseBlanker(96000, 2)
    r0, r1 = oracle.OracleQuiskRx(96000, t), oracle.OracleQuiskRx(96000, t)
    r1.set_noise_blanker(2)
    for r in (r0, r1):
        r.set_mode(rxfilter.USB)
        r.set_tune(5000)
        r.set_filters(*rxfilter.make_filter_coef(12000, None, 2700, rxfilter.get_filter_center("USB", 2700)))
    y0 = np.concatenate([r0.process(nb.process(x[k:k + 24000])) for k in range(0, 96000, 24000)])
    y1 = np.concatenate([r1.process(x[k:k + 24000]) for k in range(0, 96000, 24000)])
    assert y0.size == 48000 and np.array_equal(y0, y1)
