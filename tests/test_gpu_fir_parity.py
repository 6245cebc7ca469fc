"""The batched FIR decimator bank (C ABI group 3) against the filter.c restatement -- which is itself
pinned bit-exactly to the reference's own filter.c -- and against the golden vectors generated from the
reference build.  -m gpu."""
import os

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "filter_golden.npz")
SPLITS = [0, 1, 7, 333, 2, 64, 1000, 5, 0, 588]
TOL64, TOL32 = 1e-12, 2e-5          # north_star: <= 1e-6 (float64) / <= 1e-3 (float32) relative RMS


def stream(seed, n):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(n) + 1j * rng.standard_normal(n)


def run_split(bank, x):
    out, pos = [], 0
    for k in SPLITS:
        out.append(bank.process_host(x[:, pos:pos + k]))
        pos += k
    return np.concatenate(out, axis=1)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("name,tapkey,d", [("cDecimate_98_d2", "taps98", 2), ("cDecimate_147_d3", "taps147", 3),
                                            ("cDecimate_245_d5", "taps245", 5), ("cDecimate_98_d1", "taps98", 1)])
def test_golden_cdecimate_ragged_calls(qh, gold, name, tapkey, d):
    """Same ragged call pattern (0, 1, 7, 333, ... samples) as the vectors made from the reference build."""
    x = stream(11, sum(SPLITS))[None, :]
    y = run_split(qh.FirBank(1, gold[tapkey], d), x)
    assert y.shape[1] == gold[name].size
    assert rel_rms(y[0], gold[name]) < TOL64


def test_golden_hb45(qh, gold):
    x = stream(11, sum(SPLITS))[None, :]
    y = run_split(qh.FirBank(1, qh.hb45_taps(), 2), x)
    assert y.shape[1] == gold["cDecim2HB45"].size
    assert rel_rms(y[0], gold["cDecim2HB45"]) < TOL64


def test_golden_config5_cascade_f64_and_f32(qh, gold):
    """BASELINE config 5's front end: 8 x HB45 then the 245-tap /5, as a cascade of banks."""
    x = stream(13, 256 * 5 * 40)[None, :]
    for dtype, tol in ((0, TOL64), (1, TOL32)):
        y = x
        for _ in range(8):
            y = qh.FirBank(1, qh.hb45_taps(), 2, dtype=dtype).process_host(y)
        y = qh.FirBank(1, gold["taps245"], 5, dtype=dtype).process_host(y)
        assert y.shape[1] == gold["cascade_8hb45_d5"].size
        assert rel_rms(y[0], gold["cascade_8hb45_d5"]) < tol


def test_complex_taps_against_oracle(qh, oracle, gold):
    """quisk_cCDecimate with taps tuned by quisk_filt_tune (both sidebands)."""
    x = stream(21, 6000)
    for freq, upper, d in ((0.0625, 1, 5), (-0.11, 0, 2)):
        f = oracle.OracleFir(gold["taps245"])
        f.tune(freq, upper)
        want = f.cCDecimate(x, d)
        ct = np.ctypeslib.as_array(f.f.ctaps, shape=(2 * 245,)).view(np.complex128).copy()
        y = qh.FirBank(1, ct, d).process_host(x[None, :])
        assert rel_rms(y[0], want) < TOL64


def test_config3_decimator_many_channels(qh, oracle):
    """BASELINE config 3's FIR: 1023-tap Blackman-windowed sinc (cutoff fs/64), decimate by 32, all 64 channels of the
    configuration, two calls; reference = the oracle's quisk_cDecimate restatement on every channel."""
    n = np.arange(1023) - 511
    taps = np.sinc(n / 32.0) / 32.0 * np.blackman(1023)
    nch = 64
    x = np.stack([stream(100 + c, 32 * 700 + 13) for c in range(nch)])
    bank = qh.FirBank(nch, taps, 32)
    y = np.concatenate([bank.process_host(x[:, :9001]), bank.process_host(x[:, 9001:])], axis=1)
    for c in range(nch):
        want = oracle.OracleFir(taps).cDecimate(x[c], 32)
        assert y.shape[1] == want.size
        assert rel_rms(y[c], want) < TOL64


@pytest.mark.parametrize("ntaps,d", [(245, 5), (98, 2), (98, 1), (1023, 32), (64, 8), (147, 3)])
def test_calls_around_one_delay_line_long(qh, oracle, ntaps, d):
    """A call of at least one delay line (P = ntaps - 1 rounded up to the tile's fold) has its tail written as the next call's delay line by the
    filter kernel's own tiles (OsfirArgs::hist_next); a shorter one goes through hist_update_kernel, which keeps part of the old line.  Calls
    of P - 1, P, P + 1 samples and their neighbours in every order, two channels, against the restatement over the whole stream."""
    rng = np.random.default_rng(1000 * ntaps + d)
    taps = rng.standard_normal(ntaps) * np.hanning(ntaps + 2)[1:-1]
    fold = 8 if d % 8 == 0 else 4 if d % 4 == 0 else 2 if d % 2 == 0 else 1
    P = max(fold, (ntaps - 1 + fold - 1) // fold * fold)
    sizes = [P - 1, P, 1, P + 1, 3, P, P - 1, 2 * P + d + 1, P + 1, 7, 5 * P, 2, P, P, 1, P - 1, 4096, P + 3]
    x = np.stack([stream(7 + c, sum(sizes)) for c in range(2)])
    bank = qh.FirBank(2, taps, d)
    out, pos = [], 0
    for k in sizes:
        out.append(bank.process_host(x[:, pos:pos + k]))
        pos += k
    y = np.concatenate(out, axis=1)
    for c in range(2):
        want = oracle.OracleFir(taps).cDecimate(x[c], d)
        assert y.shape[1] == want.size
        assert rel_rms(y[c], want) < TOL64, (c, rel_rms(y[c], want))


def test_float32_bank(qh, oracle, gold):
    x = stream(31, 20000)
    want = oracle.OracleFir(gold["taps245"]).cDecimate(x, 5)
    y = qh.FirBank(1, gold["taps245"], 5, dtype=1).process_host(x[None, :].astype(np.complex64))
    assert rel_rms(y[0], want) < TOL32


def test_edge_cases(qh):
    bank = qh.FirBank(2, np.ones(5), 4)
    assert bank.process_host(np.zeros((2, 0), dtype=np.complex128)).shape == (2, 0)       # empty input
    assert bank.process_host(np.ones((2, 3), dtype=np.complex128)).shape == (2, 0)        # fewer than decim
    y = bank.process_host(np.ones((2, 1), dtype=np.complex128))                           # 4th sample -> 1 output
    assert y.shape == (2, 1) and abs(y[0, 0] - 4.0) < 1e-12
    with pytest.raises(qh.QuiskHipError):
        qh.FirBank(1, np.ones(5000), 1)                                                   # does not fit a tile
