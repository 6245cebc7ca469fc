"""The polyphase rational resampler (C ABI group 3c) against the golden vectors made from the reference's own
filter.c build and against its restatement, with the same ragged call pattern.  -m gpu."""
import os

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "filter_golden.npz")
SPLITS = [0, 1, 7, 333, 2, 64, 1000, 5, 0, 588]
TOL64, TOL32 = 1e-12, 2e-5          # north_star: <= 1e-6 (float64) / <= 1e-3 (float32) relative RMS


def stream(seed, n, complex_=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    if complex_:
        x = x + 1j * rng.standard_normal(n)
    return x


def run_split(bank, x):
    out, pos = [], 0
    for k in SPLITS:
        out.append(bank.process_host(x[:, pos:pos + k]))
        pos += k
    return np.concatenate(out, axis=1)


def hb45_interp_taps(qh):
    """quisk_cInterp2HB45 (filter.c:455-488) as a 2-phase polyphase filter: 45 taps, centre 0.5."""
    t = qh.hb45_taps()
    g = np.zeros(45)
    for k in range(11):
        g[2 * k + 1] = t[2 * k]
        g[43 - 2 * k] = t[2 * k]
    g[22] = 0.5
    return g


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("dtype,tol", [(0, TOL64), (1, TOL32)])
def test_golden_cinterpolate(qh, gold, dtype, tol):
    x = stream(11, sum(SPLITS))[None, :]
    y = run_split(qh.RationalFir(1, gold["taps36"], 2, dtype=dtype), x)
    assert y.shape[1] == gold["cInterpolate_36_x2"].size
    assert rel_rms(y[0], gold["cInterpolate_36_x2"]) < tol


def test_golden_dinterpolate_two_real_streams_in_one_complex(qh, gold):
    xr = stream(12, sum(SPLITS), False)
    x = (xr + 1j * xr[::-1])[None, :]                   # a second real stream rides in the imaginary part
    bank = qh.RationalFir(1, gold["taps36"], 3)
    y = bank.process_host(x)
    assert rel_rms(y[0].real, gold["dInterpolate_36_x3"]) < TOL64
    ref_im = qh.RationalFir(1, gold["taps36"], 3).process_host((xr[::-1] + 0j)[None, :])[0].real
    assert rel_rms(y[0].imag, ref_im) < TOL64


def test_golden_cinterpdecim(qh, gold):
    x = stream(11, sum(SPLITS))[None, :]
    y = run_split(qh.RationalFir(1, gold["taps98"], 2, 3), x)
    assert y.shape[1] == gold["cInterpDecim_98_6_5"].size
    assert rel_rms(y[0], gold["cInterpDecim_98_6_5"]) < TOL64


def test_golden_interp2hb45(qh, gold):
    x = stream(11, sum(SPLITS))[None, :]
    y = run_split(qh.RationalFir(1, hb45_interp_taps(qh), 2), x)
    assert y.shape[1] == gold["cInterp2HB45"].size
    assert rel_rms(y[0], gold["cInterp2HB45"]) < TOL64
    xr = stream(12, sum(SPLITS), False)
    y = run_split(qh.RationalFir(1, hb45_interp_taps(qh), 2), (xr + 0j)[None, :])
    assert rel_rms(y[0].real, gold["dInterp2HB45"]) < TOL64


@pytest.mark.parametrize("interp,decim", [(6, 5), (4, 5), (1, 4), (3, 7), (8, 1)])
def test_ratios_against_restatement(qh, oracle, interp, decim):
    """Quisk's 6/5 and 4/5 stages (quisk.c:1836-1837) and a few others, multi-channel, ragged calls."""
    rng = np.random.default_rng(interp * 10 + decim)
    taps = rng.standard_normal(125 if interp == 6 else 48 * max(interp, 1))
    taps = taps[:(taps.size // interp) * interp]
    x = np.stack([stream(30 + c, sum(SPLITS)) for c in range(3)])
    bank = qh.RationalFir(3, taps, interp, decim)
    y = run_split(bank, x)
    for c in range(3):
        f = oracle.OracleFir(taps)
        ref, pos = [], 0
        for k in SPLITS:
            ref.append(f.cInterpDecim(x[c, pos:pos + k], interp, decim))
            pos += k
        ref = np.concatenate(ref)
        assert y.shape[1] == ref.size
        assert rel_rms(y[c], ref) < TOL64


def test_reset_and_phase(qh, gold):
    bank = qh.RationalFir(1, gold["taps98"], 2, 3)
    x = stream(3, 100)[None, :]
    a = bank.process_host(x)
    assert bank.phase == (a.shape[1] * 3 - 100 * 2)
    bank.reset()
    assert bank.phase == 0
    assert rel_rms(bank.process_host(x), a) < 1e-15
