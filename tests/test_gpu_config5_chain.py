"""BASELINE config 5 end to end in fp32 -- fused 8 x quisk_cDecim2HB45 (61.44 Msps -> 240 k), the 245-tap / 5
(quiskFilt240D5CoefsSharp, filters.h:478) and WDSP's overlap-save bandpass (fircore, size 256, nc 2048, 300..3000 Hz,
wdsp/firmin.c:409-430 with fir_bandpass taps wdsp/fir.c:187-254) -- against the fp64 oracle (the filter.c restatement,
bit-pinned to the reference's own build, then the oracle's partitioned fircore).  north_star gate for fp32: 1e-3
relative RMS; the measured error is written next to the assertions.  Also the fp32 complex-tap FirBank on its own.
-m gpu."""
import os

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "filter_golden.npz")
FS = 61.44e6


def _input(n, seed=31):
    """Noise plus a carrier that lands at -1 kHz at 48 k -- inside WDSP's "300..3000" band, which passes conventional
    -3000..-300 Hz (the modulation of fir_bandpass has a -sin imaginary part, wdsp/fir.c:246-249) -- and one at +7 kHz."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x += 0.3 * np.exp(-2j * np.pi * 1000.0 / FS * t) + 0.2 * np.exp(2j * np.pi * 7000.0 / FS * t)
    return x


def _oracle_chain(po, x, taps245, imp):
    y = x
    for _ in range(8):
        y = po.OracleHB45().cDecim2(y)
    y = po.OracleFir(taps245).cDecimate(y, 5)
    n = (y.size // 256) * 256
    return po.Fircore(256, 2048, imp)(y[:n]), y[:n]


@pytest.fixture(scope="module")
def design(oracle):
    gold = np.load(GOLD)
    # create_bandpass / calc_bandpass: fir_bandpass(nc, f_low, f_high, rate, wintype 1?, rtype 1, gain / (2 size)), bandpass.c:302;
    # nbp0 uses wintype 0 (RXA.c:99); the 1/(2 size) is undone by the unnormalised inverse FFT of 2*size points
    imp = oracle.fir_bandpass(2048, 300.0, 3000.0, 48000.0, 0, 1, 1.0 / 512.0)
    return gold["taps245"], imp


def test_config5_chain_fp32_against_fp64_oracle(qh, oracle, design):
    taps245, imp = design
    nout = 256 * 36                     # 9216 samples at 48 k: the 2048-tap filter fills and runs 7000 samples beyond
    x = _input(1280 * nout)
    ref, ref5 = _oracle_chain(oracle, x, taps245, imp)
    # fircore's output is the linear convolution with the impulse times 2*size (test_oracle_wdsp.py): the bank takes taps
    core_taps = imp * 512.0
    for dtype, tol5, tol in ((0, 1e-12, 1e-11), (1, 2e-5, 1e-3)):
        y = qh.HalfBandCascade(1, 8, dtype=dtype).process_host(x[None, :])
        y5 = qh.FirBank(1, taps245, 5, dtype=dtype).process_host(y)
        assert y5.shape[1] == ref5.size
        e5 = rel_rms(y5[0], ref5)
        out = qh.FirBank(1, core_taps, 1, dtype=dtype).process_host(y5)
        e = rel_rms(out[0], ref)
        print("config 5 chain dtype %d: after /5 %.3e, after bandpass %.3e (rel RMS vs fp64 oracle)" % (dtype, e5, e))
        assert e5 < tol5 and e < tol
        # the in-band carrier comes through, the +7 kHz one and the noise outside the pass band do not
        tail = out[0, -4800:].astype(np.complex128)        # 100 cycles of 1 kHz, 700 of 7 kHz: no leakage between the two
        k = np.arange(4800)
        a = np.vdot(np.exp(-2j * np.pi * 1000.0 / 48000.0 * k), tail) / 4800
        assert abs(abs(a) - 0.3) < 0.01
        b = np.vdot(np.exp(2j * np.pi * 7000.0 / 48000.0 * k), tail) / 4800
        assert abs(b) < 1e-4


def test_config5_chain_fp32_ragged_calls_carry_state(qh, oracle, design):
    """The same chain fed in uneven pieces (multiples of the cascade's 256-sample unit) equals the one-call result."""
    taps245, imp = design
    x = _input(1280 * 256 * 12, seed=5).astype(np.complex64)[None, :]
    casc, d5, core = (qh.HalfBandCascade(1, 8, dtype=1), qh.FirBank(1, taps245, 5, dtype=1), qh.FirBank(1, imp * 512.0, 1, dtype=1))
    whole = core.process_host(d5.process_host(casc.process_host(x)))
    for o in (casc, d5, core):
        o.reset()
    parts, pos = [], 0
    for units in (1, 700, 3, 2999, 41, 11616):
        seg = x[:, pos:pos + 256 * units]
        pos += 256 * units
        parts.append(core.process_host(d5.process_host(casc.process_host(seg))))
    assert pos == x.shape[1]
    got = np.concatenate(parts, axis=1)
    assert got.shape == whole.shape
    assert rel_rms(got, whole) < 2e-6       # fp32: tile boundaries fall elsewhere, the sums round differently


@pytest.mark.parametrize("decim,ntaps", [(1, 2048), (1, 300), (3, 1001), (4, 512)])
def test_fp32_complex_tap_firbank(qh, oracle, decim, ntaps):
    """quisk_cCDecimate-shaped bank (complex taps) in fp32 against the fp64 direct form."""
    rng = np.random.default_rng(7 + ntaps)
    taps = (rng.standard_normal(ntaps) + 1j * rng.standard_normal(ntaps)) * np.hanning(ntaps) / ntaps ** 0.5
    x = (rng.standard_normal((2, 20000)) + 1j * rng.standard_normal((2, 20000)))
    y = qh.FirBank(2, taps, decim, dtype=1).process_host(x)
    y64 = qh.FirBank(2, taps, decim, dtype=0).process_host(x)
    for ch in range(2):
        full = np.convolve(x[ch], taps)[:x.shape[1]]
        want = full[decim - 1::decim]           # filter.c:203-229: the first output after `decim` inputs
        assert y.shape[1] == want.size
        assert rel_rms(y64[ch], want) < 1e-12
        assert rel_rms(y[ch], want) < 2e-5
