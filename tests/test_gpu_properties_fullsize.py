"""Size-independent properties of the HIP chain at sizes the CPU oracle could not finish in seconds.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu


def _engine(qh, nch):
    e = qh.RxaEngine(nch)
    e.SetRXAShiftRun(-1, 1)
    for c in range(nch):
        e.SetRXAShiftFreq(c, synth.shift_freq(c))
    e.RXANBPSetRun(-1, 1)
    e.SetRXAMode(-1, 1)
    e.RXASetPassband(-1, 300.0, 3000.0)
    e.SetRXAAGCMode(-1, 0)
    e.SetRXAAGCFixed(-1, 0.0)
    return e


def test_chunking_invariance_and_linearity(qh):
    """One call of 512 blocks == 512/k calls of k blocks (state carry), and the chain is linear."""
    nch, nblk = 8, 512
    rng = np.random.default_rng(9)
    x1 = rng.standard_normal((nch, nblk * 1024)) + 1j * rng.standard_normal((nch, nblk * 1024))
    x2 = synth.make_input_numpy(nch, nblk * 1024)
    y1 = _engine(qh, nch).process_host(x1)
    e = _engine(qh, nch)
    parts = [e.process_host(x1[:, b * 1024:(b + 37) * 1024]) for b in range(0, nblk, 37)]
    assert rel_rms(np.concatenate(parts, axis=1), y1) < 1e-12
    y2 = _engine(qh, nch).process_host(x2)
    y12 = _engine(qh, nch).process_host(0.5 * x1 - 2.0j * x2)
    assert rel_rms(y12, 0.5 * y1 - 2.0j * y2) < 1e-12


def test_bench_shape_tone_gain_and_rejection(qh):
    """256 channels (BASELINE config 2's channel count) x 2^16 samples: every channel's in-band tone leaves
    with gain 4.0 and the out-of-band tone and the noise outside 300..3000 Hz are gone."""
    nch, nblk = 256, 64
    x = synth.make_input_numpy(nch, nblk * 1024, sigma=0.0)
    y = _engine(qh, nch).process_host(x)
    tail = y[:, -4096:]
    g = np.abs(tail).mean(axis=1) / 0.1
    assert np.all(np.abs(g - 4.0) < 1e-3)
    # a pure -1000 Hz tone at the 48 kHz output rate
    t = np.arange(4096)
    ref = np.exp(-2j * np.pi * 1000.0 / 48000.0 * t)
    for c in (0, 100, 255):
        a = np.vdot(ref, tail[c]) / 4096
        resid = tail[c] - a * ref
        assert np.sqrt(np.mean(np.abs(resid) ** 2)) < 1e-6 * abs(a)
