"""The fused half-band cascade (C ABI group 3b) against the filter.c restatement's quisk_cDecim2HB45 chained
nstage times (bit-exactly pinned to the reference's own filter.c build) and the golden cascade vector.  -m gpu."""
import os

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "filter_golden.npz")
TOL64, TOL32 = 1e-12, 2e-5          # north_star: <= 1e-6 (float64) / <= 1e-3 (float32) relative RMS


def stream(seed, nch, n):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((nch, n)) + 1j * rng.standard_normal((nch, n))


def oracle_chain(oracle, x, nstage):
    out = []
    for ch in range(x.shape[0]):
        y = x[ch]
        for _ in range(nstage):
            y = oracle.OracleHB45().cDecim2(y)
        out.append(y)
    return np.stack(out)


@pytest.mark.parametrize("nstage", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("dtype,tol", [(0, TOL64), (1, TOL32)])
def test_cascade_matches_chained_hb45(qh, oracle, nstage, dtype, tol):
    n = (1 << nstage) * 300 if nstage >= 5 else 2048 * 5 + (1 << nstage) * 3       # ragged last step
    x = stream(21 + nstage, 3, n)
    ref = oracle_chain(oracle, x, nstage)
    y = qh.HalfBandCascade(3, nstage, dtype=dtype).process_host(x)
    assert y.shape == ref.shape
    for ch in range(3):
        assert rel_rms(y[ch], ref[ch]) < tol


@pytest.mark.parametrize("nstage", [2, 8])
def test_state_carries_between_ragged_calls(qh, oracle, nstage):
    """History shorter and longer than the calls: 1, 40, 3, 200, 7 decimation units."""
    d = 1 << nstage
    splits = [d * k for k in (1, 40, 3, 200, 0, 7)]
    x = stream(5, 2, sum(splits))
    ref = oracle_chain(oracle, x, nstage)
    c = qh.HalfBandCascade(2, nstage)
    out, pos = [], 0
    for k in splits:
        out.append(c.process_host(x[:, pos:pos + k]))
        pos += k
    y = np.concatenate(out, axis=1)
    assert y.shape == ref.shape
    assert rel_rms(y, ref) < TOL64
    c.reset()
    assert rel_rms(c.process_host(x[:, :d * 50]), ref[:, :50]) < TOL64


@pytest.mark.parametrize("nstage,dtype,tol", [(1, 0, TOL64), (3, 0, TOL64), (4, 1, TOL32), (8, 0, TOL64), (6, 1, TOL32)])
def test_calls_around_one_history_long(qh, oracle, nstage, dtype, tol):
    """A launch keeps ceil(42 (2^s - 1) / 4096) x 4096 input samples of history (s = its own stages: a cascade of six and more runs as 4 + the
    rest); a call at least that long has it written by the kernel's own segments, a shorter one by hb45_hist_kernel, which keeps part of the
    old history.  Calls one decimation unit short of it, exactly it, one over, and their neighbours in turn."""
    d = 1 << nstage
    head = 4 if nstage >= 6 else nstage
    W = (42 * ((1 << head) - 1) + 4095) // 4096 * 4096
    sizes = [W - d, W, d, W + d, 3 * d, W, W - d, 2 * W + 5 * d, d, W // 16 // d * d + d, W, 4 * W, W - d, W + d]
    x = stream(31 + nstage, 2, sum(sizes))
    ref = oracle_chain(oracle, x, nstage)
    c = qh.HalfBandCascade(2, nstage, dtype=dtype)
    out, pos = [], 0
    for k in sizes:
        out.append(c.process_host(x[:, pos:pos + k].astype(np.complex64) if dtype else x[:, pos:pos + k]))
        pos += k
    y = np.concatenate(out, axis=1)
    assert y.shape == ref.shape
    for ch in range(2):
        assert rel_rms(y[ch], ref[ch]) < tol, (ch, rel_rms(y[ch], ref[ch]))


def test_multi_segment_long_stream(qh, oracle):
    """Long enough for several segments per channel (segment boundaries re-derive state by warm-up)."""
    x = stream(9, 1, 1 << 20)
    ref = oracle_chain(oracle, x, 4)
    y = qh.HalfBandCascade(1, 4).process_host(x)
    assert rel_rms(y, ref) < TOL64


def test_golden_config5_front(qh):
    """BASELINE config 5: 8 x HB45 fused, then the 245-tap /5 bank, against the vector made from the reference build."""
    gold = np.load(GOLD)
    rng = np.random.default_rng(13)
    n = 256 * 5 * 40
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n))[None, :]
    for dtype, tol in ((0, TOL64), (1, TOL32)):
        y = qh.HalfBandCascade(1, 8, dtype=dtype).process_host(x)
        y = qh.FirBank(1, gold["taps245"], 5, dtype=dtype).process_host(y)
        assert y.shape[1] == gold["cascade_8hb45_d5"].size
        assert rel_rms(y[0], gold["cascade_8hb45_d5"]) < tol


def test_rejects_partial_decimation_unit(qh):
    c = qh.HalfBandCascade(1, 3)
    with pytest.raises(qh.QuiskHipError):
        c.process_host(np.zeros((1, 12), dtype=np.complex128))
