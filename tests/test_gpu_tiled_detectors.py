"""Time-tiled detectors (quisk_amd/csrc/qh_tiled.hpp).  The linear recurrences (AM fade leveller, FM dc removal, CTCSS notch)
are cut into 16 time segments per call and chained exactly; the FM loop runs one tile of 256 samples per lane with a
768-sample warm-up from a zero state, and a verify pass re-runs in order the tiles whose warm-up had not met the true
trajectory (no carrier: two runs of the loop on noise meet after ~135 samples on average, with an exponential tail).  A call
that is fed block by block takes the sequential route through the same kernels (every tile begins at sample 0 of the call,
from the carried state), so "one long call == many short calls" measures what tiling adds; the oracle comparison (after lock,
DESIGN.md parity caveat) ties both to the reference's sample-by-sample loop (wdsp/fmd.c:151-172, amd.c:131-146,
iir.c:76-95).  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu
AM, FM = 6, 5


def _engine(qh, nch, mode, **kw):
    e = qh.RxaEngine(nch)
    for c in range(nch):
        e.SetRXAShiftRun(c, 1)
        e.SetRXAShiftFreq(c, synth.shift_freq(c))
        e.RXANBPSetRun(c, 1)
        e.SetRXAMode(c, mode)
        e.SetRXAAGCMode(c, 0)
        e.SetRXAAGCFixed(c, 0.0)
        e.RXASetPassband(c, *((-8000.0, 8000.0) if mode == FM else (-4000.0, 4000.0)))
        if "ctcss_run" in kw:
            e.SetRXACTCSSRun(c, kw["ctcss_run"])
    return e


def _oracle(po, c, mode):
    o = po.WdspChannel(1024, 256, 192000, 48000, 48000)
    o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(c)); o.RXANBPSetRun(1); o.SetRXAMode(mode)
    o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0)
    o.RXASetPassband(*((-8000.0, 8000.0) if mode == FM else (-4000.0, 4000.0)))
    return o


def _fm_inputs(nblk):
    """clean FM, FM at low SNR (noise 14 dB under the carrier in the full 192 kHz), FM that fades out, and no carrier at all"""
    n = nblk * 1024
    x = np.stack([synth.make_mode_input_numpy("fm", c, n, sigma=s) for c, s in ((0, 0.01), (1, 0.02), (2, 0.01), (3, 0.01))])
    x[2] *= np.linspace(1.0, 0.0, n) ** 2                   # carrier sinks into its own noise floor ... and below
    rng = np.random.default_rng(77)
    x[2] += 0.003 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x[3] = 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x


# While the filters ahead of the detector fill, its input is FFT round-off (1e-17) and atan2 is scale invariant: the loop's
# path through those first blocks depends on the last bit of the transform in use, in the reference as much as here
# (parity caveat).  What that start-up leaves behind decays with the dc removal (0.02 s) and, when it runs,
# the CTCSS notch (pole radius 0.9994): comparisons start `settle` blocks in.
def test_fm_one_long_call_equals_block_calls_and_the_oracle(qh, oracle):
    nblk = 200                                              # 51 200 samples at 48 k = 200 tiles per channel in the long call
    x = _fm_inputs(nblk)
    el = _engine(qh, 4, FM, ctcss_run=0)
    long = el.process_host(x)
    print("tiles re-run by the verify pass: %d of %d" % (el.pll_repairs(), 4 * 200))
    assert el.pll_repairs() < 40                            # speculation mostly holds, even on the carrier-less channels
    e = _engine(qh, 4, FM, ctcss_run=0)
    short = np.concatenate([e.process_host(np.ascontiguousarray(x[:, b * 1024:(b + 1) * 1024])) for b in range(nblk)], axis=1)
    settle = 100 * 256
    for c in range(4):
        # block calls run every tile from the carried state: the sequential loop.  The long call's tiles are either verified
        # (their warm-up met the predecessor's end state to 1e-12) or re-run from it: same trajectory
        err = rel_rms(long[c][settle:], short[c][settle:])
        assert err < 1e-9, (c, err)
    long = _engine(qh, 4, FM).process_host(x)               # with the notch, against the oracle
    settle = 144 * 256
    for c in (0, 1):
        ref = _oracle(oracle, c, FM).xrxa(x[c])
        assert rel_rms(long[c][settle:], ref[settle:]) < 1e-6, c
        assert np.abs(ref[-4096:]).max() > 1e-3


def test_fm_state_carries_across_ragged_long_calls(qh):
    nblk = 200
    x = _fm_inputs(nblk)[:2]
    whole = _engine(qh, 2, FM, ctcss_run=0).process_host(x)
    e = _engine(qh, 2, FM, ctcss_run=0)
    parts, pos = [], 0
    for nb in (1, 37, 5, 64, 93):
        parts.append(e.process_host(np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])))
        pos += nb
    assert pos == nblk
    settle = 100 * 256
    assert rel_rms(np.concatenate(parts, axis=1)[:, settle:], whole[:, settle:]) < 1e-9


@pytest.mark.parametrize("mode,sig", [(AM, "am"), (FM, "fm")])
def test_linear_scans_are_exact_across_segments(qh, oracle, mode, sig):
    """AM leveller (two one-pole averages, tau 0.02 s and 1.4 s) and the FM chain's dc removal + CTCSS notch, call lengths that
    put 1 .. 16 segments to work (a segment is a sixteenth of the call, in whole batches of 64 samples)."""
    nblk = 150
    x = np.stack([synth.make_mode_input_numpy(sig, c, nblk * 1024) for c in range(3)])
    e = _engine(qh, 3, mode)
    parts, pos = [], 0
    for nb in (1, 2, 3, 9, 35, 100):                        # 256 ... 25 600 samples per call
        parts.append(e.process_host(np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])))
        pos += nb
    y = np.concatenate(parts, axis=1)
    settle = 0 if mode == AM else 120 * 256
    for c in range(3):
        ref = _oracle(oracle, c, mode).xrxa(x[c])
        assert rel_rms(y[c][settle:], ref[settle:]) < (1e-8 if mode == AM else 1e-6), c


SAM = 10


def _sam_input(c, n, df, seed):
    """AM carrier df Hz off the channel's centre, two audio tones, noise 30 dB under the carrier in the full bandwidth"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 192000.0
    env = 1.0 + 0.5 * np.cos(2 * np.pi * 700.0 * t) + 0.3 * np.cos(2 * np.pi * 1900.0 * t)
    x = 0.2 * env * np.exp(2j * np.pi * (df - synth.shift_freq(c)) * t)
    return x + 0.006 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))


def test_sam_long_calls_are_time_tiled_and_equal_block_calls_and_the_oracle(qh, oracle):
    """SAM without sideband separation: calls of 32768 DSP-rate samples and more run the loop one 4096-sample tile per lane behind
    an 8192-sample warm-up (qh_tiled.hpp), shorter ones the sequential kernel.  Acquisition amplifies last-bit differences and
    the fade leveller (1.4 s) remembers them for seconds, so the two engines that are compared acquire the same way -- block by
    block -- and part only afterwards: what is left is what tiling adds."""
    nacq, nlong = 160, 160
    offs = (12.0, -35.0, 30.0)
    x = np.stack([_sam_input(c, (nacq + 2 * nlong) * 1024, offs[c], 40 + c) for c in range(3)])
    ea, eb = _engine(qh, 3, SAM), _engine(qh, 3, SAM)
    for b in range(nacq):
        blk = np.ascontiguousarray(x[:, b * 1024:(b + 1) * 1024])
        assert np.array_equal(ea.process_host(blk), eb.process_host(blk))
    long = np.concatenate([ea.process_host(np.ascontiguousarray(x[:, (nacq + k * nlong) * 1024:(nacq + (k + 1) * nlong) * 1024])) for k in range(2)], axis=1)
    print("tiles re-run by the verify pass: %d of %d" % (ea.pll_repairs(), 3 * 2 * 10))
    assert ea.pll_repairs() == 0                            # locked: the speculation holds
    short = np.concatenate([eb.process_host(np.ascontiguousarray(x[:, b * 1024:(b + 1) * 1024])) for b in range(nacq, nacq + 2 * nlong)], axis=1)
    assert eb.pll_repairs() == 0                            # block calls never tile
    assert rel_rms(long, short) < 1e-11
    # long calls from a cold start (tiles fail their check while the loop acquires and are stepped in order), against the
    # oracle, the leveller off so that the comparison can begin once the loop has locked
    ec = _engine(qh, 3, SAM)
    for c in range(3):
        ec.SetRXAAMDFadeLevel(c, 0)
    y = np.concatenate([ec.process_host(np.ascontiguousarray(x[:, k * nlong * 1024:(k + 1) * nlong * 1024])) for k in range(3)], axis=1)
    for c in range(3):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(c)); o.RXANBPSetRun(1); o.SetRXAMode(SAM)
        o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0); o.RXASetPassband(-4000.0, 4000.0); o.SetRXAAMDFadeLevel(0)
        ref = o.xrxa(x[c])
        settle = 160 * 256
        assert rel_rms(y[c][settle:], ref[settle:]) < 1e-6, c
        assert np.abs(ref[-4096:]).max() > 1e-2


@pytest.mark.parametrize("sbmode", [1, 2], ids=["SAM-L", "SAM-U"])
def test_sam_sideband_modes_over_time_segments(qh, oracle, sbmode):
    """SAM with a sideband selected: behind the time-tiled loop the four all-pass chains (poles up to 0.9999: no warm-up reaches that
    far) run over time segments -- zero-state end states, start states by the chains' transition matrices, the chains again from the
    true states (qh_tiled.hpp, sam_sb_*).  Long calls equal block calls (which step the chains 64 samples at a time on one wavefront),
    the state crosses between the two forms, and both follow the oracle."""
    nacq, nlong = 160, 160
    offs = (12.0, -35.0, 30.0)
    x = np.stack([_sam_input(c, (nacq + 2 * nlong) * 1024, offs[c], 60 + c) for c in range(3)])

    def make():
        e = _engine(qh, 3, SAM)
        for c in range(3):
            e.SetRXAAMDSBMode(c, sbmode)
        return e
    ea, eb = make(), make()
    for b in range(nacq):
        blk = np.ascontiguousarray(x[:, b * 1024:(b + 1) * 1024])
        assert np.array_equal(ea.process_host(blk), eb.process_host(blk))
    long = np.concatenate([ea.process_host(np.ascontiguousarray(x[:, (nacq + k * nlong) * 1024:(nacq + (k + 1) * nlong) * 1024])) for k in range(2)], axis=1)
    short = np.concatenate([eb.process_host(np.ascontiguousarray(x[:, b * 1024:(b + 1) * 1024])) for b in range(nacq, nacq + 2 * nlong)], axis=1)
    assert np.abs(short).max() > 1e-2
    assert rel_rms(long, short) < 1e-10, rel_rms(long, short)
    # a long call, short calls, a long call: the chains' state goes from one form to the other and back
    ec = make()
    for c in range(3):
        ec.SetRXAAMDFadeLevel(c, 0)
    cuts = [0, nlong, nlong + 7, nlong + 8, 2 * nlong + 8, 3 * nlong]
    y = np.concatenate([ec.process_host(np.ascontiguousarray(x[:, a * 1024:b * 1024])) for a, b in zip(cuts[:-1], cuts[1:])], axis=1)
    for c in range(3):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(c)); o.RXANBPSetRun(1); o.SetRXAMode(SAM); o.SetRXAAMDSBMode(sbmode)
        o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0); o.RXASetPassband(-4000.0, 4000.0); o.SetRXAAMDFadeLevel(0)
        ref = o.xrxa(x[c])
        settle = 320 * 256                  # the chains remember the acquisition (poles up to 0.9999): compared once that has gone
        assert rel_rms(y[c][settle:], ref[settle:]) < 1e-6, (c, rel_rms(y[c][settle:], ref[settle:]))
