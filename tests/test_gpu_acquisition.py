"""The detectors' ACQUISITION phase against the oracle, from the first detector sample.

The other parity tests compare AM / SAM / FM behind the lock-in: a channel that starts in one of these modes feeds its detector the
rounding noise of whichever FFT filled the filters (1e-28 .. 1e-16 of full scale), a scale-invariant detector (atan2) amplifies the
last bit of that to full scale, and two correct implementations part until the loop has locked.  Here the
detectors start on a WELL-CONDITIONED input instead: the channel runs as USB until its filters are full of a strong carrier, then
SetRXAMode switches the detector in (its state still the initial one: xamd / xfmd never ran) -- on both sides at the same block.
From that sample on the loop's pull-in (SAM: zeta 1, omega_N 250 rad/s from phase 0 and the frequency limits; FM: its 20 krad/s
loop from rest) is a smooth function of the input, and the GPU must follow the oracle through it, not only behind it.  Short calls
take the sequential kernels, the long call the time-tiled ones (whose tiles fail their check while the loop is still moving and
are then stepped in order).  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu


def _chan(po, c):
    o = po.WdspChannel(1024, 256, 192000, 48000, 48000)
    o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(c)); o.RXANBPSetRun(1); o.SetRXAMode(1)
    o.RXASetPassband(300.0, 3000.0); o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0)
    return o


@pytest.mark.parametrize("calls", [[3, 1, 2, 5, 1, 8, 4, 16, 20, 40, 100], [200]], ids=["short-calls", "one-long-call"])
@pytest.mark.parametrize("mode,sig,passband,sbmode", [(6, "am", (-4000.0, 4000.0), 0), (10, "am", (-4000.0, 4000.0), 0),
                                                      (10, "am", (-4000.0, 4000.0), 1), (5, "fm", (-8000.0, 8000.0), 0)],
                         ids=["AM", "SAM", "SAM-L", "FM"])
def test_detector_pull_in_matches_the_oracle_from_its_first_sample(qh, oracle, mode, sig, passband, sbmode, calls):
    nch, prime = 3, 12                                   # 12 blocks of USB: every filter of the chain holds carrier samples
    nblk = prime + sum(calls)
    x = np.stack([synth.make_mode_input_numpy(sig, c, nblk * 1024, sigma=1e-4) for c in range(nch)])
    if mode == 10:                                       # SAM: the carrier 35 Hz off centre, inside the loop's pull-in range
        t = np.arange(nblk * 1024)
        x = x * np.exp(2j * np.pi * 35.0 / 192000.0 * t)[None, :]
    e = qh.RxaEngine(nch)
    refs = [_chan(oracle, c) for c in range(nch)]
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
        e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 0); e.SetRXAAGCFixed(c, 0.0)
    y0 = e.process_host(np.ascontiguousarray(x[:, :prime * 1024]))
    for c in range(nch):
        assert rel_rms(y0[c], refs[c].xrxa(x[c, :prime * 1024])) < 1e-9
    for c in range(nch):                                 # the detector comes in on a primed chain
        e.SetRXAMode(c, mode); e.RXASetPassband(c, *passband)
        refs[c].SetRXAMode(mode); refs[c].RXASetPassband(*passband)
        if sbmode:
            e.SetRXAAMDSBMode(c, sbmode); refs[c].SetRXAAMDSBMode(sbmode)
    outs, pos = [], prime
    for nb in calls:
        outs.append(e.process_host(np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])))
        pos += nb
    y = np.concatenate(outs, axis=1)
    for c in range(nch):
        want = refs[c].xrxa(x[c, prime * 1024:])
        assert np.abs(want).max() > 0.01
        # from the FIRST sample behind the switch: the whole pull-in, then the locked stretch
        first = slice(0, 40 * 256)
        assert rel_rms(y[c][first], want[first]) < 1e-6, (c, rel_rms(y[c][first], want[first]))
        assert rel_rms(y[c], want) < 1e-6, (c, rel_rms(y[c], want))


def test_mixed_modes_in_the_direct_form_with_detector_setters_between_ragged_calls(qh, oracle):
    """USB, AM and FM channels in one engine with fixed gain: the form in which the last filter of every channel stores to the caller's rows,
    the FM channels' dc removal rides in the de-emphasis stage's load and the AM channels' fade leveller in nbp0's store and bp1's load
    (qh_engine.hip fmdc_fused / am_lv_fused, round 6).  Calls of 1 .. 90 blocks -- one sample short of, on and past the detectors' and the
    filters' tile boundaries -- with SetRXAAMDFadeLevel, SetRXAFMDeviation, SetRXACTCSSRun and a mode change in between, each channel
    against its restatement from the first detector sample on."""
    nch, prime = 9, 12
    kinds = ["usb", "am", "fm"]
    modes = {"usb": (1, (300.0, 3000.0)), "am": (6, (-4000.0, 4000.0)), "fm": (5, (-8000.0, 8000.0))}
    calls = [1, 7, 8, 9, 1, 16, 40, 3, 90, 2, 33, 8, 64, 5]
    nblk = prime + sum(calls)
    x = np.stack([synth.make_mode_input_numpy(kinds[c % 3], c, nblk * 1024, sigma=1e-4) for c in range(nch)])
    e = qh.RxaEngine(nch)
    refs = [_chan(oracle, c) for c in range(nch)]
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
        e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 0); e.SetRXAAGCFixed(c, 0.0)
    y0 = e.process_host(np.ascontiguousarray(x[:, :prime * 1024]))
    for c in range(nch):
        assert rel_rms(y0[c], refs[c].xrxa(x[c, :prime * 1024])) < 1e-9
    for c in range(nch):
        m, pb = modes[kinds[c % 3]]
        e.SetRXAMode(c, m); e.RXASetPassband(c, *pb); refs[c].SetRXAMode(m); refs[c].RXASetPassband(*pb)
    # setters between calls: (call index, channel, name, args)
    plan = {2: [(1, "SetRXAAMDFadeLevel", (0,))], 4: [(2, "SetRXAFMDeviation", (2500.0,))], 6: [(1, "SetRXAAMDFadeLevel", (1,)), (5, "SetRXACTCSSRun", (0,))],
            8: [(4, "SetRXAAMDFadeLevel", (0,)), (8, "SetRXAFMDeviation", (5000.0,))], 10: [(7, "SetRXAMode", (1,)), (5, "SetRXACTCSSRun", (1,))],
            12: [(4, "SetRXAAMDFadeLevel", (1,)), (2, "SetRXACTCSSFreq", (100.0,))]}
    pos, got, want = prime, [[] for _ in range(nch)], [[] for _ in range(nch)]
    for k, nb in enumerate(calls):
        for c, name, args in plan.get(k, []):
            getattr(e, name)(c, *args); getattr(refs[c], name)(*args)
            if name == "SetRXAMode":
                e.RXASetPassband(c, 300.0, 3000.0); refs[c].RXASetPassband(300.0, 3000.0)
        seg = np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])
        y = e.process_host(seg)
        for c in range(nch):
            got[c].append(y[c]); want[c].append(refs[c].xrxa(seg[c]))
        pos += nb
    for c in range(nch):
        g, w = np.concatenate(got[c]), np.concatenate(want[c])
        assert np.abs(w).max() > 0.01
        # the FM loop pulls in from rest on both sides; behind the pull-in the two agree to rounding, through it to 1e-6 of the signal
        assert rel_rms(g, w) < 1e-6, (c, kinds[c % 3], rel_rms(g, w))
        tail = slice(40 * 256, None)
        assert rel_rms(g[tail], w[tail]) < 1e-8, (c, kinds[c % 3], rel_rms(g[tail], w[tail]))
    e.close()
