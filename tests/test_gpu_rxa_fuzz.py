"""Seeded random walks over the RXA engine's setters, applied identically to the GPU engine (per channel) and to one
oracle channel per GPU channel, with a few DSP blocks between changes: state carried across parameter changes (filter
histories, AGC, LMS weights, notch database, mask rebuilds) is where a batched re-implementation goes wrong first.
Gate: relative RMS over the whole run <= 1e-6 per channel (fp64 chain tolerance); 1e-4 for a channel on which ANF or ANR
was ever switched on -- the LMS predictor of a nearly periodic or DC-heavy input (an AM envelope, say) is ill-conditioned,
its weights wander along the null space on rounding noise, and two correct implementations that sum 64 products in a
different order agree to a few 1e-6 there, not to 1e-13 (seen: 1.0e-6 .. 3.6e-6 on 3 of 24 walks).  -m gpu."""
import numpy as np
import pytest
import torch          # before libquiskhip: one HIP runtime per process (torch's), as in bench.py

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu

NCH = 4


def _apply(rng, targets, wide=False):
    """Draw one setter call and apply it to every (object, leading-args) pair.  wide: also the long filters (RXASetNC to 16384: partitions,
    tests/test_gpu_long_nc.py) and the AGC's time constants (a moved attack window takes the channel off the time tiles)."""
    k = int(rng.integers(0, 29 if wide else 25))

    done = []

    def call(name, *args):
        done.append((name,) + args)
        for t, lead in targets:
            getattr(t, name)(*lead, *args)
    if k == 0:
        mode = int(rng.choice([0, 1, 3, 4, 6, 10, 7]))          # LSB USB CWL CWU AM SAM DIGU
        call("SetRXAMode", mode)
    elif k == 1:
        lo = float(rng.uniform(-4000, 3000)); hi = lo + float(rng.uniform(200, 4000))
        call("RXASetPassband", lo, hi)
    elif k == 2:
        call("SetRXAShiftFreq", float(rng.uniform(-20000, 20000)))
    elif k == 3:
        call("SetRXAAGCMode", int(rng.integers(0, 5)))
    elif k == 4:
        call("SetRXAAGCFixed", float(rng.uniform(-10, 30)))
    elif k == 5:
        call("RXASetNC", int(rng.choice([256, 512, 1024, 2048])))
    elif k == 6:
        call("SetRXAANFRun", int(rng.integers(0, 2)))
    elif k == 7:
        call("SetRXAANRRun", int(rng.integers(0, 2)))
    elif k == 8:
        pos = int(rng.integers(0, 2))
        call("SetRXAANFPosition", pos); call("SetRXAANRPosition", pos)
    elif k == 9:
        call("SetRXAPanelGain1", float(rng.uniform(0.5, 6.0)))
    elif k == 10:
        call("SetRXAPanelSelect", int(rng.integers(0, 4))); call("SetRXAPanelCopy", int(rng.integers(0, 4)))
    elif k == 11:
        call("RXASetMP", int(rng.integers(0, 2)))
    elif k == 12:
        call("SetRXAShiftRun", int(rng.integers(0, 2)))
    elif k == 13:
        call("SetRXAAMDFadeLevel", int(rng.integers(0, 2))); call("SetRXAAMDSBMode", int(rng.integers(0, 3)))
    elif k == 14:
        call("RXANBPSetNotchesRun", int(rng.integers(0, 2)))
    elif k == 15:        # a notch inside the usual passbands (the database refuses out-of-order indices: always append at 0)
        call("RXANBPAddNotch", 0, float(rng.uniform(-3000, 3000)), float(rng.uniform(50, 400)), int(rng.integers(0, 2)))
    elif k == 16:
        call("RXANBPSetTuneFrequency", float(rng.uniform(-500, 500)))
    elif k == 17:
        call("SetRXAAMDRun", int(rng.integers(0, 2)))
    elif k == 18:
        call("SetRXABandpassFreqs", float(rng.uniform(-4000, 0)), float(rng.uniform(100, 4000)))
    elif k == 19:
        call("SetRXAAGCTop", float(rng.uniform(40, 100))); call("SetRXAAGCSlope", int(rng.integers(0, 20)))
    elif k == 20:
        call("SetRXAAMSQRun", int(rng.integers(0, 2)))
    elif k == 21:
        call("SetRXAAMSQThreshold", float(rng.uniform(-60, -10))); call("SetRXAAMSQMaxTail", float(rng.uniform(0.0, 0.3)))
    elif k == 22:
        call("SetRXAEMNRRun", int(rng.integers(0, 2)))
    elif k == 23:
        call("SetRXAEMNRgainMethod", int(rng.integers(0, 4))); call("SetRXAEMNRaeRun", int(rng.integers(0, 2)))
    elif k == 24:
        pos = int(rng.integers(0, 2))
        call("SetRXAEMNRPosition", pos); call("SetRXAEMNRnpeMethod", int(rng.integers(0, 3)))
    elif k == 25:
        call("RXASetNC", int(rng.choice([256, 2048, 4096, 8192, 16384])))
    elif k == 26:
        call("SetRXAAGCAttack", int(rng.integers(1, 6)))
    elif k == 27:
        call("SetRXAAGCDecay", int(rng.choice([50, 250, 500, 2000]))); call("SetRXAAGCHang", int(rng.choice([0, 100, 500])))
    else:
        call("SetRXAAGCHangThreshold", int(rng.integers(0, 101)))
    return done


@pytest.mark.parametrize("seed", list(range(1, 41)))
def test_random_setter_walk(qh, oracle, seed):
    _walk(qh, oracle, seed, replay=False)


@pytest.mark.parametrize("seed", list(range(101, 113)) + list(range(5001, 5009)))
def test_random_setter_walk_block_at_a_time_with_graph_replay(qh, oracle, seed):
    """The same walks fed one DSP block per call from fixed device buffers with qh_rxa_set_graph_replay on: every setter
    must invalidate the captured launch sequences, and replayed blocks must leave the ping-pong state where plain ones do."""
    _walk(qh, oracle, seed, replay=True)


@pytest.mark.parametrize("seed", list(range(201, 213)) + [900190])
def test_random_setter_walk_with_long_filters_agc_windows_and_long_calls(qh, oracle, seed):
    """The walks with RXASetNC up to 16384 and the AGC's time constants among the setters, and calls of 70 - 90 DSP blocks among the short
    ones: the time-tiled detectors and AGC, the per-channel fall-back after a moved attack window and the partitioned filters meet the
    per-block paths on one carried state.  (900190, round 6's sweep: the one channel with nc = 16384 leaves bp1's list -- SetRXAAMDRun 0 -- and
    comes back by SetRXABandpassRun in a long call: the stage's form may not follow the running channels alone, or its 16383-sample line
    is cut to 4095 while it sits out.)"""
    _walk(qh, oracle, seed, replay=False, wide=True)


def _walk(qh, oracle, seed, replay, wide=False):
    rng = np.random.default_rng(seed)
    nseg = 30 if wide else 45
    seglen = [int(rng.integers(1, 6)) for _ in range(nseg)]
    if wide:
        for k in rng.choice(nseg, 5, replace=False):
            seglen[int(k)] = int(rng.integers(70, 91))
    nblk = sum(seglen)
    x = synth.make_input_numpy(NCH, nblk * 1024)
    x[1] = synth.make_mode_input_numpy("am", 1, nblk * 1024)
    e = qh.RxaEngine(NCH)
    e.load_emnr_tables()
    if replay:
        e.set_graph_replay(True)
        e.enable_meters(True)
        dev = torch.device("cuda:0")
        d_in = torch.zeros((NCH, 1024), dtype=torch.complex128, device=dev)
        d_out = torch.zeros((NCH, 256), dtype=torch.complex128, device=dev)
    os_ = [oracle.WdspChannel(1024, 256, 192000, 48000, 48000) for _ in range(NCH)]
    for c in range(NCH):
        for t, lead in ((e, (c,)), (os_[c], ())):
            t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, synth.shift_freq(c)); t.RXANBPSetRun(*lead, 1)
            t.SetRXAMode(*lead, (1, 6, 0, 1)[c]); t.RXASetPassband(*lead, *((300.0, 3000.0), (-4000.0, 4000.0), (-3000.0, -300.0), (300.0, 3000.0))[c])
            t.SetRXAAGCMode(*lead, (0, 3, 4, 2)[c])
    ys, rs = [], [[] for _ in range(NCH)]
    pos = 0
    log = []
    lms_used = [False] * NCH
    import os
    twin_c = int(os.environ["QH_TWIN"]) if os.environ.get("QH_TWIN") else None       # diagnostics: that channel's restatement once more, fed 1e-13 of noise
    if twin_c is not None:
        twin = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        c = twin_c
        twin.SetRXAShiftRun(1); twin.SetRXAShiftFreq(synth.shift_freq(c)); twin.RXANBPSetRun(1)
        twin.SetRXAMode((1, 6, 0, 1)[c]); twin.RXASetPassband(*((300.0, 3000.0), (-4000.0, 4000.0), (-3000.0, -300.0), (300.0, 3000.0))[c])
        twin.SetRXAAGCMode((0, 3, 4, 2)[c])
        tws, pert = [], np.random.default_rng(11)
    skip = set(filter(None, os.environ.get("QH_SKIP", "").split(",")))         # diagnostics: these setters are left out on every side

    class _Skipping:
        def __init__(self, t): self._t = t
        def __getattr__(self, name):
            return (lambda *a: None) if name in skip else getattr(self._t, name)
    e_set, os_set = (_Skipping(e), [_Skipping(o) for o in os_]) if skip else (e, os_)
    # a minimum-phase filter of 4096 - 16384 taps: mp_imp's cepstrum (fir.c:319-368) takes the logarithm of a stop band 200 dB down over a
    # 16 nc-point transform -- two transforms that differ in their last bits leave designs 1e-6 apart (seen: 1.4e-6 at 16384 taps, 1.5e-6 at 4096)
    mp_now, nc_now, mp_long = [0] * NCH, [2048] * NCH, [False] * NCH
    notches2 = [[0] for _ in range(NCH)]
    for s, n in enumerate(seglen):
        if s:
            for _ in range(int(rng.integers(1, 3))):
                c = int(rng.integers(0, NCH))
                tg = [(e_set, (c,)), (os_set[c], ())] + ([(twin, ())] if twin_c == c else [])
                if seed >= 5000 and rng.integers(0, 2):            # (walks from 5000 up: the second menu too, _apply2 below)
                    log.append((s, c, _apply2(rng, tg, notches2[c], fm=False)))
                else:
                    log.append((s, c, _apply(rng, tg, wide)))
                    notches2[c][0] += sum(1 for d in log[-1][2] if d[0] == "RXANBPAddNotch")
                lms_used[c] = lms_used[c] or any(d[0] in ("SetRXAANFRun", "SetRXAANRRun") and d[1] for d in log[-1][2])
                for d in log[-1][2]:
                    if d[0] == "RXASetMP": mp_now[c] = d[1]
                    if d[0] == "RXASetNC": nc_now[c] = d[1]
                mp_long[c] = mp_long[c] or (mp_now[c] and nc_now[c] >= 4096)
        seg = x[:, pos * 1024:(pos + n) * 1024]
        if replay:
            yb = np.empty((NCH, n * 256), dtype=np.complex128)
            for b in range(n):
                d_in.copy_(torch.from_numpy(seg[:, b * 1024:(b + 1) * 1024]))
                torch.cuda.synchronize()
                e.process_ptr(d_in.data_ptr(), 1024, d_out.data_ptr(), 256, 1)
                e.synchronize()
                yb[:, b * 256:(b + 1) * 256] = d_out.cpu().numpy()
            ys.append(yb)
        else:
            ys.append(e.process_host(seg))
        for c in range(NCH):
            rs[c].append(os_[c].xrxa(seg[c]))
        if twin_c is not None:
            # (QH_TWIN_EPS: the relative perturbation; QH_TWIN_ADD: an absolute one -- a floor of rounding noise where the input is exactly 0)
            tws.append(twin.xrxa(seg[twin_c] * (1.0 + float(os.environ.get("QH_TWIN_EPS", "1e-13")) * pert.standard_normal(seg.shape[1]))
                                 + float(os.environ.get("QH_TWIN_ADD", "0")) * (pert.standard_normal(seg.shape[1]) + 1j * pert.standard_normal(seg.shape[1]))))
        pos += n
    y = np.concatenate(ys, axis=1)
    if twin_c is not None:
        print("seed %d channel %d: engine %.3e, the restatement against its twin %.3e" % (seed, twin_c, rel_rms(y[twin_c], np.concatenate(rs[twin_c])), rel_rms(np.concatenate(tws), np.concatenate(rs[twin_c]))), flush=True)
    if replay:
        assert e.graph_launches() > nblk // 3
    for c in range(NCH):
        ref = np.concatenate(rs[c])
        assert np.all(np.isfinite(ref))
        if np.abs(ref).max() < 1e-9:
            continue                     # a squelch switched on during the start-up keeps the channel at 1e-18 for the whole walk: 0 / 0
        err = rel_rms(y[c], ref)
        tol = 1e-4 if lms_used[c] else 1e-5 if mp_long[c] else 1e-6
        if np.sqrt(np.mean(np.abs(y[c] - ref) ** 2)) < 1e-12 * max(1.0, np.abs(x[c]).max()):
            continue                     # muted from the first blocks on (SetRXAPanelSelect 0, walk 960097): what is left of the start-up sits at 1e-7
                                         # of the input, and rounding at 1e-13 of the INPUT is 2e-6 of that
        if os.environ.get("QH_REPORT"):      # diagnostics: every channel's figure, not only the first one over its tolerance
            print("seed %d channel %d: rel rms %.3e, tolerance %.0e" % (seed, c, err, tol), flush=True)
            if err >= tol:
                q0, per = 0, []
                for s2, n2 in enumerate(seglen):
                    a, b = q0 * 256, (q0 + n2) * 256
                    per.append("%d:%.1e/%.1e" % (s2, np.sqrt(np.mean(np.abs(y[c, a:b] - ref[a:b]) ** 2)), np.sqrt(np.mean(np.abs(ref[a:b]) ** 2))))
                    q0 += n2
                print("   per segment rms error / rms of the reference: %r\n   setters: %r" % (per, [l for l in log if l[1] == c]), flush=True)
        if err >= tol:
            # first segment that is off, for the failure message
            p0 = 0
            for s, n in enumerate(seglen):
                a, b = p0 * 256, (p0 + n) * 256
                if np.abs(y[c, a:b] - ref[a:b]).max() > tol * np.abs(ref).max():
                    per, q0 = [], 0
                    for s2, n2 in enumerate(seglen):
                        per.append("%d:%.1e" % (s2, np.abs(y[c, q0 * 256:(q0 + n2) * 256] - ref[q0 * 256:(q0 + n2) * 256]).max() / np.abs(ref).max()))
                        q0 += n2
                    raise AssertionError("seed %d channel %d: rel rms %.3e, first bad segment %d; setters so far %r; per-segment max error over the largest reference sample %r" %
                                         (seed, c, err, s, [l for l in log if l[0] <= s and l[1] == c], per))
                p0 += n
        assert err < tol


class _Without:
    """the object with one setter left out"""
    def __init__(self, t, name): self._t, self._name = t, name
    def __getattr__(self, name):
        return (lambda *a: None) if name == self._name else getattr(self._t, name)


def _apply2(rng, targets, notches, fm):
    """The setters _apply leaves out: the notch database's edits and its filter's window / auto-increase / edges / shift, the LMS filters'
    sizes and constants, bp1's run flag, the second panel gain -- and, on the one channel that may (fm), the FM detector: mode 5 in and
    out in mid-stream (its filters primed by then: the pull-in is then well-conditioned), deviation, CTCSS notch,
    the limiter.  notches: how many notches the channel's database holds (a list of one int, kept by the caller)."""
    k = int(rng.integers(-1, 14 if fm else 10))
    done = []

    def call(name, *args):
        done.append((name,) + args)
        for t, lead in targets:
            getattr(t, name)(*lead, *args)
    if k == -1:                                        # EMNR's tuning constants (emnr.c:1145-1175), around their defaults
        which = int(rng.integers(0, 4))
        if which == 0: call("SetRXAEMNRaeZetaThresh", float(rng.choice([0.6, 0.75, 0.9])))
        elif which == 1: call("SetRXAEMNRaePsi", float(rng.choice([5.0, 10.0, 20.0])))
        elif which == 2: call("SetRXAEMNRtrainZetaThresh", float(rng.choice([-0.7, -0.5, -0.2])))
        else: call("SetRXAEMNRtrainT2", float(rng.choice([0.1, 0.2, 0.4])))
    elif k == 0 and notches[0] > 0:
        call("RXANBPDeleteNotch", int(rng.integers(0, notches[0]))); notches[0] -= 1
    elif k == 1 and notches[0] > 0:
        call("RXANBPEditNotch", int(rng.integers(0, notches[0])), float(rng.uniform(-3000, 3000)), float(rng.uniform(50, 400)), int(rng.integers(0, 2)))
    elif k <= 2:
        call("RXANBPAddNotch", notches[0], float(rng.uniform(-3000, 3000)), float(rng.uniform(50, 400)), int(rng.integers(0, 2))); notches[0] += 1
    elif k == 3:
        call("RXANBPSetWindow", int(rng.integers(0, 2)))
    elif k == 4:
        call("RXANBPSetAutoIncrease", int(rng.integers(0, 2)))
    elif k == 5:
        lo = float(rng.uniform(-4000, 2000))
        call("RXANBPSetFreqs", lo, lo + float(rng.uniform(300, 4000)))
    elif k == 6:
        call("RXANBPSetShiftFrequency", float(rng.uniform(-2000, 2000)))
    elif k == 7:
        which = "SetRXAANFVals" if rng.integers(0, 2) else "SetRXAANRVals"
        call(which, int(rng.choice([16, 32, 64])), int(rng.choice([8, 16, 50])), float(rng.choice([1e-4, 2e-4, 1e-3])), float(rng.choice([0.1, 1e-3, 1e-2])))      # (the engine: up to 64 taps, one per lane)
    elif k == 8:
        call("SetRXABandpassRun", int(rng.integers(0, 2)))
    elif k == 9:
        call("SetRXAPanelGain2", float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)))
    elif k == 10:
        call("SetRXAMode", int(rng.choice([5, 5, 1])))
    elif k == 11:
        call("SetRXAFMDeviation", float(rng.choice([2500.0, 5000.0])))
    elif k == 12:
        call("SetRXACTCSSFreq", float(rng.choice([67.0, 100.0, 151.4, 250.3]))); call("SetRXACTCSSRun", int(rng.integers(0, 2)))
    else:
        call("SetRXAFMLimRun", int(rng.integers(0, 2))); call("SetRXAFMLimGain", float(rng.uniform(0.0, 20.0)))
    return done


@pytest.mark.parametrize("seed", list(range(301, 313)))
def test_random_setter_walk_with_the_notch_database_the_lms_sizes_and_fm(qh, oracle, seed):
    """The walks with the second menu mixed in (_apply2); channel 3 alone may become an FM channel (one FM filter length per engine)."""
    rng = np.random.default_rng(seed)
    nseg = 40
    seglen = [int(rng.integers(2, 9)) for _ in range(nseg)]
    nblk = sum(seglen)
    x = synth.make_input_numpy(NCH, nblk * 1024)
    x[1] = synth.make_mode_input_numpy("am", 1, nblk * 1024)
    x[3] = synth.make_mode_input_numpy("fm", 3, nblk * 1024)
    e = qh.RxaEngine(NCH)
    e.load_emnr_tables()
    e.enable_meters(True)
    os_ = [oracle.WdspChannel(1024, 256, 192000, 48000, 48000) for _ in range(NCH)]
    for c in range(NCH):
        for t, lead in ((e, (c,)), (os_[c], ())):
            t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, synth.shift_freq(c)); t.RXANBPSetRun(*lead, 1)
            t.SetRXAMode(*lead, (1, 6, 0, 1)[c]); t.RXASetPassband(*lead, *((300.0, 3000.0), (-4000.0, 4000.0), (-3000.0, -300.0), (-8000.0, 8000.0))[c])
            t.SetRXAAGCMode(*lead, (0, 3, 4, 0)[c])
    notches = [[0] for _ in range(NCH)]
    ys, rs, pos, log, lms_used, fm_used = [], [[] for _ in range(NCH)], 0, [], [False] * NCH, False
    import os
    skip = set(filter(None, os.environ.get("QH_SKIP", "").split(",")))         # diagnostics: these setters are left out on every side

    class _Skipping:
        def __init__(self, t): self._t = t
        def __getattr__(self, name):
            return (lambda *a: None) if name in skip else getattr(self._t, name)
    if skip:
        e_run, e = e, _Skipping(e)
        os_run, os_ = os_, [_Skipping(o) for o in os_]
    else:
        e_run, os_run = e, os_
    twin_c = int(os.environ["QH_TWIN"]) if os.environ.get("QH_TWIN") else None       # diagnostics: that channel's restatement once more, fed 1e-13 of noise
    if twin_c is not None:
        c = twin_c
        twin = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        twin.SetRXAShiftRun(1); twin.SetRXAShiftFreq(synth.shift_freq(c)); twin.RXANBPSetRun(1)
        twin.SetRXAMode((1, 6, 0, 1)[c]); twin.RXASetPassband(*((300.0, 3000.0), (-4000.0, 4000.0), (-3000.0, -300.0), (-8000.0, 8000.0))[c])
        twin.SetRXAAGCMode((0, 3, 4, 0)[c])
        tws, pert = [], np.random.default_rng(11)
    for s, n in enumerate(seglen):
        if s > 3:                                     # (every filter holds signal by then)
            for _ in range(int(rng.integers(1, 3))):
                c = int(rng.integers(0, NCH))
                tg = [(e, (c,)), (os_[c], ())] + ([(twin, ())] if twin_c == c else [])
                if c == 3:                          # (the AM detector forced on beside the FM one: the engine refuses that pair)
                    tg = [(_Without(t, "SetRXAAMDRun"), lead) for t, lead in tg]
                if rng.integers(0, 2):
                    d = _apply2(rng, tg, notches[c], fm=(c == 3))
                else:
                    d = _apply(rng, tg)
                    notches[c][0] += sum(1 for x_ in d if x_[0] == "RXANBPAddNotch")
                if c == 3 and any(x_[0] == "SetRXAMode" for x_ in d):
                    fm_used = fm_used or any(x_[0] == "SetRXAMode" and x_[1] == 5 for x_ in d)
                log.append((s, c, d))
                lms_used[c] = lms_used[c] or any(x_[0] in ("SetRXAANFRun", "SetRXAANRRun") and x_[1] for x_ in d)
        seg = x[:, pos * 1024:(pos + n) * 1024]
        ys.append(e_run.process_host(seg))
        for c in range(NCH):
            rs[c].append(os_run[c].xrxa(seg[c]))
        if seed % 2 == 0:                              # every other walk with WDSP's meters read after every segment (meter.c:75-130: mlog10's steps)
            for c in range(NCH):
                if np.abs(rs[c][-1]).max() < 1e-9 or lms_used[c]:
                    continue
                for mt in (0, 1, 2, 3, 5, 6):
                    got_m, ref_m = e_run.GetRXAMeter(c, mt), os_run[c].GetRXAMeter(mt)
                    assert abs(got_m - ref_m) < 0.0045, "seed %d segment %d channel %d meter %d: %.4f against %.4f; setters %r" % (
                        seed, s, c, mt, got_m, ref_m, [l for l in log if l[1] == c])
        if twin_c is not None:
            tws.append(twin.xrxa(seg[twin_c] * (1.0 + 1e-13 * pert.standard_normal(seg.shape[1]))))
            print("segment %d: engine %.2e, twin %.2e of %.3e   %r" % (s, np.abs(ys[-1][twin_c] - rs[twin_c][-1]).max(), np.abs(tws[-1] - rs[twin_c][-1]).max(), np.abs(rs[twin_c][-1]).max(),
                                                                         [(l[1], l[2]) for l in log if l[0] == s]), flush=True)
        pos += n
    y = np.concatenate(ys, axis=1)
    if twin_c is not None:
        print("seed %d channel %d: engine %.3e, the restatement against its twin %.3e" % (seed, twin_c, rel_rms(y[twin_c], np.concatenate(rs[twin_c])), rel_rms(np.concatenate(tws), np.concatenate(rs[twin_c]))), flush=True)
    for c in range(NCH):
        ref = np.concatenate(rs[c])
        assert np.all(np.isfinite(ref))
        if np.abs(ref).max() < 1e-9:
            continue
        err = rel_rms(y[c], ref)
        tol = 1e-4 if lms_used[c] else 1e-6
        assert err < tol, "seed %d channel %d: rel rms %.3e; setters %r" % (seed, c, err, [l for l in log if l[1] == c])
