"""AM / SAM / FM demodulator chains on the GPU against the CPU oracle, and a mixed-mode engine in the shape of
BASELINE config 4 (mode by channel mod 3).  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu
AM, FM, SAM, USB = 6, 5, 10, 1
TOL = 1e-8          # north_star gate: 1e-6 relative RMS (float64)


def _cfg_oracle(po, c, mode, **kw):
    ch = po.WdspChannel(1024, 256, 192000, 48000, 48000)
    ch.SetRXAShiftRun(1)
    ch.SetRXAShiftFreq(synth.shift_freq(c))
    ch.RXANBPSetRun(1)
    ch.SetRXAMode(mode)
    ch.SetRXAAGCMode(0)
    ch.SetRXAAGCFixed(0.0)
    _apply(lambda name, *a: getattr(ch, name)(*a), mode, kw)
    return ch


def _apply(call, mode, kw):
    if mode == USB:
        call("RXASetPassband", 300.0, 3000.0)
    elif mode in (AM, SAM):
        call("RXASetPassband", -4000.0, 4000.0)
    elif mode == FM:
        call("RXASetPassband", -8000.0, 8000.0)
    if "sbmode" in kw:
        call("SetRXAAMDSBMode", kw["sbmode"])
    if "levelfade" in kw:
        call("SetRXAAMDFadeLevel", kw["levelfade"])
    if "deviation" in kw:
        call("SetRXAFMDeviation", kw["deviation"])
    if "ctcss_run" in kw:
        call("SetRXACTCSSRun", kw["ctcss_run"])
    if "ctcss_freq" in kw:
        call("SetRXACTCSSFreq", kw["ctcss_freq"])


def _cfg_engine(e, c, mode, **kw):
    e.SetRXAShiftRun(c, 1)
    e.SetRXAShiftFreq(c, synth.shift_freq(c))
    e.RXANBPSetRun(c, 1)
    e.SetRXAMode(c, mode)
    e.SetRXAAGCMode(c, 0)
    e.SetRXAAGCFixed(c, 0.0)
    _apply(lambda name, *a: getattr(e, name)(c, *a), mode, kw)


# PLL modes (SAM, FM): while the filters fill up the detector input is FFT round-off noise (1e-28), and atan2 is
# scale invariant, so the loop's trajectory during acquisition depends on the last bit of whichever FFT is in
# use -- in the reference as much as here (a 1e-15 relative change of the INPUT moves the oracle's own SAM output
# by 1e-2 in the first blocks).  Differences then decay with the loop / dc-removal / fade-leveler time constants
# (FM 0.02 s, SAM fade leveler 1.4 s), so these modes are compared after `settle` DSP blocks.
@pytest.mark.parametrize("mode,sig,kw,nblk,settle", [
    (AM, "am", {}, 48, 0), (AM, "am", {"levelfade": 0}, 48, 0),
    (SAM, "am", {"levelfade": 0}, 160, 100), (SAM, "am", {"sbmode": 1, "levelfade": 0}, 160, 100),
    (SAM, "am", {"sbmode": 2, "levelfade": 0}, 160, 100), (SAM, "am", {}, 4000, 3900),
    (FM, "fm", {}, 192, 144), (FM, "fm", {"deviation": 2500.0, "ctcss_run": 0}, 192, 144),
    (FM, "fm", {"ctcss_freq": 100.0}, 192, 144),
])
def test_single_mode_chain(qh, oracle, mode, sig, kw, nblk, settle):
    x = synth.make_mode_input_numpy(sig, 0, nblk * 1024)
    e = qh.RxaEngine(1)
    _cfg_engine(e, 0, mode, **kw)
    # three calls of uneven length (state carried through the scans and the PLL)
    y = np.concatenate([e.process_host(x[None, :5 * 1024]), e.process_host(x[None, 5 * 1024:6 * 1024]),
                        e.process_host(x[None, 6 * 1024:])], axis=1)[0]
    ref = _cfg_oracle(oracle, 0, mode, **kw).xrxa(x)
    err = rel_rms(y[settle * 256:], ref[settle * 256:])
    assert err < (TOL if settle == 0 else 1e-6), err
    # the demodulated 1 kHz tone is really there (not a trivially small output)
    assert np.abs(ref[-4096:]).max() > 1e-3


def test_mixed_modes_config4_shape(qh, oracle):
    """12 channels, mode by c mod 3: USB / AM / FM (BASELINE config 4), two calls."""
    nch, nblk = 12, 192
    modes = [(USB, "usb"), (AM, "am"), (FM, "fm")]
    x = np.stack([synth.make_mode_input_numpy(modes[c % 3][1], c, nblk * 1024) for c in range(nch)])
    e = qh.RxaEngine(nch)
    for c in range(nch):
        _cfg_engine(e, c, modes[c % 3][0])
    y = np.concatenate([e.process_host(x[:, :7 * 1024]), e.process_host(x[:, 7 * 1024:])], axis=1)
    for c in range(nch):
        ref = _cfg_oracle(oracle, c, modes[c % 3][0]).xrxa(x[c])
        if modes[c % 3][0] == FM:       # after lock (see the note above)
            err = rel_rms(y[c][144 * 256:], ref[144 * 256:])
            assert err < 1e-6, (c, err)
        else:
            err = rel_rms(y[c], ref)
            assert err < TOL, (c, err)


def test_mode_switch_mid_stream(qh, oracle):
    """USB -> AM -> USB on a running channel: bp1 is flushed when it starts (RXA.c:825) and its gain doubles."""
    x = synth.make_mode_input_numpy("am", 0, 30 * 1024)
    e = qh.RxaEngine(1)
    _cfg_engine(e, 0, USB)
    o = _cfg_oracle(oracle, 0, USB)
    ys, rs = [], []
    for k, mode in enumerate((USB, AM, USB)):
        e.SetRXAMode(0, mode); o.SetRXAMode(mode)
        seg = x[k * 10240:(k + 1) * 10240]
        ys.append(e.process_host(seg[None, :])[0]); rs.append(o.xrxa(seg))
    assert rel_rms(np.concatenate(ys), np.concatenate(rs)) < TOL


def test_fm_detector_limiter(qh, oracle):
    """SetRXAFMLimRun / SetRXAFMLimGain (wdsp/fmd.c:179-184,336-362): pre-gain 0.4 and a wcpAGC of its own on the FM audio."""
    from quisk_amd import synth
    nch, nblk = 2, 200
    x = np.stack([synth.make_mode_input_numpy("fm", c, nblk * 1024) for c in range(nch)])
    e = qh.RxaEngine(nch, dsp_size=256, in_rate=192000, dsp_rate=48000, out_rate=48000)
    e.SetRXAShiftRun(-1, 1); e.RXANBPSetRun(-1, 1); e.SetRXAMode(-1, 5); e.RXASetPassband(-1, -8000.0, 8000.0)
    e.SetRXAFMLimRun(0, 1); e.SetRXAFMLimGain(0, 6.0)                   # channel 0 only; channel 1 stays unlimited
    refs = []
    for c in range(nch):
        e.SetRXAShiftFreq(c, synth.shift_freq(c))
        ch = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        ch.SetRXAShiftRun(1); ch.SetRXAShiftFreq(synth.shift_freq(c)); ch.RXANBPSetRun(1); ch.SetRXAMode(5)
        ch.RXASetPassband(-8000.0, 8000.0)
        if c == 0:
            ch.SetRXAFMLimRun(1); ch.SetRXAFMLimGain(6.0)
        refs.append(ch.xrxa(x[c]))
    y = np.concatenate([e.process_host(x[:, :70 * 1024]), e.process_host(x[:, 70 * 1024:])], axis=1)
    lo = 150 * 256                                                      # after PLL / notch settling (DESIGN.md parity caveat)
    for c in range(nch):
        assert rel_rms(y[c][lo:], refs[c][lo:]) < 1e-6, (c, rel_rms(y[c][lo:], refs[c][lo:]))
    assert rel_rms(refs[0][lo:], refs[1][lo:] ) > 1e-2                 # the limiter does something


def test_fm_channel_leaves_and_returns_keeps_its_delay_lines(qh, oracle):
    """SetRXAMode only clears fmd's run flag (RXA.c:758-776): the de-emphasis and audio fircores keep their delay lines while
    the channel is in another mode.  Channel 0 goes FM -> USB for an ODD number of calls -> FM while channel 1 stays in FM (so
    the engine's ping-pong history pair keeps flipping): the first blocks after the return still hold the old samples."""
    nblk = 60
    x = np.stack([synth.make_mode_input_numpy("fm", c, 5 * nblk * 1024) for c in range(2)])
    e = qh.RxaEngine(2)
    os_ = []
    for c in range(2):
        _cfg_engine(e, c, FM)
        os_.append(_cfg_oracle(oracle, c, FM))
    plan = [(FM, 2 * nblk), (USB, 7), (USB, 5), (USB, 9), (FM, nblk)]          # 3 calls away
    pos, got, ref = 0, None, None
    for mode, nb in plan:
        e.SetRXAMode(0, mode); os_[0].SetRXAMode(mode)
        if mode == USB:
            e.RXASetPassband(0, 300.0, 3000.0); os_[0].RXASetPassband(300.0, 3000.0)
        else:
            e.RXASetPassband(0, -8000.0, 8000.0); os_[0].RXASetPassband(-8000.0, 8000.0)
        seg = np.ascontiguousarray(x[:, pos * 1024:(pos + nb) * 1024])
        pos += nb
        got = e.process_host(seg)
        ref = [os_[c].xrxa(seg[c]) for c in range(2)]
    # the last segment: channel 0 is back in FM; compare from its first sample on
    assert rel_rms(got[0], ref[0]) < 1e-6
    assert rel_rms(got[1][20 * 256:], ref[1][20 * 256:]) < 1e-6


def test_last_stage_writes_the_output_directly_and_equals_the_separate_output_pass(qh, oracle):
    """USB / AM / SAM / FM (one with its CTCSS notch off) in one call, fixed gain and panel gains that differ per channel: with no
    stage between a channel's last filter and the output matrix that filter's store applies the matrix (qh_engine.hip, `direct`).
    The same engine with meters on runs every channel through the separate output pass; a caller working in place does as well.
    All three agree (to the last bit where the operations are the same), and with the oracle."""
    import torch
    nch, nblk = 10, 160
    modes = [(USB, "usb"), (AM, "am"), (FM, "fm"), (SAM, "am"), (FM, "fm")]
    kws = [{}, {}, {}, {"levelfade": 0}, {"ctcss_run": 0}]   # SAM without the fade leveller's 1.4 s memory of the pull-in
    x = np.stack([synth.make_mode_input_numpy(modes[c % 5][1], c, nblk * 1024) for c in range(nch)])
    def make(meters):
        e = qh.RxaEngine(nch)
        for c in range(nch):
            _cfg_engine(e, c, modes[c % 5][0], **kws[c % 5])
            e.SetRXAAGCFixed(c, 3.0 * c - 6.0)
            e.SetRXAPanelGain1(c, 0.5 + 0.1 * c)
            e.SetRXAPanelGain2(c, 1.0 - 0.05 * c, 0.6 + 0.03 * c)
            e.SetRXAPanelCopy(c, c % 4)
        e.enable_meters(meters)
        return e
    def same(u, v):
        # the AM channels' nbp0 runs on tiles of 2048 outputs instead of 2049 when its store takes the envelope, and the real bp1
        # filter of an AM / SAM channel shares its transforms with a partner channel on the direct route (other rounding, same
        # arithmetic); every other channel is the same sequence of operations on either route
        for c in range(nch):
            if modes[c % 5][0] in (AM, SAM): assert rel_rms(u[c], v[c]) < 1e-12, c
            else: assert np.array_equal(u[c], v[c]), c
    ya = make(False).process_host(x)
    yb = make(True).process_host(x)
    same(ya, yb)
    e = make(False)                                      # in place: output rows over the input rows
    d = torch.from_numpy(x.view(np.float64).copy()).cuda()
    e.process_ptr(d.data_ptr(), nblk * 1024, d.data_ptr(), nblk * 1024, nblk)
    e.synchronize()
    yc = d.cpu().numpy().view(np.complex128)[:, :nblk * 256]
    same(ya, yc)
    for c in range(nch):
        o = _cfg_oracle(oracle, c, modes[c % 5][0], **kws[c % 5])
        o.SetRXAAGCFixed(3.0 * c - 6.0); o.SetRXAPanelGain1(0.5 + 0.1 * c); o.SetRXAPanelGain2(1.0 - 0.05 * c, 0.6 + 0.03 * c); o.SetRXAPanelCopy(c % 4)
        ref = o.xrxa(x[c])
        lo = 0 if modes[c % 5][0] in (USB, AM) else 120 * 256
        assert rel_rms(ya[c][lo:], ref[lo:]) < (TOL if lo == 0 else 1e-6), c


@pytest.mark.parametrize("am_bands", [[(-4000.0, 4000.0), (-4000.0, 4000.0), (-3000.0, 3000.0), (-4000.0, 4000.0)],
                                      [(-4000.0, 4000.0), (-4000.0, 3000.0), (-4000.0, 4000.0), (-4000.0, 4000.0)]],
                         ids=["real-bp1-two-designs", "one-complex-bp1"])
def test_partner_channels_of_the_real_filters(qh, oracle, am_bands):
    """The filters behind the detectors (bp1 of AM, FM de-emphasis) have real taps and a real input: two channels share a tile
    (osfir_kernel PAIR).  Three FM channels (one is its own partner), four AM channels of which one has another passband (its own
    partner) or an asymmetric one (complex taps: nobody is paired); the second AM channel is 80 dB under its partner, whose rounding
    noise it now shares.  Every channel against the oracle."""
    modes = [USB, FM, AM, FM, AM, FM, AM, AM]
    nch, nblk = len(modes), 192
    sig = {USB: "usb", FM: "fm", AM: "am"}
    x = np.stack([synth.make_mode_input_numpy(sig[m], c, nblk * 1024) for c, m in enumerate(modes)])
    ams = [c for c, m in enumerate(modes) if m == AM]
    x[ams[1]] *= 1e-4
    e = qh.RxaEngine(nch)
    refs = []
    for c, m in enumerate(modes):
        _cfg_engine(e, c, m)
        refs.append(_cfg_oracle(oracle, c, m))
    for k, c in enumerate(ams):
        e.RXASetPassband(c, *am_bands[k]); refs[c].RXASetPassband(*am_bands[k])
    y = np.concatenate([e.process_host(x[:, :50 * 1024]), e.process_host(x[:, 50 * 1024:])], axis=1)
    for c, m in enumerate(modes):
        ref = refs[c].xrxa(x[c])
        lo = 144 * 256 if m == FM else 0
        err = rel_rms(y[c][lo:], ref[lo:])
        assert err < (1e-6 if m == FM else TOL), (c, err)
