"""Parity of the HIP RXA chain (through the C ABI) against the CPU oracle.  -m gpu."""
import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-9      # north_star asks <= 1e-6 relative RMS in float64; the chain meets a far tighter bound


def _oracle_channel(po, c, in_rate=192000, dsp_size=256, nc=None, shift=True, nbp=True, passband=(300.0, 3000.0),
                    agc_db=0.0):
    ch = po.WdspChannel(dsp_size * (in_rate // 48000), dsp_size, in_rate, 48000, 48000)
    ch.SetRXAShiftRun(1 if shift else 0)
    if shift:
        ch.SetRXAShiftFreq(synth.shift_freq(c))
    ch.RXANBPSetRun(1 if nbp else 0)
    ch.SetRXAMode(1)
    ch.RXASetPassband(*passband)
    if nc:
        ch.RXASetNC(nc)
    ch.SetRXAAGCMode(0)
    ch.SetRXAAGCFixed(agc_db)
    return ch


def _engine(qh, nch, in_rate=192000, dsp_size=256, nc=None, shift=True, nbp=True, passband=(300.0, 3000.0), agc_db=0.0):
    e = qh.RxaEngine(nch, dsp_size=dsp_size, in_rate=in_rate, dsp_rate=48000, out_rate=48000)
    e.SetRXAShiftRun(-1, 1 if shift else 0)
    if shift:
        for c in range(nch):
            e.SetRXAShiftFreq(c, synth.shift_freq(c))
    e.RXANBPSetRun(-1, 1 if nbp else 0)
    e.SetRXAMode(-1, 1)
    e.RXASetPassband(-1, *passband)
    if nc:
        e.RXASetNC(-1, nc)
    e.SetRXAAGCMode(-1, 0)
    e.SetRXAAGCFixed(-1, agc_db)
    return e


def test_ssb_chain_c1_config(qh, oracle):
    """BASELINE config 1/2 shape: 192 k -> 48 k, shift + 561-tap resampler + NBP nc 2048, several channels,
    several calls of different length (state carried across calls)."""
    nch, nblk = 5, 40
    x = synth.make_input_numpy(nch, nblk * 1024)
    e = _engine(qh, nch)
    splits = [3, 1, 17, 19]
    outs, pos = [], 0
    for k in splits:
        outs.append(e.process_host(x[:, pos * 1024:(pos + k) * 1024]))
        pos += k
    y = np.concatenate(outs, axis=1)
    for c in range(nch):
        ref = _oracle_channel(oracle, c).xrxa(x[c])
        err = rel_rms(y[c], ref)
        assert err < TOL, (c, err)
    # the in-band tone comes out with the panel gain of 4
    assert abs(np.abs(y[0][-2000:]).mean() / 0.1 - 4.0) < 0.2


@pytest.mark.parametrize("meters", [False, True])
@pytest.mark.parametrize("in_rate", [192000, 48000])
def test_calls_around_one_delay_line_long(qh, oracle, in_rate, meters):
    """The front stage keeps 2240 input samples, a fircore stage 4095 DSP-rate samples; a call at least that long has the next call's delay
    line written by the tile kernels themselves (OsfirArgs::hist_next), a shorter one by hist_update_kernel.  Calls of 2 / 3 blocks (2048 /
    3072 input samples at 192 k) and 15 / 16 / 17 blocks (3840 / 4096 / 4352 DSP samples) in turn, with and without the meter taps (another
    kernel variant), at 192 k (the /4 front) and at 48 k (no resampler: the shift alone)."""
    nch = 3
    calls = [2, 3, 15, 16, 17, 1, 16, 15, 3, 2, 16, 40, 1, 15, 17]
    per = 256 * (in_rate // 48000)
    x = synth.make_input_numpy(nch, sum(calls) * per)
    e = _engine(qh, nch, in_rate=in_rate)
    e.enable_meters(meters)
    refs = [_oracle_channel(oracle, c, in_rate=in_rate) for c in range(nch)]
    pos = 0
    for k, nb in enumerate(calls):
        seg = np.ascontiguousarray(x[:, pos * per:(pos + nb) * per])
        pos += nb
        y = e.process_host(seg)
        for c in range(nch):
            ref = refs[c].xrxa(seg[c])
            assert rel_rms(y[c], ref) < TOL or np.abs(ref).max() < 1e-12, (k, nb, c, rel_rms(y[c], ref))
    e.close()


def test_single_block_calls(qh, oracle):
    """One DSP block per call (the drop-in's pattern): 64 calls."""
    nblk = 64
    x = synth.make_input_numpy(1, nblk * 1024, first_channel=3)
    e = _engine(qh, 1)
    e.SetRXAShiftFreq(0, synth.shift_freq(3))
    y = np.concatenate([e.process_host(x[:, b * 1024:(b + 1) * 1024]) for b in range(nblk)], axis=1)
    ref = _oracle_channel(oracle, 3).xrxa(x[0])
    assert rel_rms(y[0], ref) < TOL


def test_shift_and_nbp_off_dc_gain(qh, oracle):
    nblk = 12
    x = np.full((2, nblk * 1024), 0.25 - 0.125j)
    e = _engine(qh, 2, shift=False, nbp=False)
    y = e.process_host(x)
    ref = _oracle_channel(oracle, 0, shift=False, nbp=False).xrxa(x[0])
    assert rel_rms(y[0], ref) < TOL
    assert abs(y[1][-1] - (1.0 - 0.5j)) < 1e-6


@pytest.mark.parametrize("nc", [256, 1024])
def test_short_filters(qh, oracle, nc):
    nblk = 20
    x = synth.make_input_numpy(2, nblk * 1024)
    e = _engine(qh, 2, nc=nc, passband=(-3000.0, -300.0), agc_db=6.0)
    y = e.process_host(x)
    for c in range(2):
        ref = _oracle_channel(oracle, c, nc=nc, passband=(-3000.0, -300.0), agc_db=6.0).xrxa(x[c])
        assert rel_rms(y[c], ref) < TOL


@pytest.mark.parametrize("in_rate", [48000, 96000, 384000])
def test_other_rates(qh, oracle, in_rate):
    d = in_rate // 48000
    nblk = 16
    x = synth.make_input_numpy(2, nblk * 256 * d, fs=float(in_rate))
    e = _engine(qh, 2, in_rate=in_rate)
    y = e.process_host(x)
    for c in range(2):
        ref = _oracle_channel(oracle, c, in_rate=in_rate).xrxa(x[c])
        assert rel_rms(y[c], ref) < TOL


def test_parameter_change_between_calls(qh, oracle):
    """Passband, shift frequency and panel gain changed mid-stream; the oracle applies them at the same block."""
    x = synth.make_input_numpy(1, 30 * 1024)
    e = _engine(qh, 1)
    o = _oracle_channel(oracle, 0)
    y1 = e.process_host(x[:, :10 * 1024]); r1 = o.xrxa(x[0, :10 * 1024])
    e.RXASetPassband(0, 200.0, 2400.0); o.RXASetPassband(200.0, 2400.0)
    e.SetRXAShiftFreq(0, 10450.0); o.SetRXAShiftFreq(10450.0)
    y2 = e.process_host(x[:, 10 * 1024:20 * 1024]); r2 = o.xrxa(x[0, 10 * 1024:20 * 1024])
    e.SetRXAPanelGain1(0, 2.0); o.SetRXAPanelGain1(2.0)
    e.SetRXAPanelCopy(0, 1); o.SetRXAPanelCopy(1)
    y3 = e.process_host(x[:, 20 * 1024:]); r3 = o.xrxa(x[0, 20 * 1024:])
    y = np.concatenate([y1[0], y2[0], y3[0]])
    r = np.concatenate([r1, r2, r3])
    assert rel_rms(y, r) < TOL


def test_unsupported_requests_fail_loudly(qh):
    e = qh.RxaEngine(1)
    e.SetRXAAGCMode(0, 7)                       # maps to WDSP's mode 5 (wcpAGC.c:406-408), which RXA never runs
    x = np.zeros((1, 1024), dtype=np.complex128)
    with pytest.raises(qh.QuiskHipError):
        e.process_host(x)
    with pytest.raises(qh.QuiskHipError):
        e.RXASetNC(0, 131072)                   # nc up to 65536 (sixteen partitions of 4096 taps, tests/test_gpu_long_nc.py)
    with pytest.raises(qh.QuiskHipError):
        qh.RxaEngine(1, in_rate=44100)          # in_rate / dsp_rate must be whole one way or the other (wdsp/channel.c:39-42)


def _agc_signal(n, fs=192000.0):
    """In-band tone whose level steps 0.001 -> 0.3 -> 0.01 -> silence -> 0.1 (exercises attack, hang, decay)."""
    t = np.arange(n)
    env = np.full(n, 0.001)
    env[n // 6:2 * n // 6] = 0.3
    env[2 * n // 6:3 * n // 6] = 0.01
    env[3 * n // 6:4 * n // 6] = 0.0
    env[4 * n // 6:] = 0.1
    rng = np.random.default_rng(17)
    f1 = synth.channel_tones(0, fs)[0]
    return env * np.exp(2j * np.pi * ((f1 / fs) * t % 1.0)) + 1e-5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_agc_modes(qh, oracle, mode):
    """WDSP's AGC state machine (long / slow / medium / fast) after the SSB chain; medium is WDSP's default."""
    nblk = 240
    x = _agc_signal(nblk * 1024)
    e = _engine(qh, 1)
    o = _oracle_channel(oracle, 0)
    e.SetRXAAGCMode(0, mode); o.SetRXAAGCMode(mode)
    y = np.concatenate([e.process_host(x[None, :100 * 1024]), e.process_host(x[None, 100 * 1024:])], axis=1)[0]
    ref = o.xrxa(x)
    assert rel_rms(y, ref) < 1e-7, rel_rms(y, ref)
    # the AGC really acted: strong and weak segments come out within 12 dB of each other, not 30 dB apart
    strong = np.abs(ref[60 * 256:75 * 256]).mean()
    weak = np.abs(ref[110 * 256:118 * 256]).mean()
    if mode >= 3:                               # long / slow hold the gain down for seconds
        assert weak > 0.05 * strong


def test_agc_parameter_setters(qh, oracle):
    nblk = 120
    x = _agc_signal(nblk * 1024)
    e = _engine(qh, 1)
    o = _oracle_channel(oracle, 0)
    for obj, call in ((e, lambda n, *a: getattr(e, n)(0, *a)), (o, lambda n, *a: getattr(o, n)(*a))):
        call("SetRXAAGCMode", 2)
        call("SetRXAAGCAttack", 2)
        call("SetRXAAGCDecay", 300)
        call("SetRXAAGCHang", 100)
        call("SetRXAAGCTop", 90.0)
        call("SetRXAAGCSlope", 35)
        call("SetRXAAGCHangThreshold", 40)
    y = e.process_host(x[None, :])[0]
    assert rel_rms(y, o.xrxa(x)) < 1e-7


@pytest.mark.parametrize("mode", [1, 5])
def test_minimum_phase_filters(qh, oracle, mode):
    """RXASetMP(1) (wdsp/RXA.c:948-958): every fircore impulse goes through mp_imp (wdsp/fir.c:319-368); switched
    on after a few blocks like a GUI would, off again later."""
    nch, nblk = 2, 60 if mode == 1 else 200
    x = np.stack([synth.make_mode_input_numpy("fm" if mode == 5 else "usb", c, nblk * 1024) for c in range(nch)])
    e = _engine(qh, nch)
    e.SetRXAMode(-1, mode)
    chans = []
    for c in range(nch):
        ch = _oracle_channel(oracle, c)
        ch.SetRXAMode(mode)
        chans.append(ch)
    outs, refs = [], [[] for _ in range(nch)]
    for (a, b), mp in zip(((0, 4), (4, 2 * nblk // 3), (2 * nblk // 3, nblk)), (0, 1, 0)):
        e.RXASetMP(-1, mp)
        outs.append(e.process_host(x[:, a * 1024:b * 1024]))
        for c in range(nch):
            chans[c].RXASetMP(mp)
            refs[c].append(chans[c].xrxa(x[c, a * 1024:b * 1024]))
    y = np.concatenate(outs, axis=1)
    for c in range(nch):
        ref = np.concatenate(refs[c])
        if mode == 5:       # PLL acquisition rings in the CTCSS notch (pole radius 0.9994): compare after it died
            assert rel_rms(y[c][100 * 256:], ref[100 * 256:]) < 1e-6, (c, rel_rms(y[c][100 * 256:], ref[100 * 256:]))
        else:
            assert rel_rms(y[c], ref) < 1e-9, (c, rel_rms(y[c], ref))


@pytest.mark.parametrize("out_rate", [96000, 24000, 12000])
def test_output_resampler(qh, oracle, out_rate):
    """out_rate != dsp_rate: xresample on the way out (wdsp/RXA.c:596, RXAResCheck :789-798)."""
    nch, nblk = 2, 24
    x = synth.make_input_numpy(nch, nblk * 1024)
    e = qh.RxaEngine(nch, dsp_size=256, in_rate=192000, dsp_rate=48000, out_rate=out_rate)
    e.SetRXAShiftRun(-1, 1); e.RXANBPSetRun(-1, 1); e.SetRXAMode(-1, 1); e.RXASetPassband(-1, 300.0, 3000.0)
    e.SetRXAAGCMode(-1, 0); e.SetRXAAGCFixed(-1, 0.0)
    assert e.dsp_outsize == 256 * out_rate // 48000
    for c in range(nch):
        e.SetRXAShiftFreq(c, synth.shift_freq(c))
    y = np.concatenate([e.process_host(x[:, :5 * 1024]), e.process_host(x[:, 5 * 1024:])], axis=1)
    for c in range(nch):
        ch = oracle.WdspChannel(1024, 256, 192000, 48000, out_rate)
        ch.SetRXAShiftRun(1); ch.SetRXAShiftFreq(synth.shift_freq(c)); ch.RXANBPSetRun(1); ch.SetRXAMode(1)
        ch.RXASetPassband(300.0, 3000.0); ch.SetRXAAGCMode(0); ch.SetRXAAGCFixed(0.0)
        ref = ch.xrxa(x[c])
        assert y.shape[1] == ref.size
        assert rel_rms(y[c], ref) < TOL, (c, rel_rms(y[c], ref))


def test_notch_database(qh, oracle):
    """The notched band-pass (wdsp/nbp.c:64-239, 358-525): notches added, edited, deleted, the VFO moved under them."""
    nch, nblk = 2, 40
    x = synth.make_input_numpy(nch, nblk * 1024)
    e = _engine(qh, nch)
    chans = [_oracle_channel(oracle, c) for c in range(nch)]

    def both(name, *args):
        r = getattr(e, name)(-1, *args)
        rs = [getattr(ch, name)(*args) for ch in chans]
        return r, rs

    steps = [
        lambda: both("RXANBPSetTuneFrequency", 7100000.0),
        lambda: both("RXANBPAddNotch", 0, 7101000.0, 300.0, 1),          # 1000 +- 150 Hz in baseband
        lambda: both("RXANBPAddNotch", 1, 7102500.0, 50.0, 1),           # narrow: widened to the minimum width
        lambda: both("RXANBPSetNotchesRun", 1),
        lambda: both("RXANBPSetTuneFrequency", 7100400.0),                # notches slide by 400 Hz
        lambda: both("RXANBPEditNotch", 0, 7101900.0, 800.0, 1),
        lambda: both("RXANBPSetAutoIncrease", 0),
        lambda: both("RXANBPAddNotch", 0, 7100500.0, 900.0, 1),          # overlaps the lower passband edge
        lambda: both("RXANBPDeleteNotch", 1),
        lambda: both("RXANBPSetWindow", 1),
        lambda: both("RXANBPSetNotchesRun", 0),
    ]
    outs, refs = [], [[] for _ in range(nch)]
    per = 3
    for k, st in enumerate(steps):
        r, rs = st()
        if r is not None:
            assert all(r == v for v in rs)                                # Add / Edit / Delete return 0 like the reference
        a, b = k * per * 1024, (k + 1) * per * 1024
        outs.append(e.process_host(x[:, a:b]))
        for c in range(nch):
            refs[c].append(chans[c].xrxa(x[c, a:b]))
    assert e.RXANBPAddNotch(0, 99, 1.0, 1.0, 1) == -1 and e.RXANBPDeleteNotch(0, 99) == -1
    assert e.RXANBPGetNumNotches(0) == 2 and e.RXANBPGetNotch(0, 0)[0] == 0 and e.RXANBPGetNotch(0, 5)[0] == -1
    assert e.RXANBPGetMinNotchWidth(0) == 2200.0 / 8
    y = np.concatenate(outs, axis=1)
    for c in range(nch):
        ref = np.concatenate(refs[c])
        assert np.abs(ref).max() > 1e-3
        assert rel_rms(y[c], ref) < TOL, (c, rel_rms(y[c], ref))


def test_input_rate_768k(qh, oracle):
    """in_rate / dsp_rate = 16: 2241-tap resampler, spectral fold by 8 then every second sample."""
    nch, nblk, fs = 2, 12, 768000
    rng = np.random.default_rng(4)
    n = nblk * 256 * 16
    t = np.arange(n)
    x = np.stack([0.2 * np.exp(2j * np.pi * ((-(20000.0 + 500 * c) - 1000.0) / fs * t % 1.0)) +
                  0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) for c in range(nch)])
    e = qh.RxaEngine(nch, dsp_size=256, in_rate=fs, dsp_rate=48000, out_rate=48000)
    assert e.dsp_insize == 4096
    e.SetRXAShiftRun(-1, 1); e.RXANBPSetRun(-1, 1); e.SetRXAMode(-1, 1); e.RXASetPassband(-1, 300.0, 3000.0)
    e.SetRXAAGCMode(-1, 0); e.SetRXAAGCFixed(-1, 0.0)
    for c in range(nch):
        e.SetRXAShiftFreq(c, 20000.0 + 500 * c)
    y = np.concatenate([e.process_host(x[:, :5 * 4096]), e.process_host(x[:, 5 * 4096:])], axis=1)
    for c in range(nch):
        ch = oracle.WdspChannel(4096, 256, fs, 48000, 48000)
        ch.SetRXAShiftRun(1); ch.SetRXAShiftFreq(20000.0 + 500 * c); ch.RXANBPSetRun(1); ch.SetRXAMode(1)
        ch.RXASetPassband(300.0, 3000.0); ch.SetRXAAGCMode(0); ch.SetRXAAGCFixed(0.0)
        ref = ch.xrxa(x[c])
        assert np.abs(ref).max() > 0.1
        assert rel_rms(y[c], ref) < TOL, (c, rel_rms(y[c], ref))


@pytest.mark.parametrize("which,position,mode", [("ANF", 0, 1), ("ANR", 0, 1), ("ANF", 1, 1), ("ANR", 1, 0), ("BOTH", 0, 6), ("BOTH", 1, 1)])
def test_lms_notch_and_noise_reduction(qh, oracle, which, position, mode):
    """xanf / xanr (wdsp/anf.c:82-133, anr.c:82-133) in both chain positions, with bp1 (gain 2) following them.  The LMS sums
    its 64 taps in a different order than the reference's loop and its step-size logic compares nearly equal numbers,
    so the gate is the fp64 tolerance 1e-6 of the chain, not bit-exactness."""
    nch, nblk = 3, 60
    x = synth.make_input_numpy(nch, nblk * 1024)
    e = qh.RxaEngine(nch)
    refs = []
    for ch in range(nch):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        for t in (e, o):
            args = (ch,) if t is e else ()
            t.SetRXAShiftRun(*args, 1); t.SetRXAShiftFreq(*args, synth.shift_freq(ch)); t.RXANBPSetRun(*args, 1)
            t.SetRXAMode(*args, mode)
            t.RXASetPassband(*args, *((-3000.0, -300.0) if mode == 0 else (300.0, 3000.0) if mode == 1 else (-4000.0, 4000.0)))
            # channel 0: fixed gain (the position-1 filters must see it applied), channel 1: AGC fast, channel 2: AGC off-ish long
            t.SetRXAAGCMode(*args, (0, 4, 1)[ch])
            if ch == 0:
                t.SetRXAAGCFixed(*args, 12.0)
            if which in ("ANF", "BOTH"):
                t.SetRXAANFPosition(*args, position); t.SetRXAANFRun(*args, 1)
            if which in ("ANR", "BOTH"):
                t.SetRXAANRPosition(*args, position); t.SetRXAANRVals(*args, 48, 20, 2e-4, 0.05); t.SetRXAANRRun(*args, 1)
        refs.append(o)
    y = e.process_host(x)
    ref = np.stack([o.xrxa(x[ch]) for ch, o in enumerate(refs)])
    assert np.abs(ref).max() > 1e-3
    for ch in range(nch):
        assert rel_rms(y[ch], ref[ch]) < 1e-6
    # switching the filter off and on again flushes its delay line and weights but keeps the step-size state
    for ch in range(nch):
        for t, args in ((e, (ch,)), (refs[ch], ())):
            if which in ("ANF", "BOTH"):
                t.SetRXAANFRun(*args, 0); t.SetRXAANFRun(*args, 1)
            else:
                t.SetRXAANRRun(*args, 0); t.SetRXAANRRun(*args, 1)
    y2 = e.process_host(x[:, :20 * 1024])
    ref2 = np.stack([o.xrxa(x[ch, :20 * 1024]) for ch, o in enumerate(refs)])
    for ch in range(nch):
        assert rel_rms(y2[ch], ref2[ch]) < 1e-6


@pytest.mark.parametrize("mode,thresh,tail", [(1, -30.0, 0.2), (6, -35.0, 0.05), (1, -20.0, 1.5)])
def test_am_squelch(qh, oracle, mode, thresh, tail):
    """xamsqcap / xamsq (wdsp/amsq.c:119-192): the squelch opens and closes on the 10 ms average of the signal behind nbp0,
    with raised-cosine slews and a tail whose length depends on the level; the input fades in and out twice."""
    nch, nblk = 2, 260
    n = nblk * 1024
    x = synth.make_input_numpy(nch, n)
    env = np.ones(n)
    env[40 * 1024:110 * 1024] = 1e-3
    env[170 * 1024:200 * 1024] = 3e-2
    x = x * env
    e = qh.RxaEngine(nch)
    refs = []
    for ch in range(nch):
        o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
        for t, a in ((e, (ch,)), (o, ())):
            t.SetRXAShiftRun(*a, 1); t.SetRXAShiftFreq(*a, synth.shift_freq(ch)); t.RXANBPSetRun(*a, 1)
            t.SetRXAMode(*a, mode); t.RXASetPassband(*a, *((300.0, 3000.0) if mode == 1 else (-4000.0, 4000.0)))
            t.SetRXAAGCMode(*a, 0 if ch == 0 else 3)
            t.SetRXAAMSQThreshold(*a, thresh); t.SetRXAAMSQMaxTail(*a, tail); t.SetRXAAMSQRun(*a, 1)
        refs.append(o)
    ys, rs = [], [[] for _ in range(nch)]
    for a, b in ((0, 50), (50, 51), (51, 180), (180, nblk)):
        ys.append(e.process_host(x[:, a * 1024:b * 1024]))
        for ch in range(nch):
            rs[ch].append(refs[ch].xrxa(x[ch, a * 1024:b * 1024]))
    y = np.concatenate(ys, axis=1)
    for ch in range(nch):
        ref = np.concatenate(rs[ch])
        muted = np.count_nonzero(ref == 0)
        assert 1000 < muted < (nblk - 20) * 256                 # it did close (start-up at least), and it did open
        assert np.array_equal(y[ch] == 0, ref == 0)             # same samples muted
        assert rel_rms(y[ch], ref) < 1e-9


@pytest.mark.parametrize("in_rate,nblk", [(144000, 24), (240000, 16), (288000, 12), (24000, 60), (12000, 60)])
def test_general_input_rate_ratios(qh, oracle, in_rate, nblk):
    """in_rate / dsp_rate = 3, 5, 6, 1/2, 1/4: calc_resample's L / M (wdsp/resample.c:35-78) through the polyphase resampler
    instead of the fused overlap-save front stage; two ragged calls.  (Ratios that are not whole either way are refused: the
    reference's block sizes come from integer divisions of the rates, wdsp/channel.c:39-42.)"""
    nch, dsp_size = 2, 256
    insize = dsp_size * in_rate // 48000
    x = synth.make_input_numpy(nch, nblk * insize, fs=float(in_rate))
    e = qh.RxaEngine(nch, dsp_size=dsp_size, in_rate=in_rate)
    assert e.dsp_insize == insize
    refs = []
    for c in range(nch):
        f = 5000.0 + 37.0 * c
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, f); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
        e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 0); e.SetRXAAGCFixed(c, 0.0)
        o = oracle.WdspChannel(insize, dsp_size, in_rate, 48000, 48000)
        o.SetRXAShiftRun(1); o.SetRXAShiftFreq(f); o.RXANBPSetRun(1); o.SetRXAMode(1)
        o.RXASetPassband(300.0, 3000.0); o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0)
        refs.append(o.xrxa(x[c]))
    k = 7 * insize
    y = np.concatenate([e.process_host(np.ascontiguousarray(x[:, :k])), e.process_host(np.ascontiguousarray(x[:, k:]))], axis=1)
    for c in range(nch):
        assert rel_rms(y[c], refs[c]) < 1e-9, (in_rate, c)
    assert np.abs(y).max() > 1e-3
    if in_rate == 144000:
        with pytest.raises(qh.QuiskHipError):
            qh.RxaEngine(1, in_rate=72000)


@pytest.mark.parametrize("mode,settle", [(1, 0), (6, 0), (5, 150)])
def test_long_filters_nc_4096_and_back(qh, oracle, mode, settle):
    """RXASetNC(4096) (wdsp/RXA.c:934-946): impulse responses longer than 2048 taps run 8192-point overlap-save tiles; the
    change of nc zeroes the delay lines like setNc_fircore does, and going back to 2048 returns to 4096-point tiles."""
    nblk = 210 if mode == 5 else 60
    sig = {1: "usb", 6: "am", 5: "fm"}[mode]
    x = synth.make_mode_input_numpy(sig, 0, nblk * 1024)
    pb = {1: (300.0, 3000.0), 6: (-4000.0, 4000.0), 5: (-8000.0, 8000.0)}[mode]
    e = qh.RxaEngine(1)
    o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
    for obj, pre in ((e, (0,)), (o, ())):
        obj.SetRXAShiftRun(*pre, 1); obj.SetRXAShiftFreq(*pre, synth.shift_freq(0)); obj.RXANBPSetRun(*pre, 1)
        obj.SetRXAMode(*pre, mode); obj.RXASetPassband(*pre, *pb); obj.SetRXAAGCMode(*pre, 0); obj.SetRXAAGCFixed(*pre, 0.0)
    e.enable_meters(True)
    ys, rs = [], []
    cuts = [0, nblk // 3, 2 * nblk // 3 + 1, nblk]
    for k, nc in enumerate((None, 4096, 2048)):
        if nc:
            e.RXASetNC(0, nc); o.RXASetNC(nc)
        seg = x[cuts[k] * 1024:cuts[k + 1] * 1024]
        ys.append(e.process_host(seg[None, :])[0]); rs.append(o.xrxa(seg))
    y, ref = np.concatenate(ys), np.concatenate(rs)
    lo = settle * 256
    assert rel_rms(y[lo:], ref[lo:]) < (1e-9 if settle == 0 else 1e-6)
    if mode == 1:
        for mt in (0, 1, 2, 3, 5, 6):
            assert abs(e.GetRXAMeter(0, mt) - o.GetRXAMeter(mt)) < 0.002, mt
