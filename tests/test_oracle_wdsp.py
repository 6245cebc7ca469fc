"""Pins for the WDSP restatement that do not need the (unbuildable) reference: independent closed forms,
numpy/scipy recomputations, and the behaviours probed from the real reference recorded in SURVEY.md 8."""
import numpy as np

from conftest import rel_rms


def _bh4(N):
    i = np.arange(N)
    c = np.cos(np.pi / (0.5 * (N - 1)) * i)
    return 0.21747 + c * (-0.45325 + c * (0.28256 + c * (-0.04672)))


def test_fir_bandpass_closed_form(oracle):
    """fir_bandpass(rtype 1) = scale * sinc lowpass of half-width (fh-fl)/2 * BH window * exp(-j*w0*(n-m)),
    symmetric about m = (N-1)/2 (wdsp/fir.c:187-254)."""
    N, fl, fh, fs = 2048, 300.0, 3000.0, 48000.0
    h = oracle.fir_bandpass(N, fl, fh, fs, 0, 1, 1.0)
    m = 0.5 * (N - 1)
    pos = np.arange(N) - m
    ft = (fh - fl) / (2 * fs)
    proto = np.sin(2 * np.pi * ft * pos) / (np.pi * pos) * _bh4(N)
    want = proto * np.exp(-1j * np.pi * (fh + fl) / fs * pos)
    assert np.abs(h - want).max() < 1e-15
    # passband sits at conventional [-fh, -fl]: unity gain at -1000 Hz, > 100 dB down at +1000 Hz
    n = np.arange(N)
    g_neg = abs(np.sum(h * np.exp(+2j * np.pi * 1000.0 / fs * n)))
    g_pos = abs(np.sum(h * np.exp(-2j * np.pi * 1000.0 / fs * n)))
    assert abs(g_neg - 1.0) < 1e-4 and g_pos < 1e-5


def test_resampler_taps_and_output(oracle):
    h, L, M, ncoef, cpp = oracle.resample_taps(192000, 48000)
    assert (L, M, ncoef, cpp) == (1, 4, 561, 561)
    assert abs(h.sum() - 1.0) < 1e-8 and np.allclose(h, h[::-1], atol=1e-18)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(4096) + 1j * rng.standard_normal(4096)
    r = oracle.Resample(192000, 48000)
    y = np.concatenate([r(x[:1000]), r(x[1000:1004]), r(x[1004:])])
    want = np.convolve(x, h)[:4096][0::4]                       # y[m] = sum_j h[j] x[4m - j]
    assert y.size == 1024 and rel_rms(y, want) < 1e-13
    for rate, nco in ((96000, 281), (384000, 1121)):
        assert oracle.resample_taps(rate, 48000)[3] == nco


def test_fircore_is_linear_convolution(oracle):
    """Partitioned overlap-save == causal linear convolution with the nc-tap impulse times 2*size
    (the unnormalised inverse FFT, wdsp/firmin.c:409-430)."""
    rng = np.random.default_rng(2)
    size, nc = 64, 256
    imp = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
    x = rng.standard_normal(size * 20) + 1j * rng.standard_normal(size * 20)
    y = oracle.Fircore(size, nc, imp)(x)
    want = np.convolve(x, imp)[:x.size] * (2 * size)
    assert rel_rms(y, want) < 1e-13


def _usb_channel(po, in_size, in_rate, slew=True, nbp=True):
    a = (0.010, 0.025, 0.0, 0.010) if slew else (0.0, 0.0, 0.0, 0.0)
    ch = po.WdspChannel(in_size, 256, in_rate, 48000, 48000, *a)
    ch.SetRXAShiftRun(0)
    ch.RXANBPSetRun(1 if nbp else 0)
    ch.SetRXAMode(1)
    ch.RXASetPassband(300.0, 3000.0)
    ch.SetRXAAGCMode(0)
    ch.SetRXAAGCFixed(0.0)
    return ch


def test_probed_latency_and_slew(oracle):
    """SURVEY.md 8 row b2, probed from the compiled reference: two-block latency (first non-zero output 513
    without slew for in_size 64/256/1024), 993 at 192k->48k and 995 at 48k with Quisk's 0.010/0.025 s slew."""
    x = np.full(1024 * 24, 0.25 - 0.125j)
    for in_size in (64, 256, 1024):
        y, errs = _usb_channel(oracle, in_size, 192000, slew=False, nbp=False).fexchange0(x)
        assert errs == 0 and np.nonzero(np.abs(y) > 0)[0][0] == 513
    y, _ = _usb_channel(oracle, 1024, 192000, nbp=False).fexchange0(x)
    assert np.nonzero(np.abs(y) > 0)[0][0] == 993
    y, _ = _usb_channel(oracle, 256, 48000, nbp=False).fexchange0(x[:256 * 24])
    assert np.nonzero(np.abs(y) > 0)[0][0] == 995
    # steady state: DC 0.25-0.125j -> 1-0.5j (panel gain 4.0 applies even after SetRXAPanelRun(0))
    y, _ = _usb_channel(oracle, 1024, 192000, slew=False, nbp=False).fexchange0(x)
    assert abs(y[-1] - (1.0 - 0.5j)) < 1e-8


def test_probed_gain_and_sign_convention(oracle):
    """SURVEY.md hard part 6: a -1000 Hz tone passes 'USB 300..3000' with gain 4.0; +1000 Hz is rejected."""
    n = 1024 * 40
    t = np.arange(n)
    for f, lo, hi in ((-1000.0, 3.99, 4.01), (+1000.0, 0.0, 1e-5)):
        y, _ = _usb_channel(oracle, 1024, 192000).fexchange0(0.1 * np.exp(2j * np.pi * f * t / 192000.0))
        g = np.abs(y[-500:]).mean() / 0.1
        assert lo <= g <= hi, (f, g)


def test_chain_equals_independent_linear_model(oracle):
    """shift -> resample -> nbp -> gain recomputed with numpy from the designed taps only."""
    fs, shift = 192000.0, 10037.0
    rng = np.random.default_rng(5)
    n = 1024 * 12
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ch = _usb_channel(oracle, 1024, 192000)
    ch.SetRXAShiftRun(1)
    ch.SetRXAShiftFreq(shift)
    y = ch.xrxa(x)
    h1 = oracle.resample_taps(192000, 48000)[0]
    h2 = oracle.fir_bandpass(2048, 300.0, 3000.0, 48000.0, 0, 1, 1.0)
    xs = x * np.exp(2j * np.pi * ((shift / fs) * np.arange(n) % 1.0))
    mid = np.convolve(xs, h1)[:n][0::4]
    want = 4.0 * np.convolve(mid, h2)[:mid.size]
    assert rel_rms(y, want) < 1e-11


def test_mode_quirks(oracle):
    """bp1 is created run=1 and only re-evaluated on a mode CHANGE (RXA.c:378,751,815-827): opening and
    setting mode LSB (the default) leaves both filters in the chain; USB drops bp1."""
    rng = np.random.default_rng(6)
    x = rng.standard_normal(1024 * 8) + 1j * rng.standard_normal(1024 * 8)
    a = oracle.WdspChannel(1024, 256, 192000, 48000, 48000); a.SetRXAAGCMode(0); a.SetRXAMode(0)
    b = oracle.WdspChannel(1024, 256, 192000, 48000, 48000); b.SetRXAAGCMode(0); b.SetRXAMode(1); b.SetRXAMode(0)
    ya, yb = a.xrxa(x), b.xrxa(x)
    assert rel_rms(ya, yb) > 1e-3           # a still runs bp1 (BH-7 window) after nbp0, b does not


def test_agc_known_answers(oracle):
    """Independent checks of the wcpAGC restatement: (1) the look-ahead delay equals attack_buffsize =
    ceil(rate * n_tau * tau_attack) = 192 samples; (2) for a steady tone of amplitude A the gain settles to
    (out_target - slope_constant * min(0, log10(A'))) / A' with A' the level at the AGC input (wcpAGC.c:335)."""
    ch = oracle.WdspChannel(256, 256, 48000, 48000, 48000)
    ch.SetRXAShiftRun(0); ch.RXANBPSetRun(0); ch.SetRXAMode(1); ch.SetRXAAGCMode(4)      # fast
    ch.SetRXAPanelGain1(1.0)
    n = 256 * 400
    x = np.zeros(n, dtype=np.complex128)
    x[1000:] = 0.05 * np.exp(2j * np.pi * 0.01 * np.arange(n - 1000))
    y = ch.xrxa(x)
    first = np.nonzero(np.abs(y) > 0)[0][0]
    assert first == 1000 + 192
    out_target = 1.0 * (1.0 - np.exp(-4.0)) * 0.9999
    var_gain, max_gain = 1.5, 10000.0
    slope = (out_target * (1.0 - 1.0 / var_gain)) / np.log10(out_target / (1.0 * var_gain * max_gain))
    want = out_target - slope * min(0.0, np.log10(0.05))
    assert abs(np.abs(y[-2000:]).mean() - want) < 1e-6 * want


def test_make_nbp_known_answers(oracle):
    """make_nbp (wdsp/nbp.c:97-179), worked by hand: a notch inside the passband splits it, one over an edge trims it,
    an inactive or outside one does nothing, one covering everything leaves no band, narrow ones are widened."""
    import ctypes as C
    L = oracle.lib()
    L.wo_make_nbp.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_double, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 3

    def run(notches, flow, fhigh, minwidth=0.0, autoincr=0):
        act = (C.c_int * 1024)(*[n[2] for n in notches])
        cen = (C.c_double * 1024)(*[n[0] for n in notches])
        wid = (C.c_double * 1024)(*[n[1] for n in notches])
        nlo = (C.c_double * 1024)(*[n[0] - n[1] / 2 for n in notches])
        nhi = (C.c_double * 1024)(*[n[0] + n[1] / 2 for n in notches])
        bl, bh, hv = (C.c_double * 1025)(), (C.c_double * 1025)(), C.c_int(0)
        n = L.wo_make_nbp(len(notches), act, cen, wid, nlo, nhi, minwidth, autoincr, flow, fhigh, bl, bh, C.byref(hv))
        return [(bl[i], bh[i]) for i in range(n)], hv.value

    assert run([(1000, 200, 1)], 300, 3000) == ([(300.0, 900.0), (1100.0, 3000.0)], 1)
    assert run([(1000, 200, 1), (2900, 400, 1), (100, 600, 1), (5000, 100, 1), (1500, 50, 0)], 300, 3000) == (
        [(400.0, 900.0), (1100.0, 2700.0)], 1)
    assert run([(1000, 10000, 1)], 300, 3000) == ([], 1)
    assert run([(1000, 20, 1)], 300, 3000, minwidth=200, autoincr=1) == ([(300.0, 900.0), (1100.0, 3000.0)], 1)
    assert run([(1000, 20, 1)], 3000, 300) == ([], 0)


def _ssb_channel(oracle):
    from quisk_amd import synth
    ch = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
    ch.SetRXAShiftRun(1); ch.SetRXAShiftFreq(synth.shift_freq(0)); ch.RXANBPSetRun(1); ch.SetRXAMode(1); ch.RXASetPassband(300.0, 3000.0)
    ch.SetRXAAGCMode(0); ch.SetRXAAGCFixed(0.0)
    return ch


def test_anf_takes_a_carrier_out_and_anr_keeps_it(oracle):
    """xanf outputs the LMS prediction error (the steady tone goes), xanr the prediction (the tone stays, noise drops); both
    force bp1 on with gain 2 (RXAbp1Check, RXA.c:800-813), which makes up for dropping the imaginary part."""
    from quisk_amd import synth
    x = synth.make_input_numpy(1, 600 * 1024)[0]

    def tone_and_floor(y):
        S = np.abs(np.fft.fft(y[-16384:] * np.hanning(16384))) ** 2
        k = int(np.argmax(S))
        return 10 * np.log10(S[k - 3:k + 4].sum()), 10 * np.log10((S.sum() - S[k - 3:k + 4].sum()) / 16384)

    plain = tone_and_floor(_ssb_channel(oracle).xrxa(x))
    a = _ssb_channel(oracle); a.SetRXAANFRun(1)
    anf = tone_and_floor(a.xrxa(x))
    r = _ssb_channel(oracle); r.SetRXAANRRun(1)
    anr = tone_and_floor(r.xrxa(x))
    assert anf[0] < plain[0] - 10.0                         # the carrier is notched
    assert abs(anr[0] - plain[0]) < 3.0                     # the carrier stays (real part only: half the amplitude, then bp1's gain of 2)
    assert anr[1] < plain[1] - 3.0                          # and the noise floor is at least 3 dB down


def test_am_squelch_opens_on_signal_and_closes_after_its_tail(oracle):
    from quisk_amd import synth
    n = 300 * 1024
    x = synth.make_input_numpy(1, n)[0]
    env = np.ones(n); env[100 * 1024:] = 1e-4
    ch = _ssb_channel(oracle)
    ch.SetRXAAMSQThreshold(-30.0); ch.SetRXAAMSQMaxTail(0.1); ch.SetRXAAMSQRun(1)
    y = ch.xrxa(x * env)
    lvl = [np.abs(y[k * 256:(k + 1) * 256]).max() for k in range(300)]
    assert lvl[0] == 0.0 and max(lvl[60:95]) > 0.1          # muted at the start, open (after the 70 ms slew) while the signal is there
    assert max(lvl[160:]) == 0.0                            # closed again: tail <= 0.1 s + 70 ms slew after the fade at block 100


def test_mlog10_is_pinned_to_the_reference_table(oracle):
    """wdsp/meterlog10.c keeps mtable[2048]; the restatements use log2(1 + m / 2048).  tests/golden/mlog10_pin.json (made here by
    tests/golden/make_mlog10_pin.py from the reference's own numbers) records that the two agree to one ulp, and the constants."""
    import ctypes as C
    import json
    import os
    pin = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlog10_pin.json")))
    assert pin["entries"] == 2048 and pin["mbits"] == 11 and pin["mconv"] == 0.301029995663981
    assert pin["max_abs_deviation_from_log2_1_plus_m_over_2048"] < 2.3e-16 and pin["first"] == 0.0
    L = oracle.lib()
    L.wo_mlog10_value.restype = C.c_double
    L.wo_mlog10_value.argtypes = [C.c_double]
    for v in (1.0, 1.0 + 1.0 / 2048.0, 0.75, 3.1e-7, 2.0 ** -40 * 1.9995):
        bits = np.float64(v).view(np.uint64)
        e = int((bits >> np.uint64(52)) & np.uint64(2047)) - 1023
        m = int((bits >> np.uint64(41)) & np.uint64(2047))
        assert L.wo_mlog10_value(v) == pin["mconv"] * (e + np.log2(1.0 + m / 2048.0))
    assert abs(L.wo_mlog10_value(1.0 + 2047.0 / 2048.0) / pin["mconv"] - pin["last"]) < 3e-16


def test_rates_that_are_whole_in_neither_direction_recycle_stale_buffer_tails(oracle):
    """Why qh_rxa_create refuses in_rate / dsp_rate / out_rate such as 96 k -> 64 k -> 48 k: the reference sizes
    its blocks with integer divisions (pre_main_build, channel.c:39-52: dsp_insize = dsp_size * (96000 / 64000) = dsp_size, dsp_outsize =
    dsp_size / (64000 / 48000) = dsp_size), xresample writes the samples it has (resample.c:120-157: about 2/3, then 3/4 of a block) and
    every stage behind it works on dsp_size samples of a buffer whose tail still holds what the block before left there.  A tone in is not
    a tone out: the tail of every output block is the tail of the block before it (here: never written at all).  The arithmetic of the
    resampler is there for any L / M (qh_rat_*); a chain that feeds on its own leftovers block by block is not a stream to reproduce."""
    o = oracle.WdspChannel(384, 256, 96000, 64000, 48000)
    assert (o.dsp_insize, o.dsp_outsize) == (256, 256)
    o.SetRXAMode(1); o.RXANBPSetRun(0); o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0)
    nblk = 8
    x = np.exp(2j * np.pi * 1000.0 / 96000.0 * np.arange(o.dsp_insize * nblk))
    y = o.xrxa(x).reshape(nblk, -1)
    for b in range(2, nblk):
        assert np.array_equal(y[b, 200:], y[b - 1, 200:])          # 256 * 2/3 * 3/4 = 128 new samples a block; the rest is left over
    step = np.angle(y[4, 1:100] * np.conj(y[4, :99])) * 48000 / (2 * np.pi)
    assert np.abs(step - 1000.0).max() > 100.0                     # and what is new is not the 1 kHz tone either
