"""qh_quisk_process_samples against the block-level restatement of quisk_process_samples (oracle qo_ps_*, quisk.c:2289-2742):
the tail (cFracDecim, the WDSP hand-off, HB45 interpolation to the playback rate), AddTestTone, measure_freq, sub-receiver 1's
digital output, Buffer2Chan with ragged blocks, the split -> played-sub-receiver hand-over of bank 1, squelch placement behind the
AGC.  Through the C ABI (quisk_amd.quiskapi).  -m gpu."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_rms
from quisk_amd import rxfilter

pytestmark = pytest.mark.gpu
D = C.c_double


def _filters(mode_name, mode, bw, fs=192000):
    frate = rxfilter.get_filter_rate(fs, mode, bw)
    return rxfilter.make_filter_coef(frate, None, bw, rxfilter.get_filter_center(mode_name, bw))


def _two_tone(fs, n, f1, f2, seed, amp=2.0 ** 22):
    t = np.arange(n)
    rng = np.random.default_rng(seed)
    return (amp * np.exp(2j * np.pi * ((f1 / fs) * t % 1.0)) + 0.5 * amp * np.exp(2j * np.pi * ((f2 / fs) * t % 1.0))
            + amp / 512 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))


def _pair(qh, oracle, fs, play, fft_size=0, data_width=0):
    api = qh.quiskapi
    api.open(fs, fft_size=fft_size, data_width=data_width, playback_rate=play)
    return api, oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())


def _both(api, ref, name, *args):
    getattr(api, name)(*args)
    getattr(ref, name)(*args)


@pytest.mark.parametrize("play", [48000, 96000, 192000, 384000])
def test_playback_rate_interpolation(qh, oracle, play):
    """quisk.c:2663-2682: HalfBand7 (8, 9) chained bring the 48 ksps audio to the playback rate ahead of process_agc."""
    fs, blk, nblk = 192000, 4096, 16
    api, ref = _pair(qh, oracle, fs, play)
    fI, fQ = _filters("USB", 3, 2700)
    _both(api, ref, "set_rx_mode", 3)
    api.set_tune(10000); ref.set_tune(10000)
    _both(api, ref, "set_filters", fI, fQ, 2700)
    x = _two_tone(fs, blk * nblk, 10900.0, 30000.0, 11)
    outs, refs = [], []
    for k in range(nblk):
        seg = x[k * blk:(k + 1) * blk]
        outs.append(api.process(seg)); refs.append(ref.process(seg))
        assert outs[-1].size == refs[-1].size == blk // 4 * (play // 48000)
    api.close()
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert np.abs(want).max() > 2.0 ** 24 and rel_rms(y, want) < 1e-8


@pytest.mark.parametrize("fs,play", [(111111, 48000), (53333, 96000), (133333, 48000), (185185, 192000)])
def test_sdriq_rates_take_the_fractional_decimator(qh, oracle, fs, play):
    """quisk.c:2654-2659: the SDR-IQ rates leave quisk_process_decimate at 55555 / 53333 / 66666 / 61728 sps; cFracDecim's
    4-point interpolator (quisk.c:622-665) brings them to 48000, ragged block lengths and all."""
    api, ref = _pair(qh, oracle, fs, play)
    fI, fQ = rxfilter.make_filter_coef(rxfilter.get_filter_rate(fs, 3, 2700), None, 2700, rxfilter.get_filter_center("USB", 2700))
    _both(api, ref, "set_rx_mode", 3)
    api.set_tune(5000); ref.set_tune(5000)
    _both(api, ref, "set_filters", fI, fQ, 2700)
    sizes = [4001, 3999, 4096, 1237, 8191, 4000, 4000, 2222, 6001, 4096, 4096, 4096]
    x = _two_tone(fs, sum(sizes), 5900.0, 17000.0, 12)
    outs, refs, pos = [], [], 0
    for s in sizes:
        seg = x[pos:pos + s]; pos += s
        outs.append(api.process(seg)); refs.append(ref.process(seg))
        assert outs[-1].size == refs[-1].size            # the output index is a closed form of the carried dindex: same counts
    api.close()
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert abs(y.size - sum(sizes) * play / fs) < 4 * play // 48000
    assert np.abs(want).max() > 2.0 ** 24 and rel_rms(y, want) < 1e-8


@pytest.mark.parametrize("mode,name,bw,inv", [(3, "USB", 2700, 0), (4, "AM", 6000, 1), (5, "FM", 12000, 0)])
def test_add_tone_inversion_and_blanker_ahead_of_the_bank(qh, oracle, mode, name, bw, inv):
    """AddTestTone (quisk.c:1258-1303: plain, AM-modulated, FM-modulated by mode), the inversion and NoiseBlanker act on the
    block ahead of the FFT ring and the tune (quisk.c:2438-2449); the panadapter sees them too."""
    fs, blk, nblk = 192000, 4096, 20
    api, ref = _pair(qh, oracle, fs, 48000, fft_size=2048, data_width=512)
    g = oracle.OracleGraph(2048, 512, float(fs))
    ref.set_graph(g)
    fI, fQ = _filters(name, mode, bw)
    _both(api, ref, "set_rx_mode", mode)
    api.set_tune(10000); ref.set_tune(10000)
    _both(api, ref, "set_filters", fI, fQ, bw)
    _both(api, ref, "add_tone", 10800)
    _both(api, ref, "invert_spectrum", inv)
    _both(api, ref, "set_noise_blanker", 2)
    rng = np.random.default_rng(13)
    x = 2.0 ** 12 * (rng.standard_normal(blk * nblk) + 1j * rng.standard_normal(blk * nblk))
    x[5000::9973] += 2.0 ** 26                           # clicks for the blanker
    outs, refs = [], []
    for k in range(nblk):
        seg = x[k * blk:(k + 1) * blk]
        outs.append(api.process(seg)); refs.append(ref.process(seg))
    y, want = np.concatenate(outs), np.concatenate(refs)
    skip = 6 * 1024 if mode == 5 else 0                  # FM: the discriminator's argument of rounding-level numbers while the filters fill
    assert y.size == want.size and np.abs(want[skip:]).max() > 2.0 ** 20
    assert rel_rms(y[skip:], want[skip:]) < 1e-6
    pix, sm, cnt = api.get_graph(1.0, 0.0)
    rp, rs, rc = g.get(1.0, 0.0)
    assert cnt == rc and np.abs(pix - rp).max() < 1e-7
    api.close()


def test_measure_freq_matches_the_restatement(qh, oracle):
    """measure_freq (quisk.c:5579-5649) on the decimated samples: /8 by HalfBand1..3, 12000-point transforms under its own Hanning
    window, |X| averaged in frequency order, peak +- 500 Hz of the Rx frequency with three-point interpolation."""
    fs, blk = 192000, 19200
    api, ref = _pair(qh, oracle, fs, 48000)
    fI, fQ = _filters("USB", 3, 2700)
    _both(api, ref, "set_rx_mode", 3)
    api.set_tune(7000); ref.set_tune(7000)
    _both(api, ref, "set_filters", fI, fQ, 2700)
    assert api.measure_frequency(4) == 0.0 and ref.measure_frequency(4) == 0.0       # a result every 2 transforms
    f_true = 7123.4
    nblk = 45                                            # 12000 samples at 6 ksps = 20 blocks per transform
    x = _two_tone(fs, blk * nblk, f_true, 40000.0, 14)
    seen = []
    for k in range(nblk):
        seg = x[k * blk:(k + 1) * blk]
        a, b = api.process(seg), ref.process(seg)
        assert a.size == b.size
        seen.append((api.measure_frequency(-1), ref.measure_frequency(-1)))
    api.close()
    got, want = seen[-1]
    assert abs(want - f_true) < 0.2
    assert abs(got - want) < 1e-6
    # both report at the same block (the transform count and the dropped tail of the filling call agree)
    first_g = next(i for i, s in enumerate(seen) if s[0] != 0.0)
    first_r = next(i for i, s in enumerate(seen) if s[1] != 0.0)
    assert first_g == first_r


def test_split_then_played_sub_receiver_share_bank_1(qh, oracle):
    """Bank 1's filter storage is shared by split Rx/Tx and the played sub-receiver while each has its own tune vector and filter set
    (quisk.c:2540-2545 txTuneVector / nFilter 0, quisk.c:2592-2600 aux1TuneVector / nFilter 1); Buffer2Chan evens out the ragged
    counts after split comes on in mid-stream (quisk.c:1577-1611) and starts over on either change (quisk.c:2361-2367)."""
    fs = 192000
    api, ref = _pair(qh, oracle, fs, 48000)
    fI, fQ = _filters("USB", 3, 2700)
    gI, gQ = _filters("USB", 3, 2400)
    for o in (api, ref):
        o.set_rx_mode(3)
        o.set_filters(fI, fQ, 2700)
        o.set_filters(gI, gQ, 2400, 1)                   # nFilter 1; the ONE sizeFilter is now this call's (quisk.c:4591)
        o.set_filters(fI, fQ, 2700)                      # ... and back to this one's
        o.set_multirx_mode(1, 3); o.set_multirx_freq(1, -15000); o.set_multirx_play_method(1)
    api.set_tune2(10000, 21000); ref.set_tune(10000, 21000)
    assert len(fI) != len(gI)                            # 72 and 82 taps: bank 1 runs the first 72 of its 82, as the reference would
    sizes = [1001, 2003, 997, 4099, 1501, 3001, 2999, 1777, 3333, 2048, 4096, 1234, 4096, 4096, 4096, 1000, 3000, 4096, 4096, 4096]
    x = _two_tone(fs, sum(sizes), 10900.0, 21900.0, 15)
    xs = _two_tone(fs, sum(sizes), -14200.0, 30000.0, 16)
    outs, refs, pos = [], [], 0
    for i, s in enumerate(sizes):
        if i == 3:
            _both(api, ref, "set_split_rxtx", 1)
        if i == 9:
            _both(api, ref, "set_split_rxtx", 0)
            _both(api, ref, "set_multirx_play_channel", 1)
        if i == 15:
            _both(api, ref, "set_multirx_play_channel", -1)
            _both(api, ref, "set_split_rxtx", 2)         # back to split: txTuneVector carries on from where it stopped
        seg, sub = x[pos:pos + s], xs[pos:pos + s]; pos += s
        api.multirx_samples(1, sub); ref.multirx_samples(1, sub)
        outs.append(api.process(seg)); refs.append(ref.process(seg))
        assert outs[-1].size == refs[-1].size, i
    api.close()
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert np.abs(want.real).max() > 2.0 ** 24 and np.abs(want.imag).max() > 2.0 ** 24
    assert rel_rms(y, want) < 1e-8
    assert rel_rms(y.real, y.imag) > 0.1


def test_split_receiver_sees_the_raw_block(qh, oracle):
    """orig_cSamples is copied at the top of the function (quisk.c:2361-2363): the second receiver of split mode gets the block
    without the test tone, the inversion and the blanker."""
    fs, blk, nblk = 192000, 4096, 12
    api, ref = _pair(qh, oracle, fs, 48000)
    fI, fQ = _filters("USB", 3, 2700)
    for o in (api, ref):
        o.set_rx_mode(3); o.set_filters(fI, fQ, 2700); o.set_split_rxtx(1); o.add_tone(21700); o.set_noise_blanker(1)
    api.set_tune2(10000, 21000); ref.set_tune(10000, 21000)
    x = _two_tone(fs, blk * nblk, 10900.0, 21900.0, 17)
    x[3000::7919] += 2.0 ** 27
    outs, refs = [], []
    for k in range(nblk):
        seg = x[k * blk:(k + 1) * blk]
        outs.append(api.process(seg)); refs.append(ref.process(seg))
    api.close()
    assert rel_rms(np.concatenate(outs), np.concatenate(refs)) < 1e-8


@pytest.mark.parametrize("mode", [7, 9, 13])
def test_sub_receiver_1_digital_output(qh, oracle, mode):
    """quisk.c:2630-2651: with a digital mode on sub-receiver 1 and a sound device for it, its samples are demodulated on bank 2 with
    aux2TuneVector and filter set 2, run through Agc3 and handed to play_sound_interface."""
    fs, blk, nblk = 96000, 4800, 12
    api, ref = _pair(qh, oracle, fs, 48000)
    fI, fQ = _filters("USB", 3, 2700, fs)
    bw2 = {7: 3200, 9: 8000, 13: 12000}[mode]
    name = {7: "DGT-U", 9: "DGT-IQ", 13: "DGT-FM"}[mode]
    frate = rxfilter.get_filter_rate(fs, mode, bw2)
    hI, hQ = rxfilter.make_filter_coef(frate, None, bw2, rxfilter.get_filter_center(name, bw2))
    for o in (api, ref):
        o.set_rx_mode(3)
        o.set_filters(fI, fQ, 2700)
        o.set_filters(hI[:len(fI)] if len(hI) >= len(fI) else np.concatenate([hI, np.zeros(len(fI) - len(hI))]),
                      hQ[:len(fQ)] if len(hQ) >= len(fQ) else np.concatenate([hQ, np.zeros(len(fQ) - len(hQ))]), bw2, 2)
        o.set_multirx_count(1); o.set_multirx_mode(0, mode); o.set_multirx_freq(0, -8000); o.set_sub_rx1_output(1)
    api.set_tune(10000); ref.set_tune(10000)
    x = _two_tone(fs, blk * nblk, 10900.0, 30000.0, 18)
    xs = _two_tone(fs, blk * nblk, -7100.0, -6400.0, 19)
    outs, refs, s_out, s_ref = [], [], [], []
    for k in range(nblk):
        seg, sub = x[k * blk:(k + 1) * blk], xs[k * blk:(k + 1) * blk]
        api.multirx_samples(0, sub); ref.multirx_samples(0, sub)
        outs.append(api.process(seg)); refs.append(ref.process(seg))
        s_out.append(api.sub_rx1_audio()); s_ref.append(ref.sub_rx1_audio())
        assert s_out[-1].size == s_ref[-1].size == blk // 2
    api.close()
    assert rel_rms(np.concatenate(outs), np.concatenate(refs)) < 1e-8
    skip = 3 * blk // 2 if mode == 13 else 0
    a, b = np.concatenate(s_out)[skip:], np.concatenate(s_ref)[skip:]
    assert np.abs(b).max() > 2.0 ** 20 and rel_rms(a, b) < 1e-6


def test_wdsp_hand_off_inside_the_block(qh, oracle):
    """quisk.c:2660-2661: the 48 ksps stereo audio goes through wdspFexchange0 (quisk_wdsp.c:24-69: re-blocked to in_size, scaled by
    1 / CLIP32 and back) between cFracDecim and the interpolation; here WDSP is this library's own RXA engine opened the way
    quisk_wdsp.py opens it (quisk_wdsp.py:69-99), the oracle side the restated shim in front of the restated RXA chain."""
    fs, play, blk, nblk = 192000, 96000, 4000, 24
    lib = qh.load()
    api, ref = _pair(qh, oracle, fs, play)
    lib.OpenChannel(0, 256, 256, 48000, 48000, 48000, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
    lib.SetRXAShiftRun(0, 0); lib.RXANBPSetRun(0, 0); lib.SetRXAAMSQRun(0, 0); lib.SetRXAMode(0, 1)
    lib.RXASetPassband(0, D(300.0), D(3000.0)); lib.RXASetNC(0, 256); lib.RXASetMP(0, 0)
    lib.SetRXAAGCMode(0, 0); lib.SetRXAAGCFixed(0, D(0.0)); lib.SetRXAPanelRun(0, 0); lib.SetRXAEMNRRun(0, 0)
    lib.qh_wdsp_set_parameter(0, 256, 0)
    ch = oracle.WdspChannel(256, 256, 48000, 48000, 48000)
    ch.SetRXAShiftRun(0); ch.RXANBPSetRun(0); ch.SetRXAMode(1); ch.RXASetPassband(300.0, 3000.0); ch.RXASetNC(256)
    ch.SetRXAAGCMode(0); ch.SetRXAAGCFixed(0.0)
    shim = oracle.OracleWdspShim(lambda pin, pout: 0)
    shim.set_parameter(in_size=256, in_use=0)
    ref.set_wdsp(shim, ch)
    fI, fQ = _filters("USB", 3, 2700)
    for o in (api, ref):
        o.set_rx_mode(3); o.set_filters(fI, fQ, 2700)
    api.set_tune(10000); ref.set_tune(10000)
    x = _two_tone(fs, blk * nblk, 10900.0, 30000.0, 20)
    try:
        outs, refs = [], []
        for k in range(nblk):
            if k == 2:                                   # switched on once the audio is running: WDSP's up-slew starts at the first
                lib.qh_wdsp_set_parameter(0, -1, 1); shim.set_parameter(in_use=1)      # non-zero sample, and FFT filters leave 1e-15 where FIR loops leave 0
            if k == 16:                                  # in_use off: the audio passes, the shim's ring is rewound (quisk_wdsp.c:32-37)
                lib.qh_wdsp_set_parameter(0, -1, 0); shim.set_parameter(in_use=0)
            if k == 20:
                lib.qh_wdsp_set_parameter(0, -1, 1); shim.set_parameter(in_use=1)
            seg = x[k * blk:(k + 1) * blk]
            outs.append(api.process(seg)); refs.append(ref.process(seg))
            assert outs[-1].size == refs[-1].size, k     # 1000 samples in, whole 256-blocks out, twice that after the interpolator
    finally:
        lib.qh_wdsp_set_parameter(0, -1, 0)
        lib.CloseChannel(0)
        api.close()
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert np.abs(want).max() > 2.0 ** 24 and rel_rms(y, want) < 1e-8


def test_key_down_and_up_at_a_faster_playback_rate(qh, oracle):
    """The key-down replacement counts playback samples (quisk.c:2372-2375) and so does the demodulated path now: the stream keeps
    one clock across key changes; sidetone, silence and the 5 ms key-up ramp at 96 ksps."""
    fs, play, blk = 192000, 96000, 3200
    api, ref = _pair(qh, oracle, fs, play)
    fI, fQ = _filters("CWU", 1, 1000)
    for o in (api, ref):
        o.set_rx_mode(1); o.set_filters(fI, fQ, 1000)
    api.set_tune(10000); ref.set_tune(10000)
    api.set_sidetone(0.5, 600, play, 50); ref.set_sidetone(0.5, 600, 50)
    x = _two_tone(fs, blk * 40, 10600.0, 50000.0, 21)
    outs, refs = [], []
    for k in range(40):
        down = 8 <= k < 14
        _both(api, ref, "set_key_state", int(down), int(down), 2, 0)
        seg = x[k * blk:(k + 1) * blk]
        outs.append(api.process(seg)); refs.append(ref.process(seg))
        assert outs[-1].size == refs[-1].size == blk * play // fs
    api.set_kill_audio(1); ref.set_kill_audio(1)
    assert np.all(api.process(x[:blk]) == 0) and api.squelch_flags() == 3
    api.close()
    y, want = np.concatenate(outs), np.concatenate(refs)
    assert np.abs(want[8 * 1600:14 * 1600]).max() > 1e9 and rel_rms(y, want) < 1e-8


def test_fm_squelch_mutes_one_side_behind_the_agc(qh, oracle):
    """quisk.c:2548-2562,2712-2728: with split the banks' squelch flags go to the output channel their audio went to, and they act
    behind the AGCs (which keep seeing the unsquelched audio).  A carrier on the Rx frequency only: the Tx side is squelched."""
    fs, blk, nblk = 192000, 4800, 30
    api, ref = _pair(qh, oracle, fs, 48000)
    fI, fQ = _filters("FM", 5, 12000)
    for o in (api, ref):
        o.set_rx_mode(5); o.set_filters(fI, fQ, 12000); o.set_split_rxtx(1); o.set_squelch(-60.0)
    api.set_tune2(10000, 60000); ref.set_tune(10000, 60000)
    t = np.arange(blk * nblk)
    rng = np.random.default_rng(22)
    x = 2.0 ** 26 * np.exp(2j * np.pi * (10000.0 / fs * t + 0.5 * np.sin(2 * np.pi * 1000.0 / fs * t))) + \
        2.0 ** 4 * (rng.standard_normal(t.size) + 1j * rng.standard_normal(t.size))
    outs, refs, fl = [], [], []
    for k in range(nblk):
        seg = x[k * blk:(k + 1) * blk]
        outs.append(api.process(seg)); refs.append(ref.process(seg))
        fl.append((api.squelch_flags(), ref.squelch_flags()))
    api.close()
    assert all(a == b for a, b in fl[2:]) and fl[-1][0] == 1          # tx (60 kHz) > rx: the Tx bank is the real channel, muted
    y, want = np.concatenate(outs)[8 * 1200:], np.concatenate(refs)[8 * 1200:]
    assert np.all(want.real == 0) and np.all(y.real == 0)
    assert np.abs(want.imag).max() > 2.0 ** 20 and rel_rms(y, want) < 1e-6


@pytest.mark.parametrize("mode,name,bw,width,fft,extra_taps", [(3, "USB", 2700, 1024, 2048, 0), (4, "AM", 6000, 950, 3800, 0), (1, "CWU", 500, 800, 800, 0),
                                                                (3, "USB", 2700, 1024, 1024, 9000)],
                         ids=["usb-1024", "am-950-of-3800", "cw-800", "usb-10k-taps"])
def test_get_filter_matches_the_restatement(qh, oracle, mode, name, bw, width, fft, extra_taps):
    """get_filter (quisk.c:5481-5568), the "RX Filter" screen's curve: multitone, the cRxFilterOut loop's own copy there, record_app's
    window, a data_width-point transform (any width), dB with the -140 floor.  Also with a filter near MAX_FILTER_SIZE (10001)."""
    api = qh.quiskapi
    api.open(48000, fft_size=fft, data_width=width)
    fI, fQ = _filters(name, mode, bw, fs=48000)
    if extra_taps:                                       # a long filter: the short one stretched by interpolation
        k = np.linspace(0, fI.size - 1, fI.size + extra_taps)
        fI, fQ = np.interp(k, np.arange(fI.size), fI) * fI.size / k.size, np.interp(k, np.arange(fQ.size), fQ) * fQ.size / k.size
        assert fI.size > 9000
    api.set_rx_mode(mode)
    api.set_filters(fI, fQ, bw)
    got = api.get_filter()
    api.close()
    want = oracle.get_filter(fI, fQ, width, fft)
    assert got.size == want.size == width
    assert want.max() > -40.0 and want.min() < -60.0
    top = want > -60.0
    assert np.abs(got[top] - want[top]).max() < 1e-6     # dB
    assert np.abs(got - want).max() < 1e-3               # far down the skirts the last digits of a 1e-6 magnitude decide


def test_mode_ext_hands_the_tuned_block_to_the_registered_demodulator(qh, oracle):
    """Mode EXT (quisk.c:2490-2493): NoiseBlanker / FFT ring as always, the tune, then the user's quisk_extern_demod (extdemod.c:13)
    in place and straight to process_agc(.., 1) ("goto start_agc").  The callback here is extdemod.c's own narrow-FM discriminator,
    restated; the expected block = the tune in closed form, that function, the restatement's process_agc."""
    fs, blk, nblk, tune = 48000, 4000, 12, 6000
    api, lib = qh.quiskapi, qh.load()
    api.open(fs, playback_rate=48000)
    api.set_rx_mode(6)
    api.set_tune(tune)
    rng = np.random.default_rng(23)
    t = np.arange(blk * nblk)
    x = 2.0 ** 24 * np.exp(2j * np.pi * ((tune / fs) * t % 1.0) + 2j * np.sin(2 * np.pi * 700.0 / fs * t)) + \
        2.0 ** 12 * (rng.standard_normal(t.size) + 1j * rng.standard_normal(t.size))
    # without a registered demodulator the mode fails loudly: 0 samples and the error counter moves
    before = lib.qh_quisk_error_count()
    assert api.process(x[:blk]).size == 0 and lib.qh_quisk_error_count() == before + 1
    assert b"quisk_extern_demod" in lib.qh_last_error()

    state = {"fm_1": 10 + 0j, "fm_2": 10 + 0j, "decims": []}

    def extern_demod(buf, n, decim):                     # extdemod.c:13-47
        state["decims"].append(decim)
        a = np.ctypeslib.as_array(buf, shape=(2 * n,))
        z = a[0::2] + 1j * a[1::2]
        out = np.empty(n)
        f1, f2 = state["fm_1"], state["fm_2"]
        for i in range(n):
            cx = z[i]
            di = f1.real * (cx.imag - f2.imag) - f1.imag * (cx.real - f2.real)
            d = f1.real * f1.real + f1.imag * f1.imag
            out[i] = 0.0 if d == 0 else di / d * fs
            f2, f1 = f1, cx
        state["fm_1"], state["fm_2"] = f1, f2
        a[0::2] = out; a[1::2] = out
        return n
    CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double)
    cb = CB(extern_demod)
    lib.qh_quisk_set_extern_demod.argtypes = [CB]
    lib.qh_quisk_set_extern_demod(cb)
    got = np.concatenate([api.process(x[k * blk:(k + 1) * blk]) for k in range(nblk)])
    lib.qh_quisk_set_extern_demod(CB())                  # unregister before the callback object goes away
    api.close()
    assert state["decims"] and all(abs(d - 1.0) < 1e-12 for d in state["decims"])
    # expected: tune (x * exp(-j 2 pi tune n / fs), the vector the reference steps sample by sample), the demodulator, process_agc
    state.update(fm_1=10 + 0j, fm_2=10 + 0j)
    tuned = x * np.exp(-2j * np.pi * ((tune / fs) * np.arange(x.size) % 1.0))       # (the refused call did not advance anything)
    buf = np.empty(2 * tuned.size)
    buf[0::2], buf[1::2] = tuned.real, tuned.imag
    extern_demod(buf.ctypes.data_as(C.POINTER(C.c_double)), tuned.size, 1.0)
    audio = buf[0::2] + 1j * buf[1::2]
    agc = oracle.OracleQuiskAgc(48000)
    want = np.concatenate([agc.process(audio[k * blk:(k + 1) * blk], True, 80.0) for k in range(nblk)])
    assert got.size == want.size
    assert np.abs(want).max() > 2.0 ** 20
    assert rel_rms(got, want) < 1e-8
