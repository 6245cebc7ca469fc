"""Seeded random walks over the one-receiver quisk_process_samples (qh_quisk_*, the `_quisk` setters' shape) between ragged calls,
the same calls made on the block-level restatement (oracle qo_ps_*, quisk.c:2289-2742): tune (rx and tx), Rx filters of other
lengths, split Rx/Tx 0-4, the key going down and up (sidetone / silence, the key-up envelope), AGC level, NoiseBlanker, auto-notch,
inversion, kill_audio, both squelches, the test tone (FM).  Mode and rates are fixed per walk (a change of mode rebuilds a bank: a
stated deviation).  Levels stay under process_agc's limiter (the end of an overload ramp is chaotic,
tests/test_gpu_bench_shapes.py; the machine is held bit for bit in tests/test_gpu_quisk_agc_chain.py).  -m gpu."""
import os

import numpy as np
import pytest

from quisk_amd import rxfilter
from test_gpu_quisk_process_bank import BW, NAMES, _filters, _signal

pytestmark = pytest.mark.gpu


def _draw(rng, mode, fs, play, api, ref, st):
    k = int(rng.integers(0, 16 if mode == 3 else 13))
    if k == 0:
        st["rx"] = int(rng.integers(-30000, 30000))
        api.set_tune2(st["rx"], st["tx"]); ref.set_tune(st["rx"], st["tx"]); return ("set_tune rx", st["rx"])
    if k == 1:
        st["tx"] = int(rng.integers(-30000, 30000))
        api.set_tune2(st["rx"], st["tx"]); ref.set_tune(st["rx"], st["tx"]); return ("set_tune tx", st["tx"])
    if k == 2:
        bw = int(rng.choice({3: [1800, 2400, 2700, 3000], 2: [1800, 2400, 2700, 3000], 4: [4000, 6000, 8000], 5: [10000, 12000, 16000], 13: [10000, 12000, 16000],
                             1: [200, 500, 1000], 0: [200, 500, 1000]}.get(mode, [BW[mode]])))
        frate = rxfilter.get_filter_rate(fs, mode, BW[mode])
        fI, fQ = rxfilter.make_filter_coef(frate, int(rng.choice([0, 0, 193, 325, 1025])) or None, bw, rxfilter.get_filter_center(NAMES[mode], bw))
        api.set_filters(fI, fQ, BW[mode]); ref.set_filters(fI, fQ, BW[mode]); return ("set_filters", bw, len(fI))
    if k == 3:
        lvl = float(rng.choice([5.0, 20.0, 60.0, 150.0]))
        api.set_agc(lvl); ref.set_agc(lvl); return ("set_agc", lvl)
    if k == 4:
        lvl = int(rng.integers(0, 4)) if mode not in (5, 13) else 0          # (FM: behind a blanked stretch the discriminator takes arg() of rounding-level numbers)
        api.set_noise_blanker(lvl); ref.set_noise_blanker(lvl); return ("set_noise_blanker", lvl)
    if k == 5:
        on = int(rng.integers(0, 2))
        api.set_auto_notch(on); ref.set_auto_notch(on); return ("set_auto_notch", on)
    if k == 6:
        inv = int(rng.integers(0, 2))
        api.invert_spectrum(inv); ref.invert_spectrum(inv); return ("invert_spectrum", inv)
    if k == 7:
        kill = int(rng.integers(0, 4) == 0)
        api.set_kill_audio(kill); ref.set_kill_audio(kill); return ("set_kill_audio", kill)
    if k == 8:
        lvl = float(rng.uniform(-90.0, -30.0))
        api.set_squelch(lvl); ref.set_squelch(lvl); return ("set_squelch", lvl)
    if k == 9:
        en, lvl = int(rng.integers(0, 2)), int(rng.integers(1, 10))
        api.set_ssb_squelch(en, lvl); ref.set_ssb_squelch(en, lvl); return ("set_ssb_squelch", en, lvl)
    if k == 10:
        sp = int(rng.integers(0, 5))
        api.set_split_rxtx(sp); ref.set_split_rxtx(sp); return ("set_split_rxtx", sp)
    if k == 11:                                        # the key: down for a call or two, then up (sidetone or silence, then the envelope)
        down = int(rng.integers(0, 2))
        args = (down, down if mode == 1 else 0, int(rng.integers(0, 2)), 0)
        api.set_key_state(*args); ref.set_key_state(*args); return ("set_key_state",) + args
    if k == 12:
        f = int(rng.choice([0, 0, 7000, -12000, 21000])) if mode in (5, 13) else 0      # (-40 dB of full scale: an overload in the other modes)
        api.add_tone(f); ref.add_tone(f); return ("add_tone", f)
    # the played sub-receiver (USB walks: it is fed every call, quisk.c:2589-2629): bank 1 with aux1TuneVector and filter set 1
    if k == 13:
        ch = int(rng.choice([-1, 1]))
        api.set_multirx_play_channel(ch); ref.set_multirx_play_channel(ch); return ("set_multirx_play_channel", ch)
    if k == 14:
        m = int(rng.integers(0, 3))
        api.set_multirx_play_method(m); ref.set_multirx_play_method(m); return ("set_multirx_play_method", m)
    f = int(rng.integers(-30000, 30000))
    api.set_multirx_freq(1, f); ref.set_multirx_freq(1, f); return ("set_multirx_freq", f)


@pytest.mark.parametrize("seed,mode,fs,play", [(1, 3, 192000, 48000), (2, 3, 111111, 96000), (3, 4, 96000, 48000), (4, 5, 192000, 48000),
                                               (5, 3, 48000, 48000), (6, 1, 133333, 48000), (7, 4, 185185, 96000), (8, 5, 96000, 192000),
                                               (9, 3, 192000, 192000), (10, 1, 48000, 96000), (11, 3, 370370, 48000), (12, 5, 53333, 48000),
                                               (21, 0, 96000, 48000), (22, 2, 192000, 48000), (23, 7, 192000, 96000), (24, 8, 111111, 48000), (25, 9, 192000, 48000),
                                               (26, 13, 96000, 48000), (27, 10, 48000, 48000),
                                               (7120, 3, 111111, 96000),      # (round 4's open mismatch: ssb_squelch reads filter_bandwidth[0] in every bank, quisk.c:1120)
                                               (910087, 4, 185185, 96000)])   # (a sweep of round 6: cFracDecim's dindex is a running sum -- 61 728 -> 48 000 brings it back to 2.0 every
                                                                              #  500 outputs, and a closed form put one such output into the call before: 4512 samples for 4510)
def test_random_setter_walk_over_the_one_receiver_api(qh, oracle, seed, mode, fs, play):
    rng = np.random.default_rng(7000 + seed)
    api = qh.quiskapi
    api.open(fs, fft_size=2048, data_width=512, playback_rate=play)
    ref = oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())
    graph = oracle.OracleGraph(2048, 512, float(fs))     # the panadapter's feed behind tone, inversion and blanker (quisk.c:2454-2475)
    ref.set_graph(graph)
    obs = np.random.default_rng(70000 + seed)            # (the observers draw from a generator of their own: the walks stay the walks they were)
    twin = None
    if os.environ.get("QH_TWIN"):                        # diagnostics: a second restatement fed 1e-13 of noise per sample, the same setters
        twin = oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())
        ref_only, ref = ref, _Both(ref, twin)
        pert = np.random.default_rng(int(os.environ.get("QH_TWIN_SEED", "3")))
    try:
        st = {"rx": 8300, "tx": 9100}
        fI, fQ = _filters(mode, fs)
        for o in (api, ref):
            o.set_rx_mode(mode); o.set_filters(fI, fQ, BW[mode]); o.set_agc(20.0)
        api.set_tune2(st["rx"], st["tx"]); ref.set_tune(st["rx"], st["tx"])
        api.set_sidetone(0.3, 600, play, 20); ref.set_sidetone(0.3, 600, 20)
        if mode == 3:
            gI, gQ = _filters(mode, fs, 2400)
            for o in (api, ref):
                o.set_filters(gI, gQ, 2400, 1); o.set_filters(fI, fQ, BW[mode])        # filter set 1, then the one global sizeFilter back (quisk.c:4591)
                o.set_multirx_mode(1, 3); o.set_multirx_freq(1, -15000); o.set_multirx_play_method(1)
        ratio = max(1, fs // 48000)
        sizes = [int(rng.choice([1, 2, 3, 5, 8])) * int(rng.integers(300, 1700)) * ratio for _ in range(24)]
        sizes = [min(s, 52000, 50000 * fs // play, 11000 * (fs // 48000 or 1)) for s in sizes]          # (the reference's interpolators stop at 52 800 outputs per call, its Buffer2Chan at 12 000 audio samples)
        n = sum(sizes)
        x = _signal(mode, 0, n, fs, float(st["rx"]), amp=2.0 ** 18)
        x[5000::9973] += 2.0 ** 21
        x[n // 2:n // 2 + n // 6] *= 0.01
        xs = _signal(mode, 1, n, fs, -15000.0, amp=2.0 ** 18) if mode == 3 else None
        log, pos, outs, loose_left = [], 0, 0, 0
        for k, s in enumerate(sizes):
            if k:
                for _ in range(int(rng.integers(1, 3))):
                    log.append((k, _draw(rng, mode, fs, play, api, ref, st)))
                    if mode in (5, 13) and log[-1][1][0] == "set_split_rxtx":
                        # the second FM receiver starts on an empty delay line: arg() of rounding-level numbers again, for as long as at the walk's
                        # own start (in output samples, not calls: a short call ends inside it)
                        loose_left = 6 * 1024 * (play // 48000) + 1024
            seg = x[pos:pos + s]
            if xs is not None:
                api.multirx_samples(1, xs[pos:pos + s]); ref.multirx_samples(1, xs[pos:pos + s])
            pos += s
            if twin:
                y, want = api.process(seg), ref_only.process(seg)
                t2 = twin.process(seg * (1.0 + 1e-13 * pert.standard_normal(s)))
                if want.size and t2.size == want.size:
                    sc = max(np.abs(want).max(), 1.0)
                    print("call %d: library %.2e, the restatement against its twin %.2e of the call's maximum" % (k, np.abs(y - want).max() / sc if y.size == want.size else -1.0, np.abs(t2 - want).max() / sc), flush=True)
            else:
                y, want = api.process(seg), ref.process(seg)
            assert y.size == want.size, (seed, k, y.size, want.size, log)
            assert api.squelch_flags() == (ref_only if twin else ref).squelch_flags(), (seed, k, log)
            if obs.integers(0, 4) == 0:                # get_graph (quisk.c:5142) now and then: the average starts over on both sides
                zoom, deltaf = float(obs.choice([1.0, 1.0, 2.0, 4.0])), float(obs.choice([0.0, 0.0, 5000.0, -12000.0]))
                got_g, want_g = api.get_graph(zoom, deltaf), graph.get(zoom, deltaf)
                assert (got_g is None) == (want_g is None), (seed, k)
                if got_g is not None:
                    assert got_g[2] == want_g[2], (seed, k, got_g[2], want_g[2])
                    assert np.abs(got_g[0] - want_g[0]).max() < 1e-6 and abs(got_g[1] - want_g[1]) < 1e-6, (seed, k, np.abs(got_g[0] - want_g[0]).max())
            if want.size == 0:
                continue
            settle = 6 * 1024 * (play // 48000) if mode in (5, 13) else 0                    # FM: arg() of rounding-level numbers while the filters fill
            lo = min(want.size, max(0, settle - outs))
            outs += want.size
            scale = max(np.abs(want).max(), 1.0)
            err = np.abs(y[lo:] - want[lo:]).max() / scale if want.size > lo else 0.0
            loose = loose_left > 0
            if np.abs(want).max() > 0.0:                 # (a key held down gives silence and stops the receivers: their run-in goes on afterwards)
                loose_left -= want.size
            assert err < (1e-4 if loose else 1e-6), "seed %d call %d (%d samples): max error %.2e of %.3e; setters %r" % (seed, k, s, err, scale, log)
    finally:
        api.close()


def test_set_auto_notch_leaves_the_rit_to_set_sidetone(qh, oracle):
    """The reference's set_auto_notch takes the flag alone (quisk.c:4596); rit_freq is the global set_sidetone writes (quisk.c:4712), and it
    tunes the split receiver too (quisk.c:2538).  Found by the walks above: qh_quisk_set_auto_notch used to write its second argument
    over it, and the second receiver of a split sat 600 Hz off from then on."""
    fs, play, mode = 133333, 48000, 1
    api = qh.quiskapi
    api.open(fs, playback_rate=play)
    ref = oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())
    try:
        fI, fQ = _filters(mode, fs)
        for o in (api, ref):
            o.set_rx_mode(mode); o.set_filters(fI, fQ, BW[mode]); o.set_agc(20.0)
        api.set_tune2(8300, 9100); ref.set_tune(8300, 9100)
        api.set_sidetone(0.3, 600, play, 20); ref.set_sidetone(0.3, 600, 20)
        n = 8000
        x = _signal(mode, 0, 10 * n, fs, 8300.0, amp=2.0 ** 18)
        for k in range(10):
            if k == 3:
                api.set_auto_notch(1); ref.set_auto_notch(1)
            if k == 5:
                api.set_split_rxtx(4); ref.set_split_rxtx(4)          # the Tx-frequency receiver alone, on both ears
            y, want = api.process(x[k * n:(k + 1) * n]), ref.process(x[k * n:(k + 1) * n])
            assert y.size == want.size
            assert np.abs(y - want).max() <= 1e-8 * max(np.abs(want).max(), 1.0), k
    finally:
        api.close()


class _Both:
    """a setter goes to both of two restatements (QH_TWIN)"""
    def __init__(self, a, b): self._a, self._b = a, b
    def __getattr__(self, name):
        fa, fb = getattr(self._a, name), getattr(self._b, name)

        def call(*args):
            fb(*args)
            return fa(*args)
        return call


def _draw_wdsp(rng, names, ch, st):
    """a setter on the WDSP channel in the audio path (levels untouched: process_agc behind it must stay under its limiter)"""
    k = int(rng.integers(0, 8))
    if k == 0:
        lo = float(rng.uniform(100.0, 600.0)); hi = lo + float(rng.uniform(1200.0, 3200.0))
        args = ("RXASetPassband", lo, hi)
    elif k == 1:
        args = ("RXASetNC", int(rng.choice([256, 512, 1024, 2048])))
    elif k == 2:
        args = ("RXASetMP", int(rng.integers(0, 2)))
    elif k == 3:
        args = ("RXANBPSetRun", int(rng.integers(0, 2)))
    elif k == 4:
        args = ("SetRXAShiftRun", int(rng.integers(0, 2)))
    elif k == 5:
        args = ("SetRXAShiftFreq", float(rng.uniform(-400.0, 400.0)))
    elif k == 6:
        args = ("SetRXAPanelGain1", float(rng.uniform(0.2, 1.0)))
    else:
        st["in_use"] ^= 1
        return ("in_use", st["in_use"])
    getattr(names, args[0])(*args[1:]); getattr(ch, args[0])(*args[1:])
    return args


@pytest.mark.parametrize("seed,mode,fs,play", [(41, 3, 192000, 48000), (42, 3, 96000, 96000), (43, 4, 192000, 48000), (44, 1, 48000, 48000),
                                               (45, 5, 192000, 96000), (46, 2, 111111, 48000), (47, 3, 370370, 192000), (48, 4, 53333, 48000)])
def test_random_setter_walk_with_wdsp_in_the_audio_path(qh, oracle, seed, mode, fs, play):
    """The same walk with quisk.c:2660-2661 live: the 48 ksps audio goes through wdspFexchange0 -- this library's shim, double rings, slews
    and RXA engine on device buffers (qh_wdsp_fexchange0_device inside qh_quisk_process_samples), the restated shim in front of the
    restated channel on the other side -- while Quisk's setters, the WDSP channel's own (passband, nc, minimum phase, shift, notch
    filter, panel gain) and in_use change between ragged calls."""
    import ctypes as C
    from test_gpu_wdsp_names_fuzz import _Names
    rng = np.random.default_rng(7000 + seed)
    lib = qh.load()
    D = C.c_double
    api = qh.quiskapi
    api.open(fs, playback_rate=play)
    ref = oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())
    lib.OpenChannel(0, 256, 256, 48000, 48000, 48000, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    assert lib.qh_wdsp_status() == 0, lib.qh_last_error()
    names = _Names(lib, 0, 256)
    ch = oracle.WdspChannel(256, 256, 48000, 48000, 48000)
    for t in (names, ch):
        t.SetRXAShiftRun(0); t.RXANBPSetRun(0); t.SetRXAMode(1); t.RXASetPassband(200.0, 3400.0); t.RXASetNC(256); t.RXASetMP(0)
        t.SetRXAAGCMode(0); t.SetRXAAGCFixed(0.0)
    lib.SetRXAAMSQRun(0, 0); lib.SetRXAPanelRun(0, 0); lib.SetRXAEMNRRun(0, 0)
    lib.qh_wdsp_set_parameter(0, 256, 0)
    shim = oracle.OracleWdspShim(lambda pin, pout: 0)
    shim.set_parameter(in_size=256, in_use=0)
    ref.set_wdsp(shim, ch)
    twin = None
    if os.environ.get("QH_TWIN"):                        # diagnostics: a second restatement fed 1e-13 of noise per sample, the same setters
        twin = oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())
        ch2 = oracle.WdspChannel(256, 256, 48000, 48000, 48000)
        ch2.SetRXAShiftRun(0); ch2.RXANBPSetRun(0); ch2.SetRXAMode(1); ch2.RXASetPassband(200.0, 3400.0); ch2.RXASetNC(256); ch2.RXASetMP(0)
        ch2.SetRXAAGCMode(0); ch2.SetRXAAGCFixed(0.0)
        shim2 = oracle.OracleWdspShim(lambda pin, pout: 0)
        shim2.set_parameter(in_size=256, in_use=0)
        twin.set_wdsp(shim2, ch2)
        ref_only, ref, ch = ref, _Both(ref, twin), _Both(ch, ch2)
        pert = np.random.default_rng(3)
    try:
        st = {"rx": 8300, "tx": 9100, "in_use": 0}
        fI, fQ = _filters(mode, fs)
        for o in (api, ref):
            o.set_rx_mode(mode); o.set_filters(fI, fQ, BW[mode]); o.set_agc(20.0)
        api.set_tune2(st["rx"], st["tx"]); ref.set_tune(st["rx"], st["tx"])
        api.set_sidetone(0.3, 600, play, 20); ref.set_sidetone(0.3, 600, 20)
        ratio = max(1, fs // 48000)
        sizes = [int(rng.choice([1, 2, 3, 5, 8])) * int(rng.integers(300, 1700)) * ratio for _ in range(24)]
        sizes = [min(s, 52000, 50000 * fs // play, 11000 * (fs // 48000 or 1)) for s in sizes]
        n = sum(sizes)
        x = _signal(mode, 0, n, fs, float(st["rx"]), amp=2.0 ** 18)
        x[n // 2:n // 2 + n // 6] *= 0.01
        log, pos, outs, loose_left = [], 0, 0, 0
        for k, s in enumerate(sizes):
            if k == 2:                                   # WDSP on once the audio is running: its up-slew starts at the first non-zero sample, and
                st["in_use"] = 1                         # FFT filters leave 1e-15 where FIR loops leave 0 (tests/test_gpu_quisk_process_samples.py)
                log.append((k, ("in_use", 1)))
            if k > 2:
                for _ in range(int(rng.integers(0, 3))):
                    log.append((k, _draw(rng, mode, fs, play, api, ref, st) if rng.integers(0, 2) else _draw_wdsp(rng, names, ch, st)))
                    if mode in (5, 13) and log[-1][1][0] == "set_split_rxtx":
                        loose_left = 6 * 1024 * (play // 48000) + 1024      # (the second FM receiver starts on an empty delay line, as in the walk above)
            lib.qh_wdsp_set_parameter(0, -1, st["in_use"]); shim.set_parameter(in_use=st["in_use"])
            seg = x[pos:pos + s]
            pos += s
            if twin:
                shim2.set_parameter(in_use=st["in_use"])
                y, want = api.process(seg), ref_only.process(seg)
                t2 = twin.process(seg * (1.0 + 1e-13 * pert.standard_normal(s)))
                if want.size:
                    d = np.abs(y - want) / max(np.abs(want).max(), 1.0)
                    off = np.nonzero(d > 1e-9)[0]
                    print("call %d: engine %.2e, the restatement against its twin %.2e of the call's maximum; %d of %d samples off by more than 1e-9%s" % (
                        k, d.max(), np.abs(t2 - want).max() / max(np.abs(want).max(), 1.0), off.size, d.size,
                        " (samples %d .. %d, worst at %d; left %.2e right %.2e)" % (off[0], off[-1], int(d.argmax()), np.abs(y.real - want.real).max(), np.abs(y.imag - want.imag).max()) if off.size else ""), flush=True)
            else:
                y, want = api.process(seg), ref.process(seg)
            assert lib.qh_wdsp_status() == 0, (seed, k, lib.qh_last_error())
            assert y.size == want.size, (seed, k, y.size, want.size, log)
            if want.size == 0:
                continue
            settle = 6 * 1024 * (play // 48000) if mode in (5, 13) else 0
            lo = min(want.size, max(0, settle - outs))
            outs += want.size
            scale = max(np.abs(want).max(), 1.0)
            err = np.abs(y[lo:] - want[lo:]).max() / scale if want.size > lo else 0.0
            loose = loose_left > 0
            if np.abs(want).max() > 0.0:                 # (a key held down gives silence and stops the receivers: their run-in goes on afterwards)
                loose_left -= want.size
            assert twin or err < (1e-3 if loose else 1e-6), "seed %d call %d (%d samples): max error %.2e of %.3e; setters %r" % (seed, k, s, err, scale, log)
    finally:
        lib.qh_wdsp_set_parameter(0, -1, 0)
        lib.wdspFexchange0(0, None, 0)
        lib.CloseChannel(0)
        api.close()
