#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (3, 4, 5) with the kernels this round ships.  One JSON
line per configuration; results are recorded in profiles/ and DESIGN.md (bench.py stays the one-line contract for
configuration 2).  Runs on one GPU."""
import json
import os
import sys
import time

import numpy as np
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, sync, steps=10, warmup=2):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def new_stream(torch, dev):
    """A stream of the leg's own.  (torch's current stream is the null stream, whose handle 0 tells an engine to make a private
    non-blocking stream: engines created that way run side by side, which only independent engines may.)"""
    return torch.cuda.Stream(dev)


def c3_taps():
    k = np.arange(1023) - 511
    return np.sinc(k / 32.0) / 32.0 * np.blackman(1023)


def setup_config3(torch, qh, dev, nch=64, n=1 << 20):
    """The engines, buffers and calls of the config 3 leg.  tests/test_gpu_bench_shapes.py checks results through this very function,
    so that what is timed and what is parity-tested cannot drift apart."""
    fs = 1536000.0
    taps = c3_taps()
    L = SimpleNamespace(nch=nch, n=n, fs=fs, taps=taps)
    # the FIR and the panadapter are independent consumers of the same samples: a stream each, side by side
    L.streams = [new_stream(torch, dev), new_stream(torch, dev)]
    s = L.streams[0].cuda_stream
    L.bank = qh.FirBank(nch, taps, 32, stream=s)
    L.pan = qh.Panadapter(nch, 16384, 1024, fs, stream=L.streams[1].cuda_stream)
    L.x = (torch.randn((nch, n), dtype=torch.float64, device=dev) + 1j * torch.randn((nch, n), dtype=torch.float64, device=dev)) * 2.0 ** 20
    L.y = torch.empty((nch, n // 32), dtype=torch.complex128, device=dev)
    # the same work with the FIR taken out of the panadapter's own transform (qh_pan_attach_fir: the stream is read once)
    L.fused = qh.Panadapter(nch, 16384, 1024, fs, stream=s)
    L.fused.attach_fir(taps, 32)
    L.y_fused = torch.empty_like(L.y)
    L.step_fir = lambda: L.bank.process_ptr(L.x.data_ptr(), n, n, L.y.data_ptr(), n // 32)
    L.step_pan = lambda: L.pan.feed_ptr(L.x.data_ptr(), n, n)
    L.step_both = lambda: (L.step_fir(), L.step_pan())
    L.step_fused = lambda: L.fused.feed_decimate_ptr(L.x.data_ptr(), n, n, L.y_fused.data_ptr(), n // 32)
    torch.cuda.synchronize(dev)         # the inputs are made on torch's stream, the engines run on streams of their own
    return L


def config3(torch, qh, dev):
    L = setup_config3(torch, qh, dev)
    sync = lambda: torch.cuda.synchronize(dev)
    t_fir = timed(L.step_fir, sync)
    t_pan = timed(L.step_pan, sync)
    t_both = timed(L.step_both, sync)
    t_fused = timed(L.step_fused, sync)
    tot = L.nch * L.n
    return {"config": "3: 64 ch x 1.536 Msps fp64, 1023-tap FIR /32 + 16384-pt panadapter every block", "samples_per_step": tot,
            "fir_ms": t_fir * 1e3, "pan_ms": t_pan * 1e3, "both_ms": t_both * 1e3, "fused_ms": t_fused * 1e3,
            "best_ms": min(t_both, t_fused) * 1e3, "Msamp_per_s": tot / min(t_both, t_fused) / 1e6,
            "fir_Msamp_per_s": tot / t_fir / 1e6, "pan_Msamp_per_s": tot / t_pan / 1e6,
            "algorithmic_GBps": 16.5 * tot / min(t_both, t_fused) / 1e9,
            "note": "16 B in + 16/32 B FIR out per sample; only the running |X| average leaves the chip (SURVEY.md 8(d): 16.5 B)"}


C4_MODES = [1, 6, 5]                      # USB / AM / FM by c mod 3
C4_KINDS = {1: "usb", 6: "am", 5: "fm"}
C4_PASSBAND = {1: (300.0, 3000.0), 6: (-4000.0, 4000.0), 5: (-8000.0, 8000.0)}


def c4_set_modes(eng, c):
    """Channel c's mode and passband (an RxaEngine with its channel index first, or the oracle's WdspChannel through a one-line adapter)."""
    m = C4_MODES[c % 3]
    eng.SetRXAMode(c, m)
    eng.RXASetPassband(c, *C4_PASSBAND[m])


def c4_common(eng, c, shift):
    eng.SetRXAShiftRun(c, 1); eng.SetRXAShiftFreq(c, shift); eng.RXANBPSetRun(c, 1)
    eng.SetRXAAGCMode(c, 0); eng.SetRXAAGCFixed(c, 0.0)


def setup_config4(torch, qh, dev, nch=256, nblk=None, modes_now=True, first=0):
    """modes_now=False (the parity test's acquisition form): every channel starts as USB 300..3000 and the caller switches the
    detectors in with c4_set_modes once the filters hold signal.  first: the job-wide index of this engine's channel 0 (a rank of
    bench.py --config 4 --gpus N owns channels first .. first + nch - 1: mode, shift and signal go by the job-wide index)."""
    from quisk_amd import synth
    nblk = nblk or int(os.environ.get("QH_C4_NBLK", "4096"))       # DSP blocks per call: 2^22 input samples per channel per step
    n_in = nblk * 1024
    L = SimpleNamespace(nch=nch, nblk=nblk, n_in=n_in, n_out=nblk * 256)
    L.stream = new_stream(torch, dev)
    L.eng = eng = qh.RxaEngine(nch, device=dev.index or 0, stream=L.stream.cuda_stream)

    class _Shifted:                     # c4_set_modes / c4_common take the channel index that decides the mode: the job-wide one
        def __getattr__(self, name):
            f = getattr(eng, name)
            return lambda c, *a: f(c - first, *a)
    view = _Shifted() if first else eng
    for c in range(first, first + nch):
        c4_common(view, c, synth.shift_freq(c))
        if modes_now:
            c4_set_modes(view, c)
        else:
            view.SetRXAMode(c, 1); view.RXASetPassband(c, 300.0, 3000.0)
    # USB channels get the two-tone input of C2, AM channels a carrier with m = 0.5 / 1 kHz, FM channels a
    # carrier with a 1 kHz tone at +-3 kHz deviation, all + noise (synth.make_mode_input_numpy)
    L.x = synth.make_mode_input_torch([C4_KINDS[C4_MODES[c % 3]] for c in range(first, first + nch)], n_in, dev, first_channel=first)
    L.y = torch.empty((nch, L.n_out), dtype=torch.complex128, device=dev)
    L.step = lambda: eng.process_ptr(L.x.data_ptr(), n_in, L.y.data_ptr(), L.n_out, nblk)
    torch.cuda.synchronize(dev)         # the inputs are made on torch's stream, the engines run on streams of their own
    return L


def config4(torch, qh, dev):
    L = setup_config4(torch, qh, dev)
    eng = L.eng
    sync = lambda: torch.cuda.synchronize(dev)
    t = timed(L.step, sync, steps=5, warmup=1)
    # the same call with the launch sequence replayed from hipGraphs (qh_rxa_set_graph_replay): ~25 launches per call
    eng.set_graph_replay(True)
    tg = timed(L.step, sync, steps=8, warmup=4)
    eng.set_graph_replay(False)
    eng.enable_timing(True)
    L.step()
    kt = eng.timing_ms()
    eng.enable_timing(False)
    tot = L.nch * L.n_in
    return {"front_ms": kt[0], "band_ms": kt[1], "rest_ms": kt[2],
            "config": "4 (one GPU's share): 256 ch x 192 k, mode by c mod 3 = USB / AM / FM, fp64", "samples_per_step": tot,
            "ms": t * 1e3, "Msamp_per_s": tot / t / 1e6, "ms_graph_replay": tg * 1e3, "Msamp_per_s_graph_replay": tot / tg / 1e6,
            "graph_launches": eng.graph_launches(),
            "pll_tiles_rerun": eng.pll_repairs(),
            "note": "AM / FM / notch recurrences time-tiled (qh_tiled.hpp); pll_tiles_rerun = FM tiles the verify pass re-ran sequentially over all steps"}


def setup_config2_agc(torch, qh, dev, nch=256, nblk=None, fading=False):
    """BASELINE config 2's chain with WDSP's AGC state machine running (SetRXAAGCMode 3, the mode Quisk's WDSP path sets by default)
    instead of the fixed gain the configuration specifies: the level detector in time tiles (qh_agc_tiled.hpp)."""
    from quisk_amd import synth
    nblk = nblk or int(os.environ.get("QH_C2A_NBLK", "4096"))
    n_in = nblk * 1024
    L = SimpleNamespace(nch=nch, nblk=nblk, n_in=n_in, n_out=nblk * 256)
    # the same buffer is fed every step: the tones sit on the buffer's frequency grid (moved by < 0.023 Hz), so that the steps are one
    # continuous stream -- a phase jump per call is a click the AGC answers for seconds, which no receiver's input has
    L.x = synth.make_mode_input_torch(["usb"] * nch, n_in, dev, periodic=True)
    if fading:
        # overs of a contact: the channel is loud for a few seconds, then 40 dB down for a few (3 .. 5 changes of level per 21.8 s
        # buffer, another rhythm per channel, periodic in the buffer).  After every DROP the level detector decays for seconds
        # (tau_decay), which is where the tiled AGC's warm-ups miss and a segment is walked again in order (agc_bounds_fix_kernel)
        t = torch.arange(n_in, dtype=torch.float64, device=dev)
        for c in range(nch):
            cyc = 3 + (c % 3)
            env = torch.where(torch.remainder(t * (cyc / n_in) + 0.07 * (c % 11), 1.0) < 0.5, 1.0, 0.01)
            L.x[c] *= env
        del t
    L.y = torch.empty((nch, L.n_out), dtype=torch.complex128, device=dev)
    L.stream = new_stream(torch, dev)
    L.eng = e = qh.RxaEngine(nch, stream=L.stream.cuda_stream)
    for c in range(nch):
        c2agc_setters(e, c, synth.shift_freq(c))
    L.step = lambda: e.process_ptr(L.x.data_ptr(), n_in, L.y.data_ptr(), L.n_out, nblk)
    torch.cuda.synchronize(dev)         # the inputs are made on torch's stream, the engines run on streams of their own
    return L


def c2agc_setters(e, c, shift):
    e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, shift); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
    e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 3)


def config2_agc(torch, qh, dev):
    L = setup_config2_agc(torch, qh, dev)
    e = L.eng
    sync = lambda: torch.cuda.synchronize(dev)
    t = timed(L.step, sync, steps=4, warmup=2)
    tot = L.nch * L.n_in
    n_log2 = L.n_in.bit_length() - 1
    tiles, segs = e.agc_repairs(), e.agc_segments_rerun()
    del L, e
    torch.cuda.empty_cache()
    # the same leg on an input full of fades
    Lf = setup_config2_agc(torch, qh, dev, fading=True)
    tf = timed(Lf.step, sync, steps=4, warmup=2)
    fading = {"ms": tf * 1e3, "Msamp_per_s": tot / tf / 1e6, "agc_tiles_rerun": Lf.eng.agc_repairs(), "agc_segments_rerun": Lf.eng.agc_segments_rerun(),
              "input": "the same tones keyed between full level and -40 dB every 2 .. 4 s (3 .. 5 overs per 21.8 s call)"}
    return {"config": "2 with the AGC state machine on (SetRXAAGCMode 3): 256 ch x 192 k SSB RXA, fp64, 2^%d samples per channel and step" % (n_log2),
            "samples_per_step": tot, "ms": t * 1e3, "Msamp_per_s": tot / t / 1e6, "agc_tiles_rerun": tiles,
            "agc_segments_rerun": segs, "fading_input": fading,
            "note": "not a BASELINE configuration (config 2 fixes the gain); agc_tiles_rerun = tiles whose boundary state the exact pass corrected"}


def c5_bandpass_taps():
    """WDSP bandpass 300..3000 at 48 k, nc 2048, BH-4 (fir_bandpass, wdsp/fir.c:187-254) as linear-convolution taps for a FirBank."""
    m = 0.5 * 2047
    pos = np.arange(2048) - m
    c = np.cos(np.pi / m * np.arange(2048))
    win = 0.21747 + c * (-0.45325 + c * (0.28256 + c * (-0.04672)))
    ft = (3000.0 - 300.0) / (2 * 48000.0)
    return np.sin(2 * np.pi * ft * pos) / (np.pi * pos) * win * np.exp(-1j * np.pi * 3300.0 / 48000.0 * pos)


def setup_config5(torch, qh, dev, n=1 << 26, unfused=True):
    """61.44 Msps stream, ~1.09 s of signal, fp32: the fused half-band cascade, the 245-tap /5 and the bandpass (step_fused, what the
    driver's line reports) and, with unfused, the eight half-band banks one after the other (step)."""
    tabs = __import__("quisk_amd.rxfilter", fromlist=["x"]).coefficient_tables()
    L = SimpleNamespace(n=n, taps245=tabs["quiskFilt240D5CoefsSharp"], bp=c5_bandpass_taps())
    L.stream = new_stream(torch, dev)           # one stream: every stage reads what the stage before it wrote
    s = L.stream.cuda_stream
    di = dev.index or 0
    L.d5 = d5 = qh.FirBank(1, L.taps245, 5, dtype=1, device=di, stream=s)
    L.core = core = qh.FirBank(1, L.bp, 1, dtype=1, device=di, stream=s)
    L.x = x = torch.randn((1, n), dtype=torch.float32, device=dev) + 1j * torch.randn((1, n), dtype=torch.float32, device=dev)
    L.bufs = bufs = [torch.empty((1, n >> (k + 1)), dtype=torch.complex64, device=dev) for k in (range(8) if unfused else (7,))]
    L.y5 = y5 = torch.empty((1, (n >> 8) // 5 + 8), dtype=torch.complex64, device=dev)
    L.yo = yo = torch.empty_like(y5)
    L.casc = casc = qh.HalfBandCascade(1, 8, dtype=1, device=di, stream=s)
    if unfused:
        L.hb = hb = [qh.FirBank(1, qh.hb45_taps(), 2, dtype=1, device=di, stream=s) for _ in range(8)]

        def step():
            cur, cn = x, n
            for k in range(8):
                m_ = hb[k].process_ptr(cur.data_ptr(), cur.shape[1], cn, bufs[k].data_ptr(), bufs[k].shape[1])
                cur, cn = bufs[k], m_
            m_ = d5.process_ptr(cur.data_ptr(), cur.shape[1], cn, y5.data_ptr(), y5.shape[1])
            return core.process_ptr(y5.data_ptr(), y5.shape[1], m_, yo.data_ptr(), yo.shape[1])
        L.step = step

    def step_fused(src=None, nn=None):
        src = x.data_ptr() if src is None else src
        nn = n if nn is None else nn
        m_ = casc.process_ptr(src, nn, nn, bufs[-1].data_ptr(), bufs[-1].shape[1])
        m_ = d5.process_ptr(bufs[-1].data_ptr(), bufs[-1].shape[1], m_, y5.data_ptr(), y5.shape[1])
        return core.process_ptr(y5.data_ptr(), y5.shape[1], m_, yo.data_ptr(), yo.shape[1])
    L.step_fused = step_fused
    torch.cuda.synchronize(dev)         # the inputs are made on torch's stream, the engines run on streams of their own
    return L


def config5(torch, qh, dev):
    L = setup_config5(torch, qh, dev)
    n, x, bufs = L.n, L.x, L.bufs
    sync = lambda: torch.cuda.synchronize(dev)
    t = timed(L.step, sync, steps=10, warmup=2)
    t1 = timed(lambda: L.hb[0].process_ptr(x.data_ptr(), n, n, bufs[0].data_ptr(), bufs[0].shape[1]), sync, steps=10, warmup=2)
    tf = timed(L.step_fused, sync, steps=10, warmup=2)
    tc = timed(lambda: L.casc.process_ptr(x.data_ptr(), n, n, bufs[7].data_ptr(), bufs[7].shape[1]), sync, steps=10, warmup=2)
    return {"fused_ms": tf * 1e3, "fused_Msamp_per_s": n / tf / 1e6, "fused_cascade_only_ms": tc * 1e3,
            "fused_algorithmic_GBps": 8.0 * n / tf / 1e9,
            "config": "5 (one GPU's channel): 1 ch x 61.44 Msps fp32, 8 x HB45 + 245-tap /5 + bandpass nc 2048, 2^26 samples per step",
            "samples_per_step": n, "ms": t * 1e3, "Msamp_per_s": n / t / 1e6, "first_stage_ms": t1 * 1e3,
            "algorithmic_GBps": 8.0 * n / t / 1e9,
            "note": "unfused cascade of overlap-save banks; the first half-band alone moves 12 B/sample"}


QN_MODES = (("USB", 3, 2700), ("AM", 4, 6000), ("FM", 5, 12000))      # name, Quisk mode number (quisk.h:55-70), bandwidth
QN_AGC_GAIN = 5000.0
QN_PIECES = 4                   # time pieces of a call of the whole-function bank (qh_qps_set_pieces)


def qn_pieces(n, pieces=QN_PIECES):
    """the piece lengths qh_qps_process cuts a call of n samples into"""
    per = ((n + pieces - 1) // pieces + 63) // 64 * 64
    return [min(per, n - pos) for pos in range(0, n, per)]


def qn_tune(c):
    return 1000 * (c % 40) - 20000


def setup_quisk_native(torch, qh, dev, name, nch=256, n=1 << 20, whole=True, nb=0, fft_size=0, pieces=QN_PIECES, pipelined=False):
    """One mode of the Quisk-native leg: 256 receivers, 192 ksps in, 48 ksps out.
    whole=True: the WHOLE of quisk_process_samples for the bank (qh_qps_*: test tone / inversion / NoiseBlanker when set, the
    panadapter's feed when fft_size > 0, tune + quisk_process_decimate + quisk_process_demodulate, process_agc -- always on, as in the
    reference -- on a second stream beside the next piece's filters, squelch);
    whole=False: the receiver bank alone (tune, decimate, demodulate: qh_qrx_*), the kernels under it."""
    from quisk_amd import rxfilter
    fs = 192000
    mode, bw = {m[0]: (m[1], m[2]) for m in QN_MODES}[name]
    L = SimpleNamespace(nch=nch, n=n, fs=fs, mode=mode, bw=bw, name=name, whole=whole)
    L.stream = new_stream(torch, dev)
    if whole:
        # (no stream passed: the bank makes its own pair -- the filters' stream and the AGC's, each on CUs of its own)
        L.bank = bank = qh.QuiskProcessBank(nch, fs, mode, bw, playback_rate=48000, fft_size=fft_size, data_width=fft_size // 2 if fft_size else 0,
                                            stream=None if os.environ.get("QH_QPS_OWN_STREAM", "1") != "0" else L.stream.cuda_stream)
        bank.set_agc(QN_AGC_GAIN)
        if nb:
            bank.set_noise_blanker(nb)
        if pieces:
            bank.set_pieces(pieces)
        if pipelined:       # a streaming caller: a call returns with its AGC still running, the next call's filters start beside it
            bank.set_pipelined(1)
    else:
        L.bank = bank = qh.QuiskRxBank(nch, fs, mode, bw, stream=L.stream.cuda_stream)
    L.rate = bank.get_filter_rate()
    L.fI, L.fQ = rxfilter.make_filter_coef(L.rate, None, bw, rxfilter.get_filter_center(name, bw))
    bank.set_tune_all([qn_tune(c) for c in range(nch)])        # one launch per table for the whole bank (qh_qps_set_tune_all)
    bank.set_filters(-1, L.fI, L.fQ)
    L.x = (torch.randn((nch, n), dtype=torch.float64, device=dev) + 1j * torch.randn((nch, n), dtype=torch.float64, device=dev)) * 2.0 ** 22
    L.m = bank.out_capacity(n) if whole else bank.out_count(n) + 64
    L.y = torch.empty((nch, L.m), dtype=torch.complex128, device=dev)
    if whole:
        L.step = lambda: bank.process_ptr(L.x.data_ptr(), n, n, L.y.data_ptr(), L.m)
    else:
        L.step = lambda: bank.process_ptr(L.x.data_ptr(), n, n, L.y.data_ptr(), L.m)
    torch.cuda.synchronize(dev)         # the inputs are made on torch's stream, the engines run on streams of their own
    return L


def quisk_native(torch, qh, dev):
    """Path A: 256 receivers, USB, AM and FM, 2^20 input samples per receiver per step: the whole of quisk_process_samples for the bank
    (process_agc on, a release gain at which the limiter works: an overload ramp every few FIFO cycles), with the NoiseBlanker and with
    the panadapter's feed in the path for USB, and the receiver bank alone.  The reference's own figure for ONE receiver on a CPU core
    is 9.9 Msamp/s."""
    nch, n = 256, 1 << 20
    out = []
    sync = lambda: torch.cuda.synchronize(dev)
    only = os.environ.get("QH_QUISK_MODES", "USB,AM,FM").split(",")          # e.g. QH_QUISK_MODES=FM for a kernel trace of one mode
    for name, mode, bw in QN_MODES:
        if name not in only:
            continue
        L = setup_quisk_native(torch, qh, dev, name)
        t = timed(L.step, sync, steps=8, warmup=3)
        row = {"mode": name, "ms": t * 1e3, "Msamp_per_s": nch * n / t / 1e6, "filter_rate": L.rate, "filter_taps": int(L.fI.size)}
        del L
        L = setup_quisk_native(torch, qh, dev, name, pipelined=True)
        tq = timed(L.step, sync, steps=8, warmup=3)
        row.update({"pipelined_ms": tq * 1e3, "pipelined_Msamp_per_s": nch * n / tq / 1e6})
        del L
        L = setup_quisk_native(torch, qh, dev, name, whole=False)
        tb = timed(L.step, sync, steps=8, warmup=2)
        row.update({"bank_only_ms": tb * 1e3, "bank_only_Msamp_per_s": nch * n / tb / 1e6})
        del L
        if name == "USB":
            L = setup_quisk_native(torch, qh, dev, name, nb=1)
            tn = timed(L.step, sync, steps=6, warmup=3)
            del L
            L = setup_quisk_native(torch, qh, dev, name, fft_size=2048)
            tp = timed(L.step, sync, steps=6, warmup=3)
            del L
            row.update({"noise_blanker_on_ms": tn * 1e3, "noise_blanker_on_Msamp_per_s": nch * n / tn / 1e6,
                        "panadapter_2048_on_ms": tp * 1e3, "panadapter_2048_on_Msamp_per_s": nch * n / tp / 1e6})
        out.append(row)
    return {"config": "Quisk-native chain (path A): the whole of quisk_process_samples for 256 receivers x 192 ksps -> 48 ksps, process_agc on, "
                      "2^20 input samples per receiver per step",
            "samples_per_step": nch * n, "modes": out,
            "note": "the reference's quisk_process_samples handles one receiver per process: 9.9 Msamp/s on a CPU core (SURVEY.md 8 a10)"}


def analyzer(torch, qh, dev):
    """WDSP display engine (wdsp/analyzer.c) for a bank of 64 displays fed 2^20 samples each per step: 16384-point frames with
    50 % overlap (127 frames per display and step), Blackman-Harris window, peak detector to 2048 pixels, recursive averaging."""
    nd, n, size = 64, 1 << 20, 16384
    stream = new_stream(torch, dev)
    a = qh.AnalyzerBank(nd, size, stream=stream.cuda_stream)
    a.SetDisplaySampleRate(1536000)
    a.SetDisplayAverageMode(0, 1)
    a.SetDisplayAvBackmult(0, 0.9)
    a.SetAnalyzer(1, 1, 1, [0], size, 8192, 1, 0.0, size // 2, 0, 0.0, 0.0, 2048, 1, 0, 0.0, 0.0, 4 * size)
    x = (torch.randn((nd, n), dtype=torch.float64, device=dev) + 1j * torch.randn((nd, n), dtype=torch.float64, device=dev)) * 0.1
    sync = lambda: torch.cuda.synchronize(dev)
    frames = []
    t = timed(lambda: frames.append(a.feed_ptr(0, x.data_ptr(), n, n)), sync)
    tot = nd * n
    return {"config": "analyzer: 64 displays x 2^20 samples, 16384-point frames at 50 % overlap -> 2048 pixels", "samples_per_step": tot,
            "frames_per_display_and_step": frames[-1], "ms": t * 1e3, "Msamp_per_s": tot / t / 1e6,
            "frames_per_s": nd * frames[-1] / t, "algorithmic_GBps": 16.0 * tot / t / 1e9,
            "note": "16 B per input sample read once; the frames' 2x overlap is served from the float copy the engine keeps"}


HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def driver_legs(torch, qh, dev, emit):
    """The extra legs of bench.py's line (`other_configs`): per configuration the algorithmic bytes per input sample,
    the rate, GB/s against the 8 TB/s roofline, and the kernel that takes most of the step.  emit(key, leg) is called as each leg
    finishes, so that a later leg's fault loses nothing that was measured."""
    def leg(key, fn, bytes_per_sample, ms_key, dominant):
        try:
            r = fn(torch, qh, dev)
            ms = r[ms_key]
            rate = r["samples_per_step"] / (ms * 1e-3)
            gbps = bytes_per_sample * rate / 1e9
            name, kms = dominant(r)
            emit(key, {"workload": r["config"], "samples_per_step": r["samples_per_step"], "ms_per_step": ms, "Msamp_per_s": rate / 1e6,
                       "algorithmic_bytes_per_sample": bytes_per_sample, "algorithmic_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
                       "dominant_kernel": name, "dominant_kernel_ms": kms, "detail": {k: v for k, v in r.items() if k not in ("config",)}})
        except Exception as exc:                                 # reported, never required
            emit(key, {"failed": repr(exc)})
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()

    # config 3: 16 B in (read once by the FIR and once by the transform) + 16/32 B FIR output; the |X| sums stay on the chip and only
    # the running average leaves it, so SURVEY.md 8(d)'s 16.5 B, not 24.5
    leg("config3", config3, 16.5, "best_ms",
        lambda r: ("pan16k_kernel (16384-point panadapter, read-once)", r["pan_ms"]) if r["pan_ms"] >= r["fir_ms"]
        else ("osfir_kernel<f64,4096,D=8,pick 4> (1023-tap /32)", r["fir_ms"]))
    # config 4: the call's launch sequence replayed from a hipGraph (qh_rxa_set_graph_replay), the engine's mode for repeated calls
    leg("config4", config4, 20.0, "ms_graph_replay", lambda r: ("osfir_kernel<f64,4096,D=4,OUTMIX> front (shared by USB / AM / FM)", r.get("front_ms")))
    leg("config5", config5, 8.0, "fused_ms", lambda r: ("hb45_cascade_kernel<float,4> x 2 (4 + 4 stages)", r["fused_cascade_only_ms"]))
    # the headline chain with WDSP's AGC state machine on (not a BASELINE configuration: config 2 fixes the gain)
    leg("config2_agc_on", config2_agc, 20.0, "ms", lambda r: ("osfir front kernel as in config 2; of the AGC's seven kernels agc_lanes_kernel / agc_apply_kernel (2.0 ms each of ~9.6)", None))
    try:
        r = quisk_native(torch, qh, dev)
        emit("quisk_native", {"workload": r["config"], "samples_per_step": r["samples_per_step"], "algorithmic_bytes_per_sample": 20.0,
                              "modes": [dict({"mode": m["mode"], "ms_per_step": m["ms"], "Msamp_per_s": m["Msamp_per_s"],
                                              "algorithmic_GBps": 20.0 * m["Msamp_per_s"] / 1e3,
                                              "frac_of_hbm_peak": 20.0 * m["Msamp_per_s"] / 1e3 / HBM_PEAK_GBPS},
                                             **{k: v for k, v in m.items() if k.startswith(("bank_only", "noise_blanker", "panadapter", "pipelined"))}) for m in r["modes"]],
                              "agc_note": "ms_per_step: the whole function with process_agc (quisk.c:2162) on a second stream beside the next piece's filters; "
                                          "bank_only_*: tune + decimate + demodulate alone (qh_qrx_*); pipelined_*: the same calls when a call does not wait "
                                          "for its own AGC (qh_qps_set_pipelined: a streaming caller's form, the next call's filters beside this call's last AGC piece)",
                              "dominant_kernel": "osfir_kernel<f64,4096,D=8,OUTMIX> (tune + collapsed 1181-tap /16)"})
    except Exception as exc:
        emit("quisk_native", {"failed": repr(exc)})


def main():
    import torch
    import quisk_amd as qh
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    which = sys.argv[1:] or ["3", "4", "5"]
    if which[0] == "driver":                 # bench.py's child: one JSON object per finished leg, appended to the file named
        path = which[1]

        def emit(key, leg):
            with open(path, "a") as fh:
                fh.write(json.dumps({"key": key, "leg": leg}) + "\n")
        driver_legs(torch, qh, dev, emit)
        return
    for w in which:
        r = {"3": config3, "4": config4, "5": config5, "2agc": config2_agc, "analyzer": analyzer, "quisk": quisk_native}[w](torch, qh, dev)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
