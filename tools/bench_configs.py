#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (3, 4, 5) with the kernels this round ships.  One JSON
line per configuration; results are recorded in profiles/ and DESIGN.md (bench.py stays the one-line contract for
configuration 2).  Runs on one GPU."""
import json
import os
import sys
import time

import numpy as np

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


def config3(torch, qh, dev):
    nch, n, fs = 64, 1 << 20, 1536000.0
    k = np.arange(1023) - 511
    taps = np.sinc(k / 32.0) / 32.0 * np.blackman(1023)
    s = torch.cuda.current_stream(dev).cuda_stream
    bank = qh.FirBank(nch, taps, 32, stream=s)
    pan = qh.Panadapter(nch, 16384, 1024, fs, stream=s)
    x = (torch.randn((nch, n), dtype=torch.float64, device=dev) + 1j * torch.randn((nch, n), dtype=torch.float64, device=dev)) * 2.0 ** 20
    y = torch.empty((nch, n // 32), dtype=torch.complex128, device=dev)
    sync = lambda: torch.cuda.synchronize(dev)
    t_fir = timed(lambda: bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 32), sync)
    t_pan = timed(lambda: pan.feed_ptr(x.data_ptr(), n, n), sync)
    t_both = timed(lambda: (bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 32), pan.feed_ptr(x.data_ptr(), n, n)), sync)
    # the same work with the FIR taken out of the panadapter's own transform (qh_pan_attach_fir: the stream is read once)
    fused = qh.Panadapter(nch, 16384, 1024, fs, stream=s)
    fused.attach_fir(taps, 32)
    t_fused = timed(lambda: fused.feed_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 32), sync)
    tot = nch * n
    return {"config": "3: 64 ch x 1.536 Msps fp64, 1023-tap FIR /32 + 16384-pt panadapter every block", "samples_per_step": tot,
            "fir_ms": t_fir * 1e3, "pan_ms": t_pan * 1e3, "both_ms": t_both * 1e3, "fused_ms": t_fused * 1e3,
            "best_ms": min(t_both, t_fused) * 1e3, "Msamp_per_s": tot / min(t_both, t_fused) / 1e6,
            "fir_Msamp_per_s": tot / t_fir / 1e6, "pan_Msamp_per_s": tot / t_pan / 1e6,
            "algorithmic_GBps": 16.5 * tot / min(t_both, t_fused) / 1e9,
            "note": "16 B in + 16/32 B FIR out per sample; only the running |X| average leaves the chip (SURVEY.md 8(d): 16.5 B)"}


def config4(torch, qh, dev):
    from quisk_amd import synth
    nch = 256
    nblk = int(os.environ.get("QH_C4_NBLK", "4096"))       # DSP blocks per call: 2^22 input samples per channel per step
    n_in = nblk * 1024
    eng = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
    modes = [1, 6, 5]
    for c in range(nch):
        m = modes[c % 3]
        eng.SetRXAShiftRun(c, 1); eng.SetRXAShiftFreq(c, synth.shift_freq(c)); eng.RXANBPSetRun(c, 1)
        eng.SetRXAMode(c, m); eng.SetRXAAGCMode(c, 0); eng.SetRXAAGCFixed(c, 0.0)
        if m == 1: eng.RXASetPassband(c, 300.0, 3000.0)
        elif m == 6: eng.RXASetPassband(c, -4000.0, 4000.0)
        else: eng.RXASetPassband(c, -8000.0, 8000.0)
    # USB channels get the two-tone input of C2, AM channels a carrier with m = 0.5 / 1 kHz, FM channels a
    # carrier with a 1 kHz tone at +-3 kHz deviation, all + noise (synth.make_mode_input_numpy)
    kinds = {1: "usb", 6: "am", 5: "fm"}
    x = synth.make_mode_input_torch([kinds[modes[c % 3]] for c in range(nch)], n_in, dev)
    y = torch.empty((nch, nblk * 256), dtype=torch.complex128, device=dev)
    sync = lambda: torch.cuda.synchronize(dev)
    t = timed(lambda: eng.process_ptr(x.data_ptr(), n_in, y.data_ptr(), nblk * 256, nblk), sync, steps=5, warmup=1)
    # the same call with the launch sequence replayed from hipGraphs (qh_rxa_set_graph_replay): ~25 launches per call
    eng.set_graph_replay(True)
    tg = timed(lambda: eng.process_ptr(x.data_ptr(), n_in, y.data_ptr(), nblk * 256, nblk), sync, steps=8, warmup=4)
    eng.set_graph_replay(False)
    eng.enable_timing(True)
    eng.process_ptr(x.data_ptr(), n_in, y.data_ptr(), nblk * 256, nblk)
    kt = eng.timing_ms()
    eng.enable_timing(False)
    tot = nch * n_in
    return {"front_ms": kt[0], "band_ms": kt[1], "rest_ms": kt[2],
            "config": "4 (one GPU's share): 256 ch x 192 k, mode by c mod 3 = USB / AM / FM, fp64", "samples_per_step": tot,
            "ms": t * 1e3, "Msamp_per_s": tot / t / 1e6, "ms_graph_replay": tg * 1e3, "Msamp_per_s_graph_replay": tot / tg / 1e6,
            "graph_launches": eng.graph_launches(),
            "pll_tiles_rerun": eng.pll_repairs(),
            "note": "AM / FM / notch recurrences time-tiled (qh_tiled.hpp); pll_tiles_rerun = FM tiles the verify pass re-ran sequentially over all steps"}


def config2_agc(torch, qh, dev):
    """BASELINE config 2's chain with WDSP's AGC state machine running (SetRXAAGCMode 3, the mode Quisk's WDSP path sets by default)
    instead of the fixed gain the configuration specifies: the level detector in time tiles (qh_agc_tiled.hpp)."""
    from quisk_amd import synth
    nch, nblk = 256, int(os.environ.get("QH_C2A_NBLK", "4096"))
    n_in = nblk * 1024
    # the same buffer is fed every step: the tones sit on the buffer's frequency grid (moved by < 0.023 Hz), so that the steps are one
    # continuous stream -- a phase jump per call is a click the AGC answers for seconds, which no receiver's input has
    x = synth.make_mode_input_torch(["usb"] * nch, n_in, dev, periodic=True)
    y = torch.empty((nch, nblk * 256), dtype=torch.complex128, device=dev)
    e = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
        e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 3)
    sync = lambda: torch.cuda.synchronize(dev)
    t = timed(lambda: e.process_ptr(x.data_ptr(), n_in, y.data_ptr(), nblk * 256, nblk), sync, steps=4, warmup=2)
    tot = nch * n_in
    return {"config": "2 with the AGC state machine on (SetRXAAGCMode 3): 256 ch x 192 k SSB RXA, fp64, 2^%d samples per channel and step" % (n_in.bit_length() - 1),
            "samples_per_step": tot, "ms": t * 1e3, "Msamp_per_s": tot / t / 1e6, "agc_tiles_rerun": e.agc_repairs(),
            "agc_segments_rerun": e.agc_segments_rerun(),
            "note": "not a BASELINE configuration (config 2 fixes the gain); agc_tiles_rerun = tiles whose boundary state the exact pass corrected"}


def config5(torch, qh, dev):
    n = 1 << 26                 # 61.44 Msps stream, ~1.09 s of signal, fp32
    s = torch.cuda.current_stream(dev).cuda_stream
    tabs = __import__("quisk_amd.rxfilter", fromlist=["x"]).coefficient_tables()
    hb = [qh.FirBank(1, qh.hb45_taps(), 2, dtype=1, stream=s) for _ in range(8)]
    d5 = qh.FirBank(1, tabs["quiskFilt240D5CoefsSharp"], 5, dtype=1, stream=s)
    # WDSP bandpass 300..3000 at 48 k, nc 2048 (fir_bandpass via the library's own design is internal; the bank takes taps)
    m = 0.5 * 2047
    pos = np.arange(2048) - m
    c = np.cos(np.pi / m * np.arange(2048))
    win = 0.21747 + c * (-0.45325 + c * (0.28256 + c * (-0.04672)))
    ft = (3000.0 - 300.0) / (2 * 48000.0)
    bp = np.sin(2 * np.pi * ft * pos) / (np.pi * pos) * win * np.exp(-1j * np.pi * 3300.0 / 48000.0 * pos)
    core = qh.FirBank(1, bp, 1, dtype=1, stream=s)
    x = torch.randn((1, n), dtype=torch.float32, device=dev) + 1j * torch.randn((1, n), dtype=torch.float32, device=dev)
    bufs = [torch.empty((1, n >> (k + 1)), dtype=torch.complex64, device=dev) for k in range(8)]
    y5 = torch.empty((1, (n >> 8) // 5 + 8), dtype=torch.complex64, device=dev)
    yo = torch.empty_like(y5)

    def step():
        cur, cn = x, n
        for k in range(8):
            m_ = hb[k].process_ptr(cur.data_ptr(), cur.shape[1], cn, bufs[k].data_ptr(), bufs[k].shape[1])
            cur, cn = bufs[k], m_
        m_ = d5.process_ptr(cur.data_ptr(), cur.shape[1], cn, y5.data_ptr(), y5.shape[1])
        core.process_ptr(y5.data_ptr(), y5.shape[1], m_, yo.data_ptr(), yo.shape[1])
    casc = qh.HalfBandCascade(1, 8, dtype=1, stream=s)

    def step_fused():
        m_ = casc.process_ptr(x.data_ptr(), n, n, bufs[7].data_ptr(), bufs[7].shape[1])
        m_ = d5.process_ptr(bufs[7].data_ptr(), bufs[7].shape[1], m_, y5.data_ptr(), y5.shape[1])
        core.process_ptr(y5.data_ptr(), y5.shape[1], m_, yo.data_ptr(), yo.shape[1])
    sync = lambda: torch.cuda.synchronize(dev)
    t = timed(step, sync, steps=10, warmup=2)
    t1 = timed(lambda: hb[0].process_ptr(x.data_ptr(), n, n, bufs[0].data_ptr(), bufs[0].shape[1]), sync, steps=10, warmup=2)
    tf = timed(step_fused, sync, steps=10, warmup=2)
    tc = timed(lambda: casc.process_ptr(x.data_ptr(), n, n, bufs[7].data_ptr(), bufs[7].shape[1]), sync, steps=10, warmup=2)
    return {"fused_ms": tf * 1e3, "fused_Msamp_per_s": n / tf / 1e6, "fused_cascade_only_ms": tc * 1e3,
            "fused_algorithmic_GBps": 8.0 * n / tf / 1e9,
            "config": "5 (one GPU's channel): 1 ch x 61.44 Msps fp32, 8 x HB45 + 245-tap /5 + bandpass nc 2048, 2^26 samples per step",
            "samples_per_step": n, "ms": t * 1e3, "Msamp_per_s": n / t / 1e6, "first_stage_ms": t1 * 1e3,
            "algorithmic_GBps": 8.0 * n / t / 1e9,
            "note": "unfused cascade of overlap-save banks; the first half-band alone moves 12 B/sample"}


def quisk_native(torch, qh, dev):
    """Path A: 256 receivers of Quisk's own chain (quisk_process_samples: tune, quisk_process_decimate, cRxFilterOut, the x4
    interpolators), 192 ksps in, 48 ksps out, USB, AM and FM, 2^20 input samples per receiver per step.  The reference's own
    figure for ONE receiver on a CPU core is 9.9 Msamp/s."""
    from quisk_amd import rxfilter
    nch, n, fs = 256, 1 << 20, 192000
    out = []
    only = os.environ.get("QH_QUISK_MODES", "USB,AM,FM").split(",")          # e.g. QH_QUISK_MODES=FM for a kernel trace of one mode
    for name, mode, bw in (("USB", rxfilter.USB, 2700), ("AM", rxfilter.AM, 6000), ("FM", rxfilter.FM, 12000)):
        if name not in only:
            continue
        bank = qh.QuiskRxBank(nch, fs, mode, bw, stream=torch.cuda.current_stream(dev).cuda_stream)
        rate = bank.get_filter_rate()
        fI, fQ = rxfilter.make_filter_coef(rate, None, bw, rxfilter.get_filter_center(name, bw))
        for c in range(nch):
            bank.set_tune(c, 1000 * (c % 40) - 20000)
        bank.set_filters(-1, fI, fQ)
        x = (torch.randn((nch, n), dtype=torch.float64, device=dev) + 1j * torch.randn((nch, n), dtype=torch.float64, device=dev)) * 2.0 ** 22
        m = bank.out_count(n)
        y = torch.empty((nch, m + 64), dtype=torch.complex128, device=dev)
        sync = lambda: torch.cuda.synchronize(dev)
        t = timed(lambda: bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), m + 64), sync, steps=8, warmup=2)
        row = {"mode": name, "ms": t * 1e3, "Msamp_per_s": nch * n / t / 1e6, "filter_rate": rate, "filter_taps": int(fI.size)}
        # process_agc on the output, as quisk_process_samples always runs it (quisk.c:2685-2701); a release gain at which the
        # limiter works (an overload ramp every few FIFO cycles): the state machine is sequential per receiver, one wavefront each
        bank.set_agc(True, 5000.0)
        ta = timed(lambda: bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), m + 64), sync, steps=6, warmup=2)
        row.update({"agc_on_ms": ta * 1e3, "agc_on_Msamp_per_s": nch * n / ta / 1e6})
        out.append(row)
        del bank, x, y
    return {"config": "Quisk-native chain (path A): 256 receivers x 192 ksps -> 48 ksps, 2^20 input samples per receiver per step",
            "samples_per_step": nch * n, "modes": out,
            "note": "the reference's quisk_process_samples handles one receiver per process: 9.9 Msamp/s on a CPU core (SURVEY.md 8 a10)"}


def analyzer(torch, qh, dev):
    """WDSP display engine (wdsp/analyzer.c) for a bank of 64 displays fed 2^20 samples each per step: 16384-point frames with
    50 % overlap (127 frames per display and step), Blackman-Harris window, peak detector to 2048 pixels, recursive averaging."""
    nd, n, size = 64, 1 << 20, 16384
    a = qh.AnalyzerBank(nd, size, stream=torch.cuda.current_stream(dev).cuda_stream)
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


def main():
    import torch
    import quisk_amd as qh
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    which = sys.argv[1:] or ["3", "4", "5"]
    for w in which:
        r = {"3": config3, "4": config4, "5": config5, "2agc": config2_agc, "analyzer": analyzer, "quisk": quisk_native}[w](torch, qh, dev)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
