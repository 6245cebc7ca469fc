#!/usr/bin/env python3
"""Wall time per fexchange0 call of the WDSP drop-in (one channel, host buffers, Quisk's call pattern)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quisk_amd as qh          # noqa: E402

D = C.c_double
lib = qh.load()
for in_size, in_rate in ((256, 48000), (1024, 192000), (4096, 192000)):
    ch = 1
    lib.OpenChannel(ch, in_size, 256, in_rate, 48000, 48000, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    lib.SetRXAShiftRun(ch, 1 if in_rate > 48000 else 0); lib.SetRXAShiftFreq(ch, D(10000.0)); lib.RXANBPSetRun(ch, 1)
    lib.SetRXAMode(ch, 1); lib.RXASetPassband(ch, D(300.0), D(3000.0)); lib.SetRXAAGCMode(ch, 0); lib.SetRXAAGCFixed(ch, D(0.0))
    out_size = in_size * 48000 // in_rate
    x = (np.random.default_rng(0).standard_normal(in_size) + 0j).astype(np.complex128)
    y = np.zeros(out_size, dtype=np.complex128)
    err = C.c_int(0)
    for _ in range(50):
        lib.fexchange0(ch, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.byref(err))
    t0 = time.perf_counter()
    n = 500
    for _ in range(n):
        lib.fexchange0(ch, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.byref(err))
    dt = (time.perf_counter() - t0) / n
    print("in_size %5d @ %6d Hz: %7.1f us per fexchange0 call = %5.2f %% of real time" % (in_size, in_rate, dt * 1e6, 100 * dt / (in_size / in_rate)))
    lib.qh_wdsp_graph_launches.restype = C.c_longlong
    print("   blocks replayed from hipGraphs so far:", lib.qh_wdsp_graph_launches())
    lib.CloseChannel(ch)

# quisk_process_samples (the Quisk-native drop-in, include/quiskhip.h group 9): one receiver, host buffer in place, USB, process_agc on
from quisk_amd import quiskapi as QS, rxfilter      # noqa: E402
for fs, blk in ((48000, 1024), (192000, 4096)):
    QS.open(fs, playback_rate=48000)
    QS.set_rx_mode(rxfilter.USB); QS.set_tune(5000)
    fI, fQ = rxfilter.make_filter_coef(QS.get_filter_rate(), None, 2700, rxfilter.get_filter_center("USB", 2700))
    QS.set_filters(fI, fQ, 2700); QS.set_agc(5000.0)
    rng = np.random.default_rng(1)
    x = ((rng.standard_normal(blk) + 1j * rng.standard_normal(blk)) * 2.0 ** 22).astype(np.complex128)
    buf = np.zeros(blk + 4096, dtype=np.complex128)
    for _ in range(50):
        buf[:blk] = x; QS.process_samples(buf, blk)
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        buf[:blk] = x; QS.process_samples(buf, blk)
    dt = (time.perf_counter() - t0) / n
    print("quisk_process_samples %5d @ %6d Hz: %7.1f us per call = %5.2f %% of real time" % (blk, fs, dt * 1e6, 100 * dt / (blk / fs)))
    QS.close()
