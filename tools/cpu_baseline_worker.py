#!/usr/bin/env python3
"""One worker of bench.py's CPU baseline: ONE RXA channel of the CPU oracle (oracle/wdsp_oracle.c, the restatement of the
reference's WDSP path) on ONE core.  Pins itself to the core it is given, makes its own input and output buffers there (first
touch on that core's memory), waits for the common start time, runs, prints one JSON line.  numpy + ctypes only (no torch).

    cpu_baseline_worker.py CORE CHANNEL LOG2_SAMPLES START_UNIX_TIME
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    core, chan, log2n, t_start = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    if core >= 0 and hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, {core})
    import numpy as np
    from oracle import pyoracle as po
    n = 1 << log2n
    fs = 192000.0
    shift = 10000.0 + 37.0 * chan                               # quisk_amd/synth.py: the bench's signal model, made here on this core
    rng = np.random.default_rng(1000 + chan)
    x = np.empty(n, dtype=np.complex128)
    step = 1 << 18                                              # in pieces: the worker's peak memory stays near its two buffers
    for k in range(0, n, step):
        t = np.arange(k, min(n, k + step), dtype=np.float64)
        g = rng.standard_normal((t.size, 2))
        x[k:k + t.size] = (0.1 * np.exp(2j * np.pi * (((-1000.0 - shift) / fs) * t % 1.0))
                           + 0.05 * np.exp(2j * np.pi * (((30000.0 - shift) / fs) * t % 1.0)) + 0.01 * (g[:, 0] + 1j * g[:, 1]))
    del t, g
    ch = po.WdspChannel(1024, 256, 192000, 48000, 48000)
    ch.SetRXAShiftRun(1)
    ch.SetRXAShiftFreq(shift)
    ch.RXANBPSetRun(1)
    ch.SetRXAMode(1)
    ch.RXASetPassband(300.0, 3000.0)
    ch.SetRXAAGCMode(0)
    ch.SetRXAAGCFixed(0.0)
    ch.xrxa(x[:1024 * 64])                                      # warm: code, twiddles, the channel's filter masks
    while time.time() < t_start:
        time.sleep(0.001)
    t0 = time.time()
    y = ch.xrxa(x)
    t1 = time.time()
    print(json.dumps({"core": core, "channel": chan, "samples": n, "t0": t0, "t1": t1, "seconds": t1 - t0,
                      "check": float(np.abs(y[-4096:]).mean())}), flush=True)


if __name__ == "__main__":
    main()
