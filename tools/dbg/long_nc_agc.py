"""A long call behind short ones with nc = 16384 on one channel of four, AGC state machine on (the shape of walk rxa_long 900190)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
import pyoracle as oracle
NCH = 4
def run(nc1, agc1, mode1, bp_direct, seglen):
    nblk = sum(seglen)
    x = synth.make_input_numpy(NCH, nblk * 1024)
    x[1] = synth.make_mode_input_numpy("am", 1, nblk * 1024)
    e = qh.RxaEngine(NCH)
    os_ = [oracle.WdspChannel(1024, 256, 192000, 48000, 48000) for _ in range(NCH)]
    for c in range(NCH):
        for t, lead in ((e, (c,)), (os_[c], ())):
            t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, synth.shift_freq(c)); t.RXANBPSetRun(*lead, 1)
            t.SetRXAMode(*lead, (1, mode1, 0, 1)[c]); t.RXASetPassband(*lead, *((300.0, 3000.0), (-4000.0, 4000.0), (-3000.0, -300.0), (300.0, 3000.0))[c])
            t.SetRXAAGCMode(*lead, (0, agc1, 4, 2)[c])
    for t, lead in ((e, (1,)), (os_[1], ())):
        t.RXASetNC(*lead, nc1)
        if bp_direct: t.SetRXABandpassRun(*lead, 1)
    pos, worst = 0, []
    for n in seglen:
        seg = x[:, pos * 1024:(pos + n) * 1024]
        y = e.process_host(np.ascontiguousarray(seg))
        errs = []
        for c in range(NCH):
            r = os_[c].xrxa(seg[c])
            errs.append(float(np.abs(y[c] - r).max() / max(np.abs(r).max(), 1e-30)))
        worst.append(errs)
        pos += n
    e.close()
    return worst
for cfg in ((16384, 3, 6, 0), (16384, 1, 4, 1), (16384, 1, 4, 0), (2048, 1, 4, 1), (16384, 0, 4, 1), (8192, 1, 4, 1), (4096, 1, 4, 1)):
    w = run(*cfg, seglen=[3, 2, 80, 2, 75, 3])
    print("nc %5d agc %d mode %d bp1 direct %d: per call, channel 1: %s   others worst %.1e" % (*cfg, " ".join("%.1e" % a[1] for a in w), max(max(a[0], a[2], a[3]) for a in w)), flush=True)
