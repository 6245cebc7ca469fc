"""the shift oscillator's law changed in mid-stream (SetRXAShiftFreq / SetRXAShiftRun) at several in_rate / dsp_rate ratios"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
import pyoracle as oracle
from quisk_amd import synth
for in_rate, dsp_size in ((192000, 256), (144000, 256), (144000, 64), (240000, 256), (96000, 128), (48000, 256), (288000, 64)):
    d_in = dsp_size * in_rate // 48000
    nblk = 60
    x = synth.make_input_numpy(1, nblk * d_in, fs=float(in_rate))
    e = qh.RxaEngine(1, dsp_size=dsp_size, in_rate=in_rate, dsp_rate=48000, out_rate=48000)
    o = oracle.WdspChannel(d_in, dsp_size, in_rate, 48000, 48000)
    for t, lead in ((e, (0,)), (o, ())):
        t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, 10000.0); t.RXANBPSetRun(*lead, 1); t.SetRXAMode(*lead, 1); t.RXASetPassband(*lead, 300.0, 3000.0); t.SetRXAAGCMode(*lead, 0)
    plan = [(10, None), (10, ("SetRXAShiftFreq", 7000.0)), (10, ("SetRXAShiftRun", 0)), (10, ("SetRXAShiftRun", 1)), (20, ("SetRXAShiftFreq", -3000.0))]
    pos, errs = 0, []
    for nb, st in plan:
        if st:
            getattr(e, st[0])(0, st[1]); getattr(o, st[0])(st[1])
        seg = np.ascontiguousarray(x[:, pos * d_in:(pos + nb) * d_in]); pos += nb
        y, r = e.process_host(seg)[0], o.xrxa(seg[0])
        errs.append(np.abs(y - r).max() / max(np.abs(r).max(), 1e-30))
    print("in_rate %6d dsp_size %4d: %s" % (in_rate, dsp_size, " ".join("%.1e" % v for v in errs)), flush=True)
    e.close()
