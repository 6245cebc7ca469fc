"""the RXA chain at in_rate / dsp_rate ratios other than powers of two (and dsp_rate above in_rate), against the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
import pyoracle as oracle
from quisk_amd import synth
from conftest import rel_rms
for in_rate, dsp_size in ((144000, 256), (144000, 64), (240000, 256), (288000, 256), (336000, 256), (24000, 256), (16000, 256), (96000, 64), (192000, 64)):
    try:
        d_in = dsp_size * in_rate // 48000
        nblk = 40
        x = synth.make_input_numpy(2, nblk * d_in, fs=float(in_rate))
        e = qh.RxaEngine(2, dsp_size=dsp_size, in_rate=in_rate, dsp_rate=48000, out_rate=48000)
        errs = []
        for c in range(2):
            e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1); e.RXASetPassband(c, 300.0, 3000.0)
            e.SetRXAAGCMode(c, 0)
        y = np.concatenate([e.process_host(np.ascontiguousarray(x[:, a * d_in:b * d_in])) for a, b in ((0, 7), (7, 8), (8, 40))], axis=1)
        for c in range(2):
            o = oracle.WdspChannel(d_in, dsp_size, in_rate, 48000, 48000)
            o.SetRXAShiftRun(1); o.SetRXAShiftFreq(synth.shift_freq(c)); o.RXANBPSetRun(1); o.SetRXAMode(1); o.RXASetPassband(300.0, 3000.0); o.SetRXAAGCMode(0)
            errs.append(rel_rms(y[c], o.xrxa(x[c])))
        print("in_rate %6d dsp_size %4d: %s" % (in_rate, dsp_size, " ".join("%.1e" % v for v in errs)), flush=True)
        e.close()
    except Exception as ex:
        print("in_rate %6d dsp_size %4d: %s" % (in_rate, dsp_size, str(ex)[:150]), flush=True)
