"""OpenChannel geometries through fexchange0 against the oracle, no setters in mid-stream"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
import pyoracle as oracle
from quisk_amd import synth
from conftest import rel_rms
lib = qh.load(); D = C.c_double
for geo in ((192, 64, 144000, 48000, 48000), (192, 64, 144000, 48000, 48000, "noshift"), (384, 128, 144000, 48000, 48000), (768, 256, 144000, 48000, 48000), (320, 64, 240000, 48000, 48000), (96, 64, 144000, 48000, 48000), (192, 64, 192000, 48000, 48000)):
    in_size, dsp_size, in_rate, dsp_rate, out_rate = geo[:5]
    shift = len(geo) == 5
    ch = 20
    lib.OpenChannel(ch, in_size, dsp_size, in_rate, dsp_rate, out_rate, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
    o = oracle.WdspChannel(in_size, dsp_size, in_rate, dsp_rate, out_rate)
    lib.SetRXAShiftRun(ch, 1 if shift else 0); o.SetRXAShiftRun(1 if shift else 0)
    lib.SetRXAShiftFreq(ch, D(10000.0)); o.SetRXAShiftFreq(10000.0)
    lib.RXANBPSetRun(ch, 1); o.RXANBPSetRun(1); lib.SetRXAMode(ch, 1); o.SetRXAMode(1)
    lib.RXASetPassband(ch, D(300.0), D(3000.0)); o.RXASetPassband(300.0, 3000.0); lib.SetRXAAGCMode(ch, 0); o.SetRXAAGCMode(0)
    nblk = 300
    x = synth.make_input_numpy(1, nblk * in_size)[0]
    y = np.zeros(nblk * o.out_size, dtype=np.complex128); err = C.c_int(0)
    for k in range(nblk):
        blk = np.ascontiguousarray(x[k * in_size:(k + 1) * in_size])
        lib.fexchange0(ch, blk.ctypes.data_as(C.c_void_p), y[k * o.out_size:].ctypes.data_as(C.c_void_p), C.byref(err))
    r, _ = o.fexchange0(x)
    lib.CloseChannel(ch)
    nz = lambda v: int(np.flatnonzero(np.abs(v) > 0)[0]) if np.any(np.abs(v) > 0) else -1
    print(geo, "rel rms %.2e  first non-zero output %d / %d  max %.3e / %.3e" % (rel_rms(y, r), nz(y), nz(r), np.abs(y).max(), np.abs(r).max()), flush=True)
