"""one walk of tests/test_gpu_wdsp_names_fuzz.py again: errors stretch by stretch"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
import pyoracle as oracle
from quisk_amd import synth
import test_gpu_wdsp_names_fuzz as T
from test_gpu_rxa_fuzz import _apply
seed = int(sys.argv[1])
lib = qh.load()
rng = np.random.default_rng(31000 + seed)
in_size, dsp_size, in_rate, dsp_rate, out_rate = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else T.GEOMETRY[(seed - 1) % len(T.GEOMETRY)]
ch = 16 + seed % 8
D = C.c_double
lib.OpenChannel(ch, in_size, dsp_size, in_rate, dsp_rate, out_rate, 0, 1, D(0.010), D(0.025), D(0.0), D(0.010), 1)
o = T._NoEmnr(oracle.WdspChannel(in_size, dsp_size, in_rate, dsp_rate, out_rate), dsp_size)
names = T._Names(lib, ch, dsp_size)
for t in (names, o):
    t.SetRXAShiftRun(1); t.SetRXAShiftFreq(float(synth.shift_freq(seed % 4))); t.RXANBPSetRun(1); t.SetRXAMode(1)
    t.RXASetPassband(300.0, 3000.0); t.SetRXAAGCMode(0)
out_size = o.out_size
nblk = 60 * max(1, 1024 // in_size)
x = synth.make_input_numpy(4, nblk * in_size * 192000 // in_rate)[seed % 4][::192000 // in_rate][:nblk * in_size].copy()
err = C.c_int(0)
b = 0
while b < nblk:
    if b:
        for _ in range(int(rng.integers(0, 2))):
            print("   block %d: %r" % (b, _apply(rng, [(names, ()), (o, ())])))
    n = min(nblk - b, int(rng.integers(1, 5)) * max(1, 1024 // in_size))
    seg = np.ascontiguousarray(x[b * in_size:(b + n) * in_size])
    y = np.zeros(n * out_size, dtype=np.complex128)
    for k in range(n):
        blk = np.ascontiguousarray(seg[k * in_size:(k + 1) * in_size])
        lib.fexchange0(ch, blk.ctypes.data_as(C.c_void_p), y[k * out_size:].ctypes.data_as(C.c_void_p), C.byref(err))
    r, _ = o.fexchange0(seg)
    print("blocks %4d..%4d: max err %.2e of %.3e" % (b, b + n, np.abs(y - r).max(), np.abs(r).max()))
    b += n
lib.CloseChannel(ch)
