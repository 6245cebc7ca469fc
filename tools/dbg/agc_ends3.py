"""first long call of a fresh engine with the AGC tiles in check-only mode: which slots / tiles / fields miss"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda:0")
nch, nblk = 256, 4096
x = synth.make_mode_input_torch(["usb"] * nch, nblk * 1024, dev)
y = torch.empty((nch, nblk * 256), dtype=torch.complex128, device=dev)
e = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
for c in range(nch):
    e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
    e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 3)
e.debug_agc(3)
buf = (C.c_double * 400000)()
e._L.qh_rxa_debug_agc_ends.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
e.process_ptr(x.data_ptr(), nblk * 1024, y.data_ptr(), nblk * 256, nblk)
print("mismatches counted", e.agc_repairs())
tot = 0
for slot in range(nch):
    n = e._L.qh_rxa_debug_agc_ends(e._h, slot, buf, 400000)
    a = np.frombuffer(buf, dtype=np.float64)[:n].reshape(2, -1, 8).copy()
    nt = 255
    w, p = a[0, 1:nt + 1, 0:5], a[1, 0:nt, 0:5]
    dv = np.abs(w[:, 0] - p[:, 0]) / np.abs(p[:, 0]); dsv = np.abs(w[:, 1] - p[:, 1]) / np.maximum(np.abs(p[:, 1]), 1e-300)
    disc = (w[:, 2:] != p[:, 2:]).any(axis=1)
    bad = np.nonzero(~((dv <= 1e-11) & (np.abs(w[:, 1] - p[:, 1]) <= 1e-9 * np.abs(p[:, 1])) & ~disc))[0]
    tot += len(bad)
    if len(bad) and tot < 600: print("slot", slot, "bad", len(bad), bad[:6], "bounds", w[bad[0]], "ends", p[bad[0]])
print("bad in the arrays", tot)
