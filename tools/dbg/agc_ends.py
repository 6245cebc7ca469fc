"""which part of the lanes' state fails the time-tiled wcpAGC's check (steady two-tone input)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import quisk_amd as qh
from quisk_amd import synth
nch, nblk = 2, 300
x = np.stack([synth.make_mode_input_numpy("usb", c, nblk * 1024) for c in range(nch)])
e = qh.RxaEngine(nch)
for c in range(nch):
    e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
    e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 3)
e.process_host(np.ascontiguousarray(x[:, :40 * 1024]))
e.process_host(np.ascontiguousarray(x[:, 40 * 1024:]))
print("repairs", e.agc_repairs())
buf = (C.c_double * 200000)()
e._L.qh_rxa_debug_agc_ends.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
n = e._L.qh_rxa_debug_agc_ends(e._h, 0, buf, 200000)
a = np.frombuffer(buf, dtype=np.float64)[:n].reshape(-1, 12)
L = int(os.environ.get("QH_AGC_TILE", "512"))
nt = (260 * 256 + L - 1) // L
for t in range(1, min(nt, 130)):
    w, p = a[t, 0:5], a[t - 1, 5:10]
    bad = abs(w[0] - p[0]) > 1e-12 * abs(p[0]) or abs(w[1] - p[1]) > 1e-12 * abs(p[1]) or any(w[2:] != p[2:])
    if t % 8 == 0 or bad:
        print(t, "BAD" if bad else "ok ", "dv %.2e dsv %.2e" % ((w[0] - p[0]) / p[0], (w[1] - p[1]) / (p[1] if p[1] else 1)), w[2:], p[2:])
