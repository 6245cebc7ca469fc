"""time-tiled wcpAGC on the benchmark's input: repairs per call and the size of the misses"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda:0")
nch, nblk = int(os.environ.get("NCH", "8")), int(os.environ.get("NBLK", "1024"))
x = synth.make_mode_input_torch(["usb"] * nch, nblk * 1024, dev)
y = torch.empty((nch, nblk * 256), dtype=torch.complex128, device=dev)
e = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
for c in range(nch):
    e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
    e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 3)
if os.environ.get("CHECK_ONLY"): e.debug_agc(3)
buf = (C.c_double * 400000)()
e._L.qh_rxa_debug_agc_ends.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
prev = 0
for call in range(4):
    e.process_ptr(x.data_ptr(), nblk * 1024, y.data_ptr(), nblk * 256, nblk)
    r = e.agc_repairs()
    n = e._L.qh_rxa_debug_agc_ends(e._h, int(os.environ.get("SLOT", "0")), buf, 400000)
    a = np.frombuffer(buf, dtype=np.float64)[:n].reshape(2, -1, 8).copy()      # boundary states, end states
    print("call", call, "repairs", r - prev); prev = r
    nt = 0
    for t in range(1, a.shape[1]):
        if a[0, t, 0] == 0: break
        nt = t
    w, p = a[0, 1:nt + 1, 0:5], a[1, 0:nt, 0:5]
    dv = np.abs(w[:, 0] - p[:, 0]) / np.abs(p[:, 0]); dsv = np.abs(w[:, 1] - p[:, 1]) / np.maximum(np.abs(p[:, 1]), 1e-300)
    disc = (w[:, 2:] != p[:, 2:]).any(axis=1)
    bad = np.nonzero((dv > 1e-11) | (dsv > 1e-9) | disc)[0]
    if len(bad): print("   first bad tiles", bad[:8], "bounds", w[bad[0]], "ends", p[bad[0]])
    print("   tiles", nt, "dv>1e-9:", int((dv > 1e-9).sum()), "dsv>1e-6:", int((dsv > 1e-6).sum()), "discrete:", int(disc.sum()),
          "dv median %.1e max %.1e" % (np.median(dv), dv.max()))
