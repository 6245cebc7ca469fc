"""what reaches xwcpagc when EMNR (and with it bp1) is switched on in mid-stream: exact zeros in the restatement -- and in the engine?
AGC mode 0 (a fixed gain), so the output shows the AGC's input.  <nc>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
from quisk_amd import synth
import pyoracle as oracle
nc = int(sys.argv[1])
x = synth.make_input_numpy(4, 215 * 1024)[2:3].copy()
e = qh.RxaEngine(1); e.load_emnr_tables()
o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
for t, lead in ((e, (0,)), (o, ())):
    t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, synth.shift_freq(2)); t.RXANBPSetRun(*lead, 1)
    t.SetRXAMode(*lead, 0); t.RXASetPassband(*lead, -3000.0, -300.0); t.SetRXAAGCMode(*lead, 0)
def both(name, *a):
    getattr(e, name)(0, *a); getattr(o, name)(*a)
def run(b0, b1):
    seg = x[:, b0 * 1024:b1 * 1024]
    return e.process_host(seg)[0], o.xrxa(seg[0])
run(0, 200)
if nc: both("RXASetNC", nc)
run(200, 201)
both("SetRXAEMNRRun", 1)
for b in range(201, 212):
    y, r = run(b, b + 1)
    print("block %d: exact zeros ref %d engine %d; smallest nonzero |engine| %.3e; max |engine| where ref == 0: %.3e; err %.2e" % (
        b, int((r == 0).sum()), int((y == 0).sum()), np.abs(y[y != 0]).min() if (y != 0).any() else 0.0,
        np.abs(y[r == 0]).max() if (r == 0).any() else 0.0, np.abs(y - r).max()))
