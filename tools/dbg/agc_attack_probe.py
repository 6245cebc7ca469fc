"""SetRXAAGCAttack in mid-stream behind a stage that has just been switched on (found by tools/dbg/fuzz_sweep.py, wide seed 1252): one LSB
channel, AGC mode <mode>; block 200: RXASetNC <nc>; block 201: SetRXAEMNRRun <emnr>; block 204: SetRXAAGCAttack 4; engine against the
restatement block by block from there.  <mode> <nc> <emnr> [gap blocks between EMNR and the attack]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
from quisk_amd import synth
import pyoracle as oracle
mode, nc, emnr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gap = int(sys.argv[4]) if len(sys.argv) > 4 else 3
nblk = 201 + gap + 40
x = synth.make_input_numpy(4, nblk * 1024)[2:3].copy()
e = qh.RxaEngine(1); e.load_emnr_tables()
o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
for t, lead in ((e, (0,)), (o, ())):
    t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, synth.shift_freq(2)); t.RXANBPSetRun(*lead, 1)
    t.SetRXAMode(*lead, 0); t.RXASetPassband(*lead, -3000.0, -300.0); t.SetRXAAGCMode(*lead, mode)
def both(name, *a):
    getattr(e, name)(0, *a); getattr(o, name)(*a)
def run(b0, b1):
    seg = x[:, b0 * 1024:b1 * 1024]
    y = e.process_host(seg)[0]
    r = o.xrxa(seg[0])
    return y, r
y, r = run(0, 200); print("blocks 0-199: %.2e" % (np.abs(y - r).max() / np.abs(r).max()))
if nc: both("RXASetNC", nc)
y, r = run(200, 201)
if emnr: both("SetRXAEMNRRun", 1)
for b in range(201, 201 + gap):
    y, r = run(b, b + 1)
    print("block %d (before the attack): err %.2e of %.3e; exact zeros ref %d engine %d; max |engine| where ref == 0: %.3e" % (
        b, np.abs(y - r).max(), np.abs(r).max(), int((r == 0).sum()), int((y == 0).sum()), np.abs(y[r == 0]).max() if (r == 0).any() else 0.0))
both("SetRXAAGCAttack", 4)
for b in range(201 + gap, nblk):
    y, r = run(b, b + 1)
    d = np.abs(y - r)
    print("block %d: max err %.3e at sample %d, |ref| max %.3e, ref[0:3] %s" % (b, d.max(), int(d.argmax()), np.abs(r).max(), np.abs(r[:3])))
    if b == 201 + gap:
        bad = np.nonzero(d > 1e-9 * np.abs(r).max())[0]
        print("   first sample off: %s; first nonzero ref sample %s" % (bad[:1], np.nonzero(np.abs(r) > 0)[0][:1]))
        for i in list(bad[:6]) + list(bad[-3:]):
            print("   sample %d: ref %r engine %r ratio-1 %.3e" % (i, r[i], y[i], abs(y[i] / r[i]) - 1))
