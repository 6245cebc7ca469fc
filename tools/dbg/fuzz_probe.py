"""one wide walk of tests/test_gpu_rxa_fuzz.py again, segment by segment: where a channel parts from the oracle, with the AGC's forms"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
import pyoracle as oracle
import test_gpu_rxa_fuzz as F
seed, chan = int(sys.argv[1]), int(sys.argv[2])
form = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
NCH, nseg = F.NCH, 30
seglen = [int(rng.integers(1, 6)) for _ in range(nseg)]
for k in rng.choice(nseg, 5, replace=False):
    seglen[int(k)] = int(rng.integers(70, 91))
nblk = sum(seglen)
x = synth.make_input_numpy(NCH, nblk * 1024)
x[1] = synth.make_mode_input_numpy("am", 1, nblk * 1024)
e = qh.RxaEngine(NCH); e.load_emnr_tables(); e.debug_agc(form)
os_ = [oracle.WdspChannel(1024, 256, 192000, 48000, 48000) for _ in range(NCH)]
for c in range(NCH):
    for t, lead in ((e, (c,)), (os_[c], ())):
        t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, synth.shift_freq(c)); t.RXANBPSetRun(*lead, 1)
        t.SetRXAMode(*lead, (1, 6, 0, 1)[c]); t.RXASetPassband(*lead, *((300.0, 3000.0), (-4000.0, 4000.0), (-3000.0, -300.0), (300.0, 3000.0))[c])
        t.SetRXAAGCMode(*lead, (0, 3, 4, 2)[c])
pos = 0
for s, n in enumerate(seglen):
    if s:
        for _ in range(int(rng.integers(1, 3))):
            c = int(rng.integers(0, NCH))
            d = F._apply(rng, [(e, (c,)), (os_[c], ())], True)
            if c == chan: print("   seg %d: %r" % (s, d))
    seg = x[:, pos * 1024:(pos + n) * 1024]
    y = e.process_host(seg)
    ref = os_[chan].xrxa(seg[chan])
    err = np.abs(y[chan] - ref).max() / max(np.abs(ref).max(), 1e-300)
    print("seg %2d  %2d blocks  max err / max |ref| = %.2e   |ref| max %.3e   tiled channels %d" % (s, n, err, np.abs(ref).max(), e.agc_tiled_channels()))
    pos += n
