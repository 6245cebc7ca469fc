"""one wide walk of tests/test_gpu_rxa_fuzz.py with a SECOND restatement whose input carries 1e-13 of noise per sample: how far the
restatement is from itself next to how far the engine is from it (an LMS filter or a cepstrum amplifies rounding; a bug does not
care).  <seed> <channel>; env SKIP=setter,... leaves those setters out on every side (the draws stay the same)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
from quisk_amd import synth
import pyoracle as oracle
import test_gpu_rxa_fuzz as F
seed, chan = int(sys.argv[1]), int(sys.argv[2])
SKIP = set(filter(None, os.environ.get("SKIP", "").split(",")))
KEEP = set(filter(None, os.environ.get("KEEP", "").split(",")))          # (if given: of the walk's Set* / RXA* calls only these get through)
walking = False


class Skipping:
    def __init__(self, t): self._t = t
    def __getattr__(self, name):
        drop = name in SKIP or (walking and KEEP and name not in KEEP and (name.startswith("Set") or name.startswith("RXA")))
        return (lambda *a: None) if drop else getattr(self._t, name)


pert = np.random.default_rng(99)
rng = np.random.default_rng(seed)
NCH, nseg = F.NCH, 30
seglen = [int(rng.integers(1, 6)) for _ in range(nseg)]
for k in rng.choice(nseg, 5, replace=False):
    seglen[int(k)] = int(rng.integers(70, 91))
nblk = sum(seglen)
x = synth.make_input_numpy(NCH, nblk * 1024)
x[1] = synth.make_mode_input_numpy("am", 1, nblk * 1024)
e0 = qh.RxaEngine(NCH); e0.load_emnr_tables(); e = Skipping(e0)
os_ = [Skipping(oracle.WdspChannel(1024, 256, 192000, 48000, 48000)) for _ in range(NCH)]
twin = Skipping(oracle.WdspChannel(1024, 256, 192000, 48000, 48000))
for c in range(NCH):
    for t, lead in ((e, (c,)), (os_[c], ())) + (((twin, ()),) if c == chan else ()):
        t.SetRXAShiftRun(*lead, 1); t.SetRXAShiftFreq(*lead, synth.shift_freq(c)); t.RXANBPSetRun(*lead, 1)
        t.SetRXAMode(*lead, (1, 6, 0, 1)[c]); t.RXASetPassband(*lead, *((300.0, 3000.0), (-4000.0, 4000.0), (-3000.0, -300.0), (300.0, 3000.0))[c])
        t.SetRXAAGCMode(*lead, (0, 3, 4, 2)[c])
pos = 0
walking = True
ys, rs, ts = [], [], []
for s, n in enumerate(seglen):
    if s:
        for _ in range(int(rng.integers(1, 3))):
            c = int(rng.integers(0, NCH))
            d = F._apply(rng, [(e, (c,)), (os_[c], ())] + ([(twin, ())] if c == chan else []), True)
            if c == chan: print("   seg %d: %r" % (s, d))
    seg = x[:, pos * 1024:(pos + n) * 1024]
    y = e.process_host(seg)
    for c in range(NCH):
        r = os_[c].xrxa(seg[c])
        if c == chan: ref = r
    tw = twin.xrxa(seg[chan] * (1.0 + 1e-13 * pert.standard_normal(seg[chan].size)))
    m = max(np.abs(ref).max(), 1e-300)
    print("seg %2d  %2d blocks  engine %.2e   twin %.2e   |ref| max %.3e" % (s, n, np.abs(y[chan] - ref).max() / m, np.abs(tw - ref).max() / m, m))
    ys.append(y[chan].copy()); rs.append(ref); ts.append(tw)
    pos += n
y, r, t = np.concatenate(ys), np.concatenate(rs), np.concatenate(ts)
rr = lambda a, b: float(np.sqrt((np.abs(a - b) ** 2).sum() / max((np.abs(b) ** 2).sum(), 1e-300)))
print("whole walk: engine vs restatement %.3e   restatement vs its twin %.3e" % (rr(y, r), rr(t, r)))
