"""one block-at-a-time walk of tests/test_gpu_rxa_fuzz.py (graph replay on) again, block by block: where a channel parts from the oracle.
replay_probe.py <seed> <channel> [replay 0|1] [meters 0|1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
import pyoracle as oracle
import test_gpu_rxa_fuzz as F
seed, chan = int(sys.argv[1]), int(sys.argv[2])
replay = int(sys.argv[3]) if len(sys.argv) > 3 else 1
meters = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(seed)
NCH, nseg = F.NCH, 45
seglen = [int(rng.integers(1, 6)) for _ in range(nseg)]
nblk = sum(seglen)
x = synth.make_input_numpy(NCH, nblk * 1024)
x[1] = synth.make_mode_input_numpy("am", 1, nblk * 1024)
e = qh.RxaEngine(NCH); e.load_emnr_tables()
if replay: e.set_graph_replay(True)
if meters: e.enable_meters(True)
dev = torch.device("cuda:0")
d_in = torch.zeros((NCH, 1024), dtype=torch.complex128, device=dev)
d_out = torch.zeros((NCH, 256), dtype=torch.complex128, device=dev)
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
            d = F._apply(rng, [(e, (c,)), (os_[c], ())], False)
            print("   seg %d channel %d: %r" % (s, c, d))
    seg = x[:, pos * 1024:(pos + n) * 1024]
    ref = os_[chan].xrxa(seg[chan])
    for c in range(NCH):
        if c != chan: os_[c].xrxa(seg[c])
    for b in range(n):
        d_in.copy_(torch.from_numpy(seg[:, b * 1024:(b + 1) * 1024]))
        torch.cuda.synchronize()
        e.process_ptr(d_in.data_ptr(), 1024, d_out.data_ptr(), 256, 1)
        e.synchronize()
        y = d_out.cpu().numpy()[chan]
        r = ref[b * 256:(b + 1) * 256]
        print("seg %2d block %3d: max err %.3e, |ref| max %.3e, |y| max %.3e, graph launches %d" % (s, pos + b, np.abs(y - r).max(), np.abs(r).max(), np.abs(y).max(), e.graph_launches()))
    pos += n
