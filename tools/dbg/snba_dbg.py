import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import quisk_amd as qh
from oracle import pyoracle as oracle
from quisk_amd import synth
import test_gpu_snba_parity as T
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
nblk = 160
x = T.crackle(0, nblk * 1024, mode, rate=rate)
e = qh.RxaEngine(1)
o = oracle.WdspChannel(1024, 256, 192000, 48000, 48000)
for t, a in ((e, (0,)), (o, ())):
    T.setup(t, a, 0, mode); t.SetRXASNBARun(*a, 1)
opt = sys.argv[3] if len(sys.argv) > 3 else ""
if "n" in opt:
    for t, a in ((e, (0,)), (o, ())):
        t.RXANBPAddNotch(*a, 0, 1210.0, 120.0, 1); t.RXANBPSetNotchesRun(*a, 1)
if "g" in opt: e.set_graph_replay(True)
if "b" in opt:
    y = np.concatenate([e.process_host(x[None, b * 1024:(b + 1) * 1024])[0] for b in range(nblk)])
else:
    y = e.process_host(x[None])[0]
r = o.xrxa(x)
d = np.abs(y - r).reshape(-1, 256).max(axis=1)
sc = np.sqrt(np.mean(np.abs(r) ** 2))
print("rel rms", np.sqrt(np.mean(np.abs(y - r) ** 2)) / sc)
print("blocks with err > 1e-9:", [(int(i), float("%.2g" % (d[i] / sc))) for i in np.nonzero(d / sc > 1e-9)[0]][:60])
