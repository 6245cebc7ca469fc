"""does a long (time-tiled, two-stream) call write below the offset it was given?  the acquisition rows of the config 4 share test,
snapshotted before the long call and compared after it; the same for the three uneven calls"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda:0")
nch, nacq, nblk = 256, 160, 256
modes, kinds = [1, 6, 5], {1: "usb", 6: "am", 5: "fm"}
n_in = (nacq + nblk) * 1024
x = torch.from_numpy(np.stack([synth.make_mode_input_numpy(kinds[modes[c % 3]], c, n_in) for c in range(nch)])).to(dev)
def make():
    e = qh.RxaEngine(nch)
    for c in range(nch):
        m = modes[c % 3]
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1)
        e.SetRXAMode(c, m); e.SetRXAAGCMode(c, 0); e.SetRXAAGCFixed(c, 0.0)
        e.RXASetPassband(c, *((300.0, 3000.0) if m == 1 else (-4000.0, 4000.0) if m == 6 else (-8000.0, 8000.0)))
    return e
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ea, eb = make(), make()
    W = (nacq + nblk) * 256
    ya = torch.full((nch, W + 4096), complex(7.0, 7.0), dtype=torch.complex128, device=dev)       # a guard band behind the rows too
    yb = torch.full((nch, W + 4096), complex(7.0, 7.0), dtype=torch.complex128, device=dev)
    torch.cuda.synchronize()
    for e, y in ((ea, ya), (eb, yb)):
        for k in range(10):
            e.process_ptr(x.data_ptr() + 16 * k * 16 * 1024, n_in, y.data_ptr() + 16 * k * 16 * 256, ya.shape[1], 16)
    ea.synchronize(); eb.synchronize()
    sa, sb = ya[:, :nacq * 256].clone(), yb[:, :nacq * 256].clone()
    print("iteration %d: acquisition equal %s" % (it, torch.equal(sa, sb)), end="; ")
    ea.process_ptr(x.data_ptr() + 16 * nacq * 1024, n_in, ya.data_ptr() + 16 * nacq * 256, ya.shape[1], nblk)
    pos = nacq
    for nb in (100, 7, 149):
        eb.process_ptr(x.data_ptr() + 16 * pos * 1024, n_in, yb.data_ptr() + 16 * pos * 256, ya.shape[1], nb)
        pos += nb
    ea.synchronize(); eb.synchronize()
    print("rows below the long call untouched %s / %s; guard band untouched %s / %s" % (torch.equal(sa, ya[:, :nacq * 256]), torch.equal(sb, yb[:, :nacq * 256]),
          bool((ya[:, W:] == complex(7.0, 7.0)).all()), bool((yb[:, W:] == complex(7.0, 7.0)).all())), flush=True)
    ea.close(); eb.close()
