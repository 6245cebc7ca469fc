"""two identical mixed-mode engines (USB / AM / FM, 256 channels) fed the same short calls side by side, again and again: the same bits?
(a rare mismatch in tests/test_gpu_properties_fullsize.py's acquisition phase asked for this)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda:0")
nch, nblk_call, ncalls = 256, 16, 10
modes, kinds = [1, 6, 5], {1: "usb", 6: "am", 5: "fm"}
n_in = nblk_call * ncalls * 1024
x = torch.from_numpy(np.stack([synth.make_mode_input_numpy(kinds[modes[c % 3]], c, n_in) for c in range(nch)])).to(dev)
def make():
    e = qh.RxaEngine(nch)
    for c in range(nch):
        m = modes[c % 3]
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1)
        e.SetRXAMode(c, m); e.SetRXAAGCMode(c, 0); e.SetRXAAGCFixed(c, 0.0)
        e.RXASetPassband(c, *((300.0, 3000.0) if m == 1 else (-4000.0, 4000.0) if m == 6 else (-8000.0, 8000.0)))
    return e
bad = 0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for it in range(iters):
    ea, eb = make(), make()
    ya = torch.empty((nch, nblk_call * ncalls * 256), dtype=torch.complex128, device=dev)
    yb = torch.empty_like(ya)
    ya.fill_(complex(float('nan'), float('nan'))); yb.fill_(complex(1e300, -1e300))       # whatever is not written, or read before it is, shows
    torch.cuda.synchronize()
    for e, y in ((ea, ya), (eb, yb)):
        for k in range(ncalls):
            e.process_ptr(x.data_ptr() + 16 * k * nblk_call * 1024, n_in, y.data_ptr() + 16 * k * nblk_call * 256, ya.shape[1], nblk_call)
    ea.synchronize(); eb.synchronize()
    if not torch.equal(ya, yb):
        d = (ya - yb).abs()
        ch = int(d.amax(dim=1).argmax().item())
        first = int((d[ch] > 0).nonzero()[0].item())
        nbad = int((d.amax(dim=1) > 0).sum().item())
        print("iteration %d: %d channels differ; worst channel %d (mode %d) from sample %d (call %d), max %.3e of %.3e" % (it, nbad, ch, modes[ch % 3], first, first // (nblk_call * 256), float(d.max()), float(ya.abs().max())), flush=True)
        bad += 1
    ea.close(); eb.close()
    if it % 2 == 0:         # another engine's allocations in between: what the next pair is handed back is not its own last state
        o = qh.RxaEngine(64 + it % 7)
        for c in range(o.nch if hasattr(o, 'nch') else 64):
            o.SetRXAMode(c, 6 if c % 2 else 5); o.SetRXAAGCMode(c, 3)
        o.process_host(np.ascontiguousarray(x[:64 + it % 7, :48 * 1024].cpu().numpy() * (1.0 + it)))
        o.close()
print("%d iterations, %d with different bits" % (iters, bad))
