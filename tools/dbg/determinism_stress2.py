"""N identical mixed-mode engines (USB / AM / FM by thirds, 256 channels) given the same short calls, issued round-robin so that their
kernels share the GPU, beside a stream of unrelated work that perturbs the scheduling: the same bits from every engine?
usage: determinism_stress2.py [iterations] [engines] [noise 0/1]      (QH_DBG_FORMS in the environment switches engine forms off)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
neng = int(sys.argv[2]) if len(sys.argv) > 2 else 4
noise = int(sys.argv[3]) if len(sys.argv) > 3 else 1
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
ns = torch.cuda.Stream()
na = torch.randn(3072, 3072, device=dev)
junk = torch.empty(1 << 26, device=dev)
bad = 0
t0 = time.time()
for it in range(iters):
    engs = [make() for _ in range(neng)]
    ys = [torch.full((nch, nblk_call * ncalls * 256), complex(float('nan'), float('nan')), dtype=torch.complex128, device=dev) for _ in range(neng)]
    torch.cuda.synchronize()
    order = it % 3
    if order == 0:          # engine by engine (the suite's order)
        seq = [(e, k) for e in range(neng) for k in range(ncalls)]
    elif order == 1:        # call by call
        seq = [(e, k) for k in range(ncalls) for e in range(neng)]
    else:                   # staggered: engine e runs e calls behind engine 0
        seq = sorted([(e, k) for e in range(neng) for k in range(ncalls)], key=lambda p: (p[1] + p[0], p[0]))
    for i, (e, k) in enumerate(seq):
        if noise and i % 3 == 0:
            with torch.cuda.stream(ns):
                if (i // 3) % 2: nb = na @ na
                else: junk.mul_(1.0001)
        engs[e].process_ptr(x.data_ptr() + 16 * k * nblk_call * 1024, n_in, ys[e].data_ptr() + 16 * k * nblk_call * 256, ys[e].shape[1], nblk_call)
    for e in engs: e.synchronize()
    torch.cuda.synchronize()
    for e in range(1, neng):
        if not torch.equal(ys[0], ys[e]):
            d = (ys[0] - ys[e]).abs()
            rows = (d.amax(dim=1) > 0).nonzero().flatten().tolist()
            ch = int(d.amax(dim=1).argmax().item())
            first = int((d[ch] > 0).nonzero()[0].item())
            by_mode = {m: sum(1 for r in rows if modes[r % 3] == m) for m in modes}
            print("iteration %d (order %d): engine %d differs from engine 0 in %d channels %r; worst channel %d (mode %d) from sample %d (call %d), max %.3e of %.3e"
                  % (it, order, e, len(rows), by_mode, ch, modes[ch % 3], first, first // (nblk_call * 256), float(d.max()), float(ys[0].abs().max())), flush=True)
            bad += 1
    for e in engs: e.close()
print("forms %s: %d iterations x %d engines (noise %d), %d engine runs with different bits, %.1f s" % (os.environ.get("QH_DBG_FORMS", "0"), iters, neng, noise, bad, time.time() - t0), flush=True)
