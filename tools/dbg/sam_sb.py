"""SAM-L channels in long calls: the time-segmented chains against the one-wavefront kernel (QH_SAM_MIN raises the tiling threshold)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda:0")
nch, nblk = 128, int(os.environ.get("NBLK", "1024"))
n_in = nblk * 1024
x = synth.make_mode_input_torch(["am"] * nch, n_in, dev)
y = torch.empty((nch, nblk * 256), dtype=torch.complex128, device=dev)
for sb in (0, 1):
    e = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 10)
        e.RXASetPassband(c, -4000.0, 4000.0); e.SetRXAAGCMode(c, 0); e.SetRXAAGCFixed(c, 0.0); e.SetRXAAMDSBMode(c, sb)
    f = lambda: e.process_ptr(x.data_ptr(), n_in, y.data_ptr(), nblk * 256, nblk)
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    print("SAM sbmode %d: %.2f ms per call of %d x 2^%d samples = %.1f Gsamp/s (pll tiles re-run %d)" % (sb, t * 1e3, nch, n_in.bit_length() - 1, nch * n_in / t / 1e9, e.pll_repairs()), flush=True)
