"""config-2 chain with the AGC state machine running (SetRXAAGCMode 1..4) against the fixed-gain mode the benchmark uses"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda:0")
nch, nblk = 256, int(os.environ.get("NBLK", "1024"))
n_in = nblk * 1024
x = synth.make_mode_input_torch(["usb"] * nch, n_in, dev)
y = torch.empty((nch, nblk * 256), dtype=torch.complex128, device=dev)
for mode in (0, 3, 2, 4):
    e = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
        e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, mode); e.SetRXAAGCFixed(c, 0.0)
    f = lambda: e.process_ptr(x.data_ptr(), n_in, y.data_ptr(), nblk * 256, nblk)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    print("AGC mode %d: %.2f ms per call of %d x 2^%d samples = %.1f Gsamp/s" % (mode, t * 1e3, nch, n_in.bit_length() - 1, nch * n_in / t / 1e9), flush=True)
