"""first call of a fresh engine, 256 channels x 2^22 samples: is every channel's output there from the start?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import quisk_amd as qh
from quisk_amd import synth
dev = torch.device("cuda:0")
nch, nblk = 256, 4096
x = synth.make_mode_input_torch(["usb"] * nch, nblk * 1024, dev)
print("input |x| mean per channel, first 2^20 samples:", [float(x[c, :1 << 20].abs().mean()) for c in (0, 100, 232, 233, 240, 255)])
y = torch.empty((nch, nblk * 256), dtype=torch.complex128, device=dev)
for mode in (0, 3):
    e = qh.RxaEngine(nch, stream=torch.cuda.current_stream(dev).cuda_stream)
    for c in range(nch):
        e.SetRXAShiftRun(c, 1); e.SetRXAShiftFreq(c, synth.shift_freq(c)); e.RXANBPSetRun(c, 1); e.SetRXAMode(c, 1)
        e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, mode); e.SetRXAAGCFixed(c, 0.0)
    y.zero_()
    e.process_ptr(x.data_ptr(), nblk * 1024, y.data_ptr(), nblk * 256, nblk)
    torch.cuda.synchronize()
    for c in (0, 100, 232, 233, 240, 255):
        a = y[c].abs()
        nz = torch.nonzero(a[:600000] > 1e-12)
        print("AGC mode", mode, "channel", c, "first non-zero output sample", int(nz[0]) if len(nz) else None, "mean |y| first 2^18 %.3e, last 2^18 %.3e" % (float(a[:1 << 18].mean()), float(a[-(1 << 18):].mean())))
