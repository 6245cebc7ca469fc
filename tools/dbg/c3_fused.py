import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import quisk_amd as qh
dev = torch.device("cuda:0")
nch, n, fs = 64, 1 << 20, 1536000.0
k = np.arange(1023) - 511
taps = np.sinc(k / 32.0) / 32.0 * np.blackman(1023)
s = torch.cuda.current_stream(dev).cuda_stream
bank = qh.FirBank(nch, taps, 32, stream=s)
pan = qh.Panadapter(nch, 16384, 1024, fs, stream=s)
fused = qh.Panadapter(nch, 16384, 1024, fs, stream=s)
fused.attach_fir(taps, 32)
x = (torch.randn((nch, n), dtype=torch.float64, device=dev) + 1j * torch.randn((nch, n), dtype=torch.float64, device=dev)) * 2.0 ** 20
y = torch.empty((nch, n // 32), dtype=torch.complex128, device=dev)
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("fir %.4f  pan %.4f  both %.4f  fused %.4f ms" % (
    timed(lambda: bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 32)), timed(lambda: pan.feed_ptr(x.data_ptr(), n, n)),
    timed(lambda: (bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 32), pan.feed_ptr(x.data_ptr(), n, n))),
    timed(lambda: fused.feed_decimate_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 32))))
