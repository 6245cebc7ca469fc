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
x = (torch.randn((nch, n), dtype=torch.float64, device=dev) + 1j * torch.randn((nch, n), dtype=torch.float64, device=dev)) * 2.0 ** 20
y = torch.empty((nch, n // 32), dtype=torch.complex128, device=dev)
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def both(nsl, order):
    m = n // nsl
    def f():
        for i in range(nsl):
            if order == 0:
                bank.process_ptr(x.data_ptr() + 16 * i * m, n, m, y.data_ptr() + 16 * i * (m // 32), n // 32)
                pan.feed_ptr(x.data_ptr() + 16 * i * m, n, m)
            else:
                pan.feed_ptr(x.data_ptr() + 16 * i * m, n, m)
                bank.process_ptr(x.data_ptr() + 16 * i * m, n, m, y.data_ptr() + 16 * i * (m // 32), n // 32)
    return f
print("fir %.4f pan %.4f" % (timed(lambda: bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), n // 32)), timed(lambda: pan.feed_ptr(x.data_ptr(), n, n))))
for nsl in (1, 2, 4, 8, 16, 32):
    print("slices %2d (%4d MB each): fir->pan %.4f ms   pan->fir %.4f ms" % (nsl, nch * (n // nsl) * 16 >> 20, timed(both(nsl, 0)), timed(both(nsl, 1))), flush=True)
