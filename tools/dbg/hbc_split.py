import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import quisk_amd as qh
dev = torch.device("cuda:0")
n = 1 << 26
s = torch.cuda.current_stream(dev).cuda_stream
x = torch.randn((1, n), dtype=torch.float32, device=dev) + 1j * torch.randn((1, n), dtype=torch.float32, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for ns in (1, 2, 3, 4, 8):
    c = qh.HalfBandCascade(1, ns, dtype=1, stream=s)
    out = torch.empty((1, n >> ns), dtype=torch.complex64, device=dev)
    print(ns, "stages: %.4f ms" % timed(lambda: c.process_ptr(x.data_ptr(), n, n, out.data_ptr(), out.shape[1])), flush=True)
for a in (2, 3, 4):
    b = 8 - a
    ca = qh.HalfBandCascade(1, a, dtype=1, stream=s)
    cb = qh.HalfBandCascade(1, b, dtype=1, stream=s)
    mid = torch.empty((1, n >> a), dtype=torch.complex64, device=dev)
    out = torch.empty((1, n >> 8), dtype=torch.complex64, device=dev)
    def f():
        ca.process_ptr(x.data_ptr(), n, n, mid.data_ptr(), mid.shape[1])
        cb.process_ptr(mid.data_ptr(), n >> a, n >> a, out.data_ptr(), out.shape[1])
    print("%d + %d stages: %.4f ms" % (a, b, timed(f)), flush=True)
