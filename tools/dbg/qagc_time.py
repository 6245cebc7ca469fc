"""Quisk-native chain with process_agc on: time per call, with and without the AGC (tools/dbg; not part of the product)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import quisk_amd as qh
from quisk_amd import rxfilter

dev = torch.device("cuda:0")
nch, n, fs = 256, 1 << int(os.environ.get("LOG2N", "20")), 192000
gain = float(os.environ.get("RG", "5000"))
for name, mode, bw in (("USB", rxfilter.USB, 2700), ("AM", rxfilter.AM, 6000)):
    for agc in (0, 1):
        bank = qh.QuiskRxBank(nch, fs, mode, bw, stream=torch.cuda.current_stream(dev).cuda_stream)
        rate = bank.get_filter_rate()
        fI, fQ = rxfilter.make_filter_coef(rate, None, bw, rxfilter.get_filter_center(name, bw))
        for c in range(nch):
            bank.set_tune(c, 1000 * (c % 40) - 20000)
        bank.set_filters(-1, fI, fQ)
        if agc:
            bank.set_agc(1, gain)
        x = (torch.randn((nch, n), dtype=torch.float64, device=dev) + 1j * torch.randn((nch, n), dtype=torch.float64, device=dev)) * 2.0 ** 22
        m = bank.out_count(n)
        y = torch.empty((nch, m + 64), dtype=torch.complex128, device=dev)
        for _ in range(2):
            bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), m + 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 4
        for _ in range(K):
            bank.process_ptr(x.data_ptr(), n, n, y.data_ptr(), m + 64)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / K
        print("%s agc=%d  %.3f ms  %.1f Gsamp/s  out rms %.3g" % (name, agc, t * 1e3, nch * n / t / 1e9, float(y[:, :m].abs().pow(2).mean().sqrt())), flush=True)
        del bank, x, y
