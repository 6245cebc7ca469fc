#!/usr/bin/env python3
"""Fused half-band cascade (qh_hbc_*), one 2^26-sample fp32 stream: time per call by number of stages, and for eight stages by
segment length (QH_HBC_SEG_STEPS, steps of 2048 samples per workgroup).  Run on the GPU box."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ns):
    import torch
    import quisk_amd as qh
    dev = torch.device("cuda:0")
    n = 1 << 26
    s = torch.cuda.current_stream(dev).cuda_stream
    x = torch.randn((1, n), dtype=torch.float32, device=dev) + 1j * torch.randn((1, n), dtype=torch.float32, device=dev)
    casc = qh.HalfBandCascade(1, ns, dtype=1, stream=s)
    out = torch.empty((1, n >> ns), dtype=torch.complex64, device=dev)
    for _ in range(3):
        casc.process_ptr(x.data_ptr(), n, n, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        casc.process_ptr(x.data_ptr(), n, n, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print("%.4f" % run(int(sys.argv[1])))
    else:
        for ns in (1, 2, 4, 6, 8):
            r = subprocess.run([sys.executable, __file__, str(ns)], capture_output=True, text=True)
            print("%d stages, default segments: %s ms" % (ns, r.stdout.strip()), flush=True)
        for seg in (16, 22, 32, 43, 48, 64, 86, 128):
            r = subprocess.run([sys.executable, __file__, "8"], capture_output=True, text=True, env=dict(os.environ, QH_HBC_SEG_STEPS=str(seg)))
            print("8 stages, %3d steps per workgroup (%d workgroups): %s ms" % (seg, (32768 + seg - 1) // seg, r.stdout.strip()), flush=True)
