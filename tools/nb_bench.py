"""Noise blanker throughput on one GPU: `nch` streams x `n` samples per launch, fp64 complex, input resident in HBM.
Algorithmic bytes: 16 B read + 16 B written per sample.  python tools/nb_bench.py [nch] [log2 n] [rate] [pulses: 0/1]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import quisk_amd as qh
from quisk_amd import synth

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 21)
rate = int(sys.argv[3]) if len(sys.argv) > 3 else 192000
pulses = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
seg = synth.impulsive_input(1, 1 << 16, seed=1)[0] if pulses else (np.random.default_rng(1).standard_normal(1 << 16) * 1e6 + 0j)
x = torch.from_numpy(np.tile(seg, n // seg.size)).to(dev).repeat(nch, 1).contiguous()
y = torch.empty_like(x)
stream = torch.cuda.Stream()
nb = qh.NoiseBlanker(nch, rate, 2, stream=stream.cuda_stream)
torch.cuda.synchronize()
for _ in range(3):
    nb.process_ptr(x.data_ptr(), n, y.data_ptr(), n, n)
nb.synchronize()
K = 10
with torch.cuda.stream(stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(K):
        nb.process_ptr(x.data_ptr(), n, y.data_ptr(), n, n)
    e1.record(stream)
nb.synchronize()
ms = e0.elapsed_time(e1) / K
print(json.dumps({"nch": nch, "n": n, "rate": rate, "pulses": bool(pulses), "ms_per_launch": round(ms, 4),
                  "Gsamp_per_s": round(nch * n / ms / 1e6, 2), "GB_per_s": round(32.0 * nch * n / ms / 1e6, 1),
                  "frac_of_8TBps": round(32.0 * nch * n / ms / 1e6 / 8000.0, 3)}))
