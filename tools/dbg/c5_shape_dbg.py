"""config 5 at the bench shape, stage by stage against the oracle (debug)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quisk_amd as qh
import bench_configs as bc
from oracle import pyoracle as po
po.build()
dev = torch.device("cuda", 0)
def rel(a, b): return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(b) ** 2)))
for log2n in (22, 26):
    n = 1 << log2n
    L = bc.setup_config5(torch, qh, dev, n=n, unfused=False)
    m = L.step_fused()
    torch.cuda.synchronize(dev)
    x = L.x[0].cpu().numpy().astype(np.complex128)
    y = x
    for _ in range(8):
        y = po.OracleHB45().cDecim2(y)
    c8 = L.bufs[-1][0, :n >> 8].cpu().numpy().astype(np.complex128)
    print(log2n, "cascade", y.size, c8.size, rel(c8, y))
    y5 = po.OracleFir(L.taps245).cDecimate(y, 5)
    g5 = L.y5[0, :y5.size].cpu().numpy().astype(np.complex128)
    print(log2n, "d5", y5.size, m, rel(g5, y5))
    want = np.convolve(y5, L.bp)[:y5.size]
    got = L.yo[0, :m].cpu().numpy().astype(np.complex128)
    print(log2n, "bp", rel(got, want), np.abs(want).max(), np.abs(got).max())
    print(got[3000:3004], want[3000:3004])
