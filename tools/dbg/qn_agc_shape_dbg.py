"""Quisk-native leg at the bench shape: the bank's process_agc against the stand-alone engine and the oracle (debug)"""
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
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = bc.setup_quisk_native(torch, qh, dev, "USB", nch=nch, n=1 << log2n)
n = L.n
ys = []
for k in range(3):
    if k == 1: L.bank.set_agc(True, bc.QN_AGC_GAIN)
    m = L.step(); torch.cuda.synchronize(dev)
    m = L.bank.out_count(0) or L.m
    ys.append(L.y[:, :L.m].clone())
y0 = ys[0]
print("out count", L.m, "no-agc max", float(y0.abs().max()))
# stand-alone engine on the same no-AGC stream (the stream repeats: x is fed three times, so call k's pre-AGC output differs; take the bank's)
L2 = bc.setup_quisk_native(torch, qh, dev, "USB", nch=nch, n=1 << log2n)
L2.x.copy_(L.x); torch.cuda.synchronize(dev)
pre = []
for k in range(3):
    L2.step(); torch.cuda.synchronize(dev); pre.append(L2.y[:, :L.m].clone())
print("pre vs bank call0", float((pre[0] - ys[0]).abs().max()))
agc = qh.QuiskAgc(nch, 48000)
for c in range(nch): agc.set_agc(c, bc.QN_AGC_GAIN)
outs = []
for k in (1, 2):
    buf = pre[k].clone(); torch.cuda.synchronize(dev)
    agc.process_ptr(buf.data_ptr(), buf.shape[1], buf.shape[1]); torch.cuda.synchronize(dev)
    outs.append(buf)
for c in (0, nch - 1):
    o = po.OracleQuiskAgc(48000)
    for k in (1, 2):
        want = o.process(pre[k][c].cpu().numpy(), False, bc.QN_AGC_GAIN)
        print("ch", c, "call", k, "bank vs oracle", rel(ys[k][c].cpu().numpy(), want), "engine vs oracle", rel(outs[k - 1][c].cpu().numpy(), want),
              "max want", np.abs(want).max(), "max bank", float(ys[k][c].abs().max()))
        d = np.abs(ys[k][c].cpu().numpy() - want)
        bad = np.nonzero(d > 1e-6 * np.abs(want).max())[0]
        print("   first bad", bad[:5], "of", bad.size)
