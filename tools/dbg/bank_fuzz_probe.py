"""one walk of tests/test_gpu_quisk_bank_fuzz.py again, call by call: errors, levels and the squelch flags of both sides"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
import pyoracle as oracle
import test_gpu_quisk_bank_fuzz as T
from test_gpu_quisk_process_bank import BW, _filters, _refs, _signal
seed, mode, fs, play = [int(v) for v in sys.argv[1:5]]
rng = np.random.default_rng(9000 + seed)
NCH = T.NCH
tunes = [7000 + 1300 * c for c in range(NCH)]
filt = [_filters(mode, fs)] * NCH
bank = qh.QuiskProcessBank(NCH, fs, mode, BW[mode], playback_rate=play)
refs = _refs(oracle, NCH, fs, play, mode, tunes, filt)
for c in range(NCH):
    bank.set_tune(c, tunes[c]); bank.set_filters(c, *filt[c])
bank.set_agc(20.0); [r.set_agc(20.0) for r in refs]
ratio = max(1, fs // 48000)
sizes = [int(rng.choice([1, 2, 3, 5, 8])) * int(rng.integers(300, 1700)) * ratio // 1 for _ in range(22)]
sizes = [min(s, 52000, 50000 * fs // play) for s in sizes]
n = sum(sizes)
x = np.stack([_signal(mode, c, n, fs, float(tunes[c]), amp=2.0 ** 18) for c in range(NCH)])
x[:, 5000::9973] += 2.0 ** 21
x[:, n // 2:n // 2 + n // 6] *= 0.01
pos = 0
for k, s in enumerate(sizes):
    if k:
        for _ in range(int(rng.integers(1, 3))):
            print("   ", k, T._draw(rng, mode, fs, bank, refs))
    seg = x[:, pos:pos + s]; pos += s
    y = bank.process_host(seg)
    line = []
    for c in range(NCH):
        w = refs[c].process(seg[c])
        e = np.abs(y[c] - w).max() / max(np.abs(w).max(), 1.0) if w.size else 0.0
        nz = lambda v: int(np.count_nonzero(v))
        line.append("%.1e (nonzero %d/%d of %d, ref flags %d)" % (e, nz(y[c]), nz(w), w.size, refs[c].squelch_flags()))
    print("call %2d n %6d flags %s: %s" % (k, s, list(bank.squelch_flags()), "  ".join(line)))
    if rng.integers(0, 4) == 0:                  # (the walk reads get_graph now and then: the same draws here)
        rng.choice([1.0, 1.0, 2.0, 4.0]); rng.choice([0.0, 0.0, 5000.0, -12000.0])
