"""Per-call output counts of the one-receiver api walk (tests/test_gpu_quisk_api_fuzz.py) on both sides, without its asserts:
tools/dbg/api_counts.py <seed> [mode fs play]   (mode / fs / play default to the walk_sweep_all combination of the seed; NO_SETTERS=1: the calls alone)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import quisk_amd as qh
import pyoracle as oracle
import test_gpu_quisk_api_fuzz as T
from quisk_amd import rxfilter

COMBOS = [(3, 192000, 48000), (3, 111111, 96000), (4, 96000, 48000), (5, 192000, 48000), (3, 48000, 48000), (1, 133333, 48000), (4, 185185, 96000),
          (5, 96000, 192000), (3, 192000, 192000), (1, 48000, 96000), (3, 370370, 48000), (5, 53333, 48000), (0, 96000, 48000), (2, 192000, 48000),
          (7, 192000, 96000), (8, 111111, 48000), (9, 192000, 48000), (13, 96000, 48000), (10, 48000, 48000)]
seed = int(sys.argv[1])
mode, fs, play = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else COMBOS[seed % len(COMBOS)]
rng = np.random.default_rng(7000 + seed)
api = qh.quiskapi
api.open(fs, fft_size=2048, data_width=512, playback_rate=play)
ref = oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())
st = {"rx": 8300, "tx": 9100}
fI, fQ = T._filters(mode, fs)
for o in (api, ref):
    o.set_rx_mode(mode); o.set_filters(fI, fQ, T.BW[mode]); o.set_agc(20.0)
api.set_tune2(st["rx"], st["tx"]); ref.set_tune(st["rx"], st["tx"])
api.set_sidetone(0.3, 600, play, 20); ref.set_sidetone(0.3, 600, 20)
ratio = max(1, fs // 48000)
sizes = [int(rng.choice([1, 2, 3, 5, 8])) * int(rng.integers(300, 1700)) * ratio for _ in range(24)]
sizes = [min(s, 52000, 50000 * fs // play, 11000 * (fs // 48000 or 1)) for s in sizes]
n = sum(sizes)
x = T._signal(mode, 0, n, fs, float(st["rx"]), amp=2.0 ** 18)
pos = 0
tot = [0, 0]
for k, s in enumerate(sizes):
    if k and not os.environ.get("NO_SETTERS"):
        for _ in range(int(rng.integers(1, 3))):
            print("   setter", T._draw(rng, mode, fs, play, api, ref, st))
    seg = x[pos:pos + s]; pos += s
    y, want = api.process(seg), ref.process(seg)
    tot[0] += y.size; tot[1] += want.size
    print("call %2d  in %6d (sum %7d)  lib %6d  restatement %6d   running %d / %d%s" % (k, s, pos, y.size, want.size, tot[0], tot[1], "   <--" if y.size != want.size else ""))
api.close()
