"""one walk of tests/test_gpu_quisk_api_fuzz.py again, call by call"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
import pyoracle as oracle
from quisk_amd import rxfilter
import test_gpu_quisk_api_fuzz as T
from test_gpu_quisk_process_bank import BW, _filters, _signal
seed, mode, fs, play = [int(v) for v in sys.argv[1:5]]
rng = np.random.default_rng(7000 + seed)
SKIP = set(os.environ.get("SKIP", "").split(","))
class Skip:
    """the same draws, some setters left out on both sides"""
    def __init__(self, o): self._o = o
    def __getattr__(self, name):
        if name in SKIP: return lambda *a, **k: None
        return getattr(self._o, name)
qh.quiskapi.open(fs, playback_rate=play)
api = Skip(qh.quiskapi)
ref = Skip(oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables()))
st = {"rx": 8300, "tx": 9100}
fI, fQ = _filters(mode, fs)
for o in (api, ref):
    o.set_rx_mode(mode); o.set_filters(fI, fQ, BW[mode]); o.set_agc(20.0)
api.set_tune2(st["rx"], st["tx"]); ref.set_tune(st["rx"], st["tx"])
api.set_sidetone(0.3, 600, play, 20); ref.set_sidetone(0.3, 600, 20)
ratio = max(1, fs // 48000)
sizes = [int(rng.choice([1, 2, 3, 5, 8])) * int(rng.integers(300, 1700)) * ratio for _ in range(24)]
sizes = [min(s, 52000, 50000 * fs // play, 11000 * (fs // 48000 or 1)) for s in sizes]
n = sum(sizes)
x = _signal(mode, 0, n, fs, float(st["rx"]), amp=2.0 ** 18)
x[5000::9973] += 2.0 ** 21
x[n // 2:n // 2 + n // 6] *= 0.01
pos = 0
for k, s in enumerate(sizes):
    if k:
        for _ in range(int(rng.integers(1, 3))):
            print("   ", k, T._draw(rng, mode, fs, play, api, ref, st))
    seg = x[pos:pos + s]; pos += s
    y, w = api.process(seg), ref.process(seg)
    if w.size:
        d = np.abs(y - w)
        i = int(d.argmax())
        print("call %2d n %6d out %6d: re err %.1e im err %.1e of %.3e (|re| %.2e |im| %.2e) worst at %d" % (k, s, w.size, np.abs(y.real - w.real).max(), np.abs(y.imag - w.imag).max(),
              np.abs(w).max(), np.abs(w.real).max(), np.abs(w.imag).max(), i))
qh.quiskapi.close()
