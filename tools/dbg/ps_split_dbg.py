import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch  # noqa
import quisk_amd
from quisk_amd import rxfilter
from oracle import pyoracle as oracle
from test_gpu_quisk_process_samples import _filters, _two_tone, _both
quisk_amd.load()
api = quisk_amd.quiskapi
fs = 192000
api.open(fs, playback_rate=48000)
ref = oracle.OracleQuiskBlock(fs, 48000, rxfilter.coefficient_tables())
fI, fQ = _filters("USB", 3, 2700)
gI, gQ = _filters("USB", 3, 2400)
for o in (api, ref):
    o.set_rx_mode(3)
    o.set_filters(fI, fQ, 2700)
    o.set_filters(gI, gQ, 2400, 1)
    o.set_filters(fI, fQ, 2700)
    o.set_multirx_mode(1, 3); o.set_multirx_freq(1, -15000); o.set_multirx_play_method(1)
api.set_tune2(10000, 21000); ref.set_tune(10000, 21000)
sizes = [1001, 2003, 997, 4099, 1501, 3001, 2999, 1777, 3333, 2048, 4096, 1234, 4096, 4096, 4096, 1000, 3000, 4096, 4096, 4096]
x = _two_tone(fs, sum(sizes), 10900.0, 21900.0, 15)
xs = _two_tone(fs, sum(sizes), -14200.0, 30000.0, 16)
pos = 0
def rr(a, b):
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / max(np.sum(np.abs(b) ** 2), 1e-300)))
for i, s in enumerate(sizes):
    if i == 3:
        _both(api, ref, "set_split_rxtx", 1)
    if i == 9:
        _both(api, ref, "set_split_rxtx", 0)
        _both(api, ref, "set_multirx_play_channel", 1)
    if i == 15:
        _both(api, ref, "set_multirx_play_channel", -1)
        _both(api, ref, "set_split_rxtx", 2)
    seg, sub = x[pos:pos + s], xs[pos:pos + s]; pos += s
    api.multirx_samples(1, sub); ref.multirx_samples(1, sub)
    a, b = api.process(seg), ref.process(seg)
    print(i, s, a.size, b.size, "re %.2e im %.2e" % (rr(a.real, b.real), rr(a.imag, b.imag)), "max", np.abs(b.real).max(), np.abs(b.imag).max())
api.close()
