"""split Rx/Tx switched on in mid-stream, with the things that were on around it in the failing walk, one at a time"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import quisk_amd as qh
import pyoracle as oracle
from quisk_amd import rxfilter
from test_gpu_quisk_process_bank import BW, _filters, _signal
def run(name, mode, fs, play, split, notch=0, key=False, invert=0, nb=0, back=False, notch_at=-1, tx=9100):
    api = qh.quiskapi
    api.open(fs, playback_rate=play)
    ref = oracle.OracleQuiskBlock(fs, play, rxfilter.coefficient_tables())
    fI, fQ = _filters(mode, fs)
    for o in (api, ref):
        o.set_rx_mode(mode); o.set_filters(fI, fQ, BW[mode]); o.set_agc(20.0); o.set_auto_notch(notch); o.invert_spectrum(invert); o.set_noise_blanker(nb)
    api.set_tune2(8300, tx); ref.set_tune(8300, tx)
    api.set_sidetone(0.3, 600, play, 20); ref.set_sidetone(0.3, 600, 20)
    n = 4000 * max(1, fs // 48000)
    x = _signal(mode, 0, 12 * n, fs, 8300.0, amp=2.0 ** 18)
    errs = []
    for k in range(12):
        if key and k == 2:
            api.set_key_state(1, 0, 0, 0); ref.set_key_state(1, 0, 0, 0)
        if key and k == 3:
            api.set_key_state(0, 0, 0, 0); ref.set_key_state(0, 0, 0, 0)
        if k == notch_at:
            api.set_auto_notch(1); ref.set_auto_notch(1)
        if k == 5:
            api.set_split_rxtx(split); ref.set_split_rxtx(split)
        if back and k == 9:
            api.set_split_rxtx(0); ref.set_split_rxtx(0)
        seg = x[k * n:(k + 1) * n]
        y, w = api.process(seg), ref.process(seg)
        assert y.size == w.size, (name, k, y.size, w.size)
        errs.append(np.abs(y - w).max() / max(np.abs(w).max(), 1.0) if w.size else 0.0)
    api.close()
    print("%-44s %s" % (name, " ".join("%.0e" % e for e in errs)), flush=True)
run("USB split 3 (bank 0 alone), notch on at call 3", 3, 192000, 48000, 3, notch_at=3)
run("USB split 4 (bank 1 alone), notch on at call 3", 3, 192000, 48000, 4, notch_at=3)
run("USB split 4, notch on at 3, tx = rx", 3, 192000, 48000, 4, notch_at=3, tx=8300)
run("USB split 4, notch on at call 0", 3, 192000, 48000, 4, notch_at=0)
run("USB split 4, notch on at call 1", 3, 192000, 48000, 4, notch_at=1)
