"""Time of the RXA chain with the LMS notch (xanf) or EMNR on: nch channels x nblk DSP blocks per call.  The LMS recurrence is
sequential per channel (one wavefront each), so the figure of merit is samples/s per channel and channels in flight.
python tools/lms_bench.py [nch] [nblk]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quisk_amd as qh
from quisk_amd import synth

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
x = synth.make_input_torch(nch, nblk * 1024, dev) if hasattr(synth, "make_input_torch") else torch.from_numpy(synth.make_input_numpy(nch, nblk * 1024)).to(dev)
y = torch.empty((nch, nblk * 256), dtype=torch.complex128, device=dev)
res = {}
for name in ("plain", "anf", "emnr", "snba"):
    e = qh.RxaEngine(nch)
    for c in range(nch):
        e.SetRXAMode(c, 1); e.RXASetPassband(c, 300.0, 3000.0); e.SetRXAAGCMode(c, 0)
    if name == "anf":
        e.SetRXAANFRun(-1, 1)
    if name == "emnr":
        e.load_emnr_tables()
        e.SetRXAEMNRRun(-1, 1)
    if name == "snba":
        e.SetRXASNBARun(-1, 1)
    e.enable_timing(True)
    torch.cuda.synchronize()
    for _ in range(2):
        e.process_ptr(x.data_ptr(), x.shape[1], y.data_ptr(), y.shape[1], nblk)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        e.process_ptr(x.data_ptr(), x.shape[1], y.data_ptr(), y.shape[1], nblk)
    e.synchronize()
    res[name] = (time.perf_counter() - t0) / 3 * 1e3
    e.close()
mid = nblk * 256
print(json.dumps({"nch": nch, "nblk": nblk, "ms_plain": round(res["plain"], 3), "ms_with_anf": round(res["anf"], 3),
                  "lms_and_bp1_ms": round(res["anf"] - res["plain"], 3),
                  "lms_Msamp_per_s_per_channel": round(mid / (res["anf"] - res["plain"]) / 1e3, 2),
                  "chain_Gsamp_per_s_with_anf": round(nch * nblk * 1024 / res["anf"] / 1e6, 2),
                  "ms_with_emnr": round(res["emnr"], 3), "emnr_and_bp1_ms": round(res["emnr"] - res["plain"], 3),
                  "emnr_frames_per_s": round(nch * nblk / 4 / (res["emnr"] - res["plain"]) * 1e3, 0),
                  "chain_Gsamp_per_s_with_emnr": round(nch * nblk * 1024 / res["emnr"] / 1e6, 2),
                  "ms_with_snba": round(res["snba"], 3), "snba_bpsnba_bp1_ms": round(res["snba"] - res["plain"], 3),
                  "snba_frames_per_s": round(nch * nblk / (res["snba"] - res["plain"]) * 1e3, 0),
                  "snba_x_realtime_per_channel": round(nblk * 256 / 48000.0 / ((res["snba"] - res["plain"]) * 1e-3), 1),
                  "chain_Gsamp_per_s_with_snba": round(nch * nblk * 1024 / res["snba"] / 1e6, 2)}))
