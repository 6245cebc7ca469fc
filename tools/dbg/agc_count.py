"""Experiment build -DQH_AGC_COUNT (QUISKHIP_LIB points at it): how the AGC boundary walks spend their rounds on the leg's two inputs."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import quisk_amd as qh
from quisk_amd import lib as qlib
import bench_configs as bc
dev = torch.device("cuda:0")
L = C.CDLL(qlib.LIB_PATH)
L.qh_dbg_agc_counts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
def counts():
    a = (C.c_ulonglong * 20)()
    assert L.qh_dbg_agc_counts(a, 1) == 0
    return list(a)
names = ["attack scans", "decay scans", "hang-decay scans", "fast-decay scans", "hang holds", "floor holds", "steps from st0", "st1", "st2", "st3", "st4",
         "attack samples", "decay samples", "hang-decay samples", "fast-decay samples", "hang samples", "floor samples"]
for fading in (False, True):
    S = bc.setup_config2_agc(torch, qh, dev, fading=fading)
    for _ in range(2): S.step()
    torch.cuda.synchronize(); counts()
    S.step(); torch.cuda.synchronize()
    c = counts()
    tot = S.nch * S.n_out
    print("fading" if fading else "steady", "(per channel sample of the call; one step)")
    for n, v in zip(names, c): print("   %-20s %12d  %.4f" % (n, v, v / tot))
    del S
    torch.cuda.empty_cache()
