"""Quisk-native whole-function leg: time against the number of pieces (debug)"""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quisk_amd as qh
import bench_configs as bc
dev = torch.device("cuda", 0)
sync = lambda: torch.cuda.synchronize(dev)
for name in sys.argv[1:] or ["USB"]:
    for P in (1, 2, 4, 8, 16, 32):
        L = bc.setup_quisk_native(torch, qh, dev, name, pieces=P)
        t = bc.timed(L.step, sync, steps=8, warmup=3)
        print(name, "pieces", P, "ms %.3f  Gsamp/s %.1f" % (t * 1e3, 256 * (1 << 20) / t / 1e9), flush=True)
        del L
    L = bc.setup_quisk_native(torch, qh, dev, name, whole=False)
    t = bc.timed(L.step, sync, steps=8, warmup=3)
    print(name, "bank only ms %.3f" % (t * 1e3))
