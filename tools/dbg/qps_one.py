"""a few steps of the whole-function Quisk-native USB leg (for kernel traces)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quisk_amd as qh
import bench_configs as bc
dev = torch.device("cuda", 0)
L = bc.setup_quisk_native(torch, qh, dev, "USB")
sync = lambda: torch.cuda.synchronize(dev)
t = bc.timed(L.step, sync, steps=4, warmup=3)
print("ms %.3f" % (t * 1e3))
