"""Config 4's kernels one after the other (the engine's timing mode: one stream, no fork), for rocprofv3 --kernel-trace --stats:
tools/kstats.sh out.csv python3 $PWD/tools/dbg/c4_serial.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import quisk_amd as qh
import bench_configs as bc
dev = torch.device("cuda", 0)
L = bc.setup_config4(torch, qh, dev)
L.eng.enable_timing(True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    L.step()
torch.cuda.synchronize(dev)
print(L.eng.timing_ms())
