"""Config 4's shape with only two FM channels (0 and 1; the others USB / AM in turn): the plain and AM channels' kernels with next to
nothing beside them on the second stream, for tools/kstats.sh (how long a kernel of theirs takes when the FM chain is not on the chip):
tools/kstats.sh out.csv python3 $PWD/tools/dbg/c4_few_fm.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import quisk_amd as qh
import bench_configs as bc


def modes(eng, c):
    m = 5 if c < 2 else (1, 6)[c % 2]
    eng.SetRXAMode(c, m)
    eng.RXASetPassband(c, *bc.C4_PASSBAND[m])


bc.c4_set_modes = modes
dev = torch.device("cuda", 0)
L = bc.setup_config4(torch, qh, dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    L.step()
torch.cuda.synchronize(dev)
