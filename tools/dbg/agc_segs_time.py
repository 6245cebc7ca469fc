"""the AGC-on leg of the driver's line (tools/bench_configs.py setup_config2_agc) timed with QH_AGC_SEGS / QH_AGC_WARM from the environment:
ms per step, segments walked again, tiles repaired; [fading]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import quisk_amd as qh
import bench_configs as B
dev = torch.device("cuda:0")
L = B.setup_config2_agc(torch, qh, dev, fading=len(sys.argv) > 1 and sys.argv[1] == "fading")
sync = lambda: torch.cuda.synchronize(dev)
t = B.timed(L.step, sync, steps=4, warmup=2)
print("QH_AGC_SEGS=%s QH_AGC_WARM=%s: %.2f ms per step, %.1f Gsamp/s, segments rerun %d, tiles rerun %d" % (
    os.environ.get("QH_AGC_SEGS"), os.environ.get("QH_AGC_WARM"), t * 1e3, L.nch * L.n_in / t / 1e9, L.eng.agc_segments_rerun(), L.eng.agc_repairs()))
