"""super-segments of the AGC boundary pass on a bursty signal: misses by warm-up (QH_AGC_SEGS / QH_AGC_WARM)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import quisk_amd as qh
from quisk_amd import synth
from test_gpu_wcpagc_batch import _input, _engine
x = _input(4, 700, seed=41)
for segs, warm in ((1, 400), (4, 400), (4, 2), (8, 2)):
    os.environ["QH_AGC_SEGS"] = str(segs); os.environ["QH_AGC_WARM"] = str(warm)
    e = _engine(qh, 4, [3, 1, 2, 4], 0)
    y = np.concatenate([e.process_host(np.ascontiguousarray(x[:, :300 * 1024])), e.process_host(np.ascontiguousarray(x[:, 300 * 1024:]))], axis=1)
    print("segs", segs, "warm", warm, "segments rerun", e.agc_segments_rerun(), "tiles rerun", e.agc_repairs(), "checksum %.12e" % np.abs(y).sum())
