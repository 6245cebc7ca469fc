#!/usr/bin/env python3
"""Extracts the `Filters` lowpass prototype table (reference filters.py, pure data: key = bandwidth at 24 ksps / 2)
into quisk_amd/data/quisk_rx_filters.npz.  Runs only in the build container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
import filters as ref_filters       # noqa: E402

out = {"k%d" % k: np.asarray(v, dtype=np.float64) for k, v in ref_filters.Filters.items()}
path = os.path.join(ROOT, "quisk_amd", "data", "quisk_rx_filters.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes", sorted(ref_filters.Filters))
