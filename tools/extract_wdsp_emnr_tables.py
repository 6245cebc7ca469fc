#!/usr/bin/env python3
"""Reads WDSP's run-time data files for EMNR -- wdsp/calculus (GG, GGS: 2 x 241 x 241 doubles, emnr.c:317-326) and
wdsp/zetaHat.bin (rows, cols, gmin, gmax, ximin, ximax, zetaHat[], zetaValid[], emnr.c:206-238) -- into
quisk_amd/data/wdsp_emnr_tables.npz.  Runs only where /root/reference is mounted; the tables are data the reference
itself loads with fopen, handed to the library by the caller like the filters.h coefficient tables."""
import os
import struct
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = np.fromfile(os.path.join(ref, "wdsp", "calculus"), dtype="<f8")
assert raw.size >= 2 * 241 * 241, raw.size
GG, GGS = raw[:241 * 241].copy(), raw[241 * 241:2 * 241 * 241].copy()
with open(os.path.join(ref, "wdsp", "zetaHat.bin"), "rb") as f:
    rows, cols = struct.unpack("<ii", f.read(8))
    gmin, gmax, ximin, ximax = struct.unpack("<dddd", f.read(32))
    zeta = np.frombuffer(f.read(8 * rows * cols), dtype="<f8").copy()
    valid = np.frombuffer(f.read(4 * rows * cols), dtype="<i4").copy()
out = os.path.join(root, "quisk_amd", "data", "wdsp_emnr_tables.npz")
np.savez_compressed(out, GG=GG, GGS=GGS, zeta_hat=zeta, zeta_valid=valid, zeta_dims=np.array([rows, cols], dtype=np.int32),
                    zeta_range=np.array([gmin, gmax, ximin, ximax]))
print(out, rows, cols, gmin, gmax, ximin, ximax, GG[:3], GGS[:3], os.path.getsize(out))
