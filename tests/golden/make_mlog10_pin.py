#!/usr/bin/env python3
"""Pins the one table behind every dB value WDSP reports (meters, analyzer pixels): wdsp/meterlog10.c holds mtable[2048], which the
restatements (oracle/wdsp_oracle.c wo_mlog10, quisk_amd/csrc/qh_wave.hpp mlog10_dev) replace by log2(1 + m / 2048).  Run HERE (needs
/root/reference): reads the table's numbers out of the reference file, compares them with that formula and writes the result --
numbers only, no source text -- to tests/golden/mlog10_pin.json, which tests/test_oracle_wdsp.py checks."""
import json
import os
import re

import numpy as np

REF = "/root/reference/wdsp/meterlog10.c"
src = open(REF).read()
i = src.index("mtable")
body = src[src.index("{", i):src.index("};", i)]
vals = np.array([float(v) for v in re.findall(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?", body)])
m = np.arange(vals.size)
dev = float(np.abs(vals - np.log2(1.0 + m / 2048.0)).max())
mbits = int(re.search(r"int\s+mbits\s*=\s*(\d+)", src).group(1))
mconv = float(re.search(r"mconv\s*=\s*([0-9.eE+-]+)", src).group(1))
out = {"entries": int(vals.size), "mbits": mbits, "mconv": mconv, "max_abs_deviation_from_log2_1_plus_m_over_2048": dev,
       "first": float(vals[0]), "last": float(vals[-1])}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mlog10_pin.json"), "w"), indent=1, sort_keys=True)
print(out)
