#!/usr/bin/env python3
"""Run tools/bench_configs.py <config> once per library variant built by tools/ab_bench.py build (on the GPU box)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "5"
keys = sys.argv[2:] or ["fused_cascade_only_ms", "fused_ms", "pan_ms", "both_ms", "ms"]
for rep in range(2):
    for lib in sorted(glob.glob(os.path.join(ROOT, "quisk_amd", "lib", "ab", "libquiskhip_*.so"))):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), cfg], env=dict(os.environ, QUISKHIP_LIB=lib),
                           capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            print(os.path.basename(lib), {k: round(j[k], 4) for k in keys if k in j}, flush=True)
        except Exception:
            print(os.path.basename(lib), "FAILED", r.stdout[-200:], r.stderr[-300:], flush=True)
