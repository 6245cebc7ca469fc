#!/usr/bin/env python3
"""Run tools/bench_configs.py 3 with every library variant under quisk_amd/lib/ab (same box)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1:] or ["3"]
for lib in sorted(glob.glob(os.path.join(ROOT, "quisk_amd", "lib", "ab", "libquiskhip_*.so"))):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py")] + which, env=dict(os.environ, QUISKHIP_LIB=lib),
                       capture_output=True, text=True)
    name = os.path.basename(lib)[len("libquiskhip_"):-3]
    for line in r.stdout.strip().splitlines():
        try:
            j = json.loads(line)
            print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k.endswith("ms") or "Msamp" in k}, flush=True)
        except Exception:
            pass
    if r.returncode:
        print(name, "FAILED", r.stderr[-400:])
