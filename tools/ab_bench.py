#!/usr/bin/env python3
"""A/B kernel experiments: build libquiskhip variants with -D overrides (here, on CPU) and print the
shell line that benches them back to back in ONE process-per-variant run on the GPU box.

  tools/ab_bench.py build  name:DEF=1,DEF2=0  name2:...     -> quisk_amd/lib/ab/libquiskhip_<name>.so
  tools/ab_bench.py run    [bench args]                     -> runs every built variant (on the GPU box)
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "quisk_amd", "lib", "ab")


def main():
    if sys.argv[1] == "build":
        from quisk_amd import build as qb
        for spec in sys.argv[2:]:
            name, _, defs = spec.partition(":")
            defines = [d for d in defs.split(",") if d]
            out = os.path.join(AB, "libquiskhip_%s.so" % name)
            qb.build(force=True, defines=defines, out=out)
            print("built", out, defines)
    else:
        args = sys.argv[2:] or ["--log2-samples", "21", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-live-traffic"]
        for lib in sorted(glob.glob(os.path.join(AB, "libquiskhip_*.so"))):
            env = dict(os.environ, QUISKHIP_LIB=lib)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True)
            name = os.path.basename(lib)[len("libquiskhip_"):-3]
            try:
                j = json.loads(r.stdout.strip().splitlines()[-1])
                print("%-28s %9.1f Msamp/s  step %.3f ms  front %.3f  band %.3f  rest %.3f  gain %.4f  meters off: %s" % (
                    name, j["value"], j["ms_per_step"], j["kernel_ms"]["front_shift_resample"], j["kernel_ms"]["band_nbp"],
                    j["kernel_ms"]["state_bookkeeping"], j["check_inband_gain"], j.get("value_meters_off")), flush=True)
            except Exception:
                print(name, "FAILED", r.stdout[-300:], r.stderr[-600:], flush=True)


if __name__ == "__main__":
    main()
