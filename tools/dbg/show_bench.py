"""print the legs of a bench.py JSON line (tools/dbg/show_bench.py <file>)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "meters off", d.get("value_meters_off"))
print("roofline", d["roofline"])
print("cpu", d.get("cpu_baseline"))
for k, v in d.get("other_configs", {}).items():
    if "modes" in v:
        print(k, [(m["mode"], round(m["Msamp_per_s"]), round(m["ms_per_step"], 3)) for m in v["modes"]])
    else:
        print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "Msamp_per_s", "frac_of_hbm_peak", "dominant_kernel_ms")})
