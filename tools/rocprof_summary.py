#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small JSON/markdown summary.

usage: tools/rocprof_summary.py <prof_dir> <out_prefix> [kernel-name-substring]
  <prof_dir>/kt/*_kernel_stats.csv            from  rocprofv3 --kernel-trace --stats
  <prof_dir>/pmc_fetch/*_counter_collection.csv   from  rocprofv3 --pmc FETCH_SIZE --kernel-trace
  <prof_dir>/pmc_write/*_counter_collection.csv   from  rocprofv3 --pmc WRITE_SIZE --kernel-trace
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM section): the summary lists the raw value and the x2 value.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0][:90]


def main():
    prof, out = sys.argv[1], sys.argv[2]
    filt = sys.argv[3] if len(sys.argv) > 3 else "qh::"
    res = {"kernels": {}}
    for f in glob.glob(os.path.join(prof, "kt", "*_kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            if filt in row["Name"]:
                res["kernels"][short(row["Name"])] = {
                    "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                    "min_ms": float(row["MinNs"]) / 1e6, "max_ms": float(row["MaxNs"]) / 1e6,
                    "total_ms": float(row["TotalDurationNs"]) / 1e6, "pct_of_gpu_time": float(row["Percentage"])}
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(prof, tag, "*_counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if filt in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            d = res["kernels"].setdefault(k, {})
            d[ctr + "_KiB_per_launch"] = sum(v) / len(v)
            d[ctr + "_launches"] = len(v)
    for k, d in res["kernels"].items():
        if "FETCH_SIZE_KiB_per_launch" in d:
            d["fetch_GB_raw"] = d["FETCH_SIZE_KiB_per_launch"] * 1024 / 1e9
            d["fetch_GB_x2_gfx950"] = 2 * d["fetch_GB_raw"]
        if "WRITE_SIZE_KiB_per_launch" in d:
            d["write_GB"] = d["WRITE_SIZE_KiB_per_launch"] * 1024 / 1e9
    json.dump(res, open(out + ".json", "w"), indent=1, sort_keys=True)
    with open(out + ".md", "w") as f:
        f.write("| kernel | calls | avg ms | min ms | max ms | FETCH GB raw | FETCH GB x2 | WRITE GB |\n|---|---|---|---|---|---|---|---|\n")
        for k, d in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("total_ms", 0)):
            f.write("| `%s` | %s | %.4f | %.4f | %.4f | %s | %s | %s |\n" % (
                k, d.get("calls", ""), d.get("avg_ms", 0), d.get("min_ms", 0), d.get("max_ms", 0),
                "%.3f" % d["fetch_GB_raw"] if "fetch_GB_raw" in d else "",
                "%.3f" % d["fetch_GB_x2_gfx950"] if "fetch_GB_x2_gfx950" in d else "",
                "%.3f" % d["write_GB"] if "write_GB" in d else ""))
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
