#!/usr/bin/env python3
"""Run on the GPU box: one rocprofv3 --pmc pass per counter group over a short bench.py run, keep only the
per-kernel averages of our kernels (the raw CSVs are far too large to copy back).

usage: tools/pmc_pass.py <out.json> [script.py] [args...]      (script defaults to bench.py)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"],
    ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU",
     "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
]


def write_c2_traffic(res, bench_args, path):
    """HBM bytes per input sample of the two kernels of BASELINE config 2 from the FETCH_SIZE / WRITE_SIZE passes (rocprofv3
    reports KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md: on gfx950 it tallies 128-byte requests at 64 bytes), stamped
    with the fingerprint of the kernel sources: bench.py reports `roofline.traffic` only from a stamp that matches its own."""
    sys.path.insert(0, ROOT)
    import bench
    log2 = 22
    meters = "on"
    for i, a in enumerate(bench_args):
        if a == "--log2-samples":
            log2 = int(bench_args[i + 1])
        if a == "--meters":
            meters = bench_args[i + 1]
    samples = 256.0 * (1 << log2)
    kern = {}
    for name, v in res.items():
        if "osfir_kernel<double, 4096, 4, false, false" in name:      # (MIX, PACKED) = (false, false): the fp64-input front, not the wire-format one
            key = "front"
        elif "osfir_kernel<double, 4096, 1" in name and (("true, false" in name.split("4096, 1,")[1][:30]) == (meters == "on")):
            key = "band"        # the METER instantiation when the meters run (template flags: MIX, PACKED, METER, ...)
        else:
            continue
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            kern[key] = {"kernel": name, "fetch_bytes_x2": 2048.0 * v["FETCH_SIZE"], "write_bytes": 1024.0 * v["WRITE_SIZE"],
                         "bytes_per_input_sample": (2048.0 * v["FETCH_SIZE"] + 1024.0 * v["WRITE_SIZE"]) / samples}
            if "SQ_INSTS_VALU" in v:        # wave-level VALU instructions executed per launch: x 64 lanes = lane operations
                kern[key]["valu_wave_insts_per_input_sample"] = v["SQ_INSTS_VALU"] / samples
    j = {"source_sha16": bench.kernel_source_sha16(), "meters": meters, "log2_samples": log2, "channels": 256, "kernels": kern,
         "source": "tools/pmc_pass.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over bench.py " + " ".join(bench_args)}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(j, open(path, "w"), indent=1, sort_keys=True)


def main():
    out = sys.argv[1]
    script = os.path.join(ROOT, "bench.py")
    rest = sys.argv[2:]
    if rest and rest[0].endswith(".py"):
        script, rest = os.path.abspath(rest[0]), rest[1:]
        bench_args = rest
    else:
        bench_args = rest or ["--steps", "2", "--warmup", "1", "--log2-samples", "20", "--no-cpu-baseline", "--no-other-configs", "--no-le24", "--no-host-fed", "--no-live-traffic"]
    res = collections.defaultdict(dict)
    tmp = "/tmp/pmc_pass"
    for gi, grp in enumerate(GROUPS):
        shutil.rmtree(tmp, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--",
                                              sys.executable, script] + bench_args
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "qh::" in row["Kernel_Name"] or "anonymous namespace" in row["Kernel_Name"]:
                    k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if not acc:
            res["_errors"]["group%d" % gi] = (r.stderr or r.stdout)[-400:]
        for k, d in acc.items():
            for n, v in d.items():
                res[k][n] = sum(v) / len(v)
                res[k]["_launches"] = len(v)
    res["_bench_args"] = bench_args
    if os.path.basename(script) == "bench.py":
        write_c2_traffic(res, bench_args, os.path.join(os.path.dirname(os.path.abspath(out)), "c2_traffic.json"))
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
