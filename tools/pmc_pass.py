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


def main():
    out = sys.argv[1]
    script = os.path.join(ROOT, "bench.py")
    rest = sys.argv[2:]
    if rest and rest[0].endswith(".py"):
        script, rest = os.path.abspath(rest[0]), rest[1:]
        bench_args = rest
    else:
        bench_args = rest or ["--steps", "2", "--warmup", "1", "--log2-samples", "20", "--no-cpu-baseline"]
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
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
