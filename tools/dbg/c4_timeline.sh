#!/bin/bash
# kernel timeline of one config-4 step (start / end per kernel, queue), last call of the run
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl && mkdir -p /tmp/ktl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -o k -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py 4 > /tmp/ktl/run.log 2>&1
f=$(find /tmp/ktl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last front kernel launches (D=4) and print the window after the second to last one
idx = [i for i, r in enumerate(rows) if "4096, 4," in r["Kernel_Name"]]
# split mode launches two fronts per step: take the last 4 fronts -> start of the second to last step
start = idx[-4]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:start + 60]:
    n = r["Kernel_Name"].replace("void ", "").replace("qh::", "")[:46]
    print("%-46s q%-3s start %8.1f  dur %7.1f  grid %s" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
PY
