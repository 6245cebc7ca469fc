#!/bin/bash
# kernel timeline of the last step of the whole-function Quisk-native leg (USB): start / end per kernel and queue
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl && mkdir -p /tmp/ktl
QH_QUISK_MODES=USB rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -o k -- python3 $GRAFT_REPO_ROOT/tools/dbg/qps_one.py > /tmp/ktl/run.log 2>&1
tail -3 /tmp/ktl/run.log
f=$(find /tmp/ktl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "q_agc_pair" in r["Kernel_Name"]]
# the last step = the last 4 AGC launches; back up to the first filter kernel of that step
start = idx[-4]
while start > 0 and int(rows[start - 1]["End_Timestamp"]) > int(rows[idx[-5]]["End_Timestamp"]) and start - 1 > idx[-5]:
    start -= 1
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    n = r["Kernel_Name"].replace("void ", "").replace("qh::", "").replace("(anonymous namespace)::", "")[:44]
    print("%-44s q%-3s start %8.1f  end %8.1f  dur %7.1f" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
