#!/bin/bash
# kernel timeline of the last config-4 step of `tools/bench_configs.py 4` (start / duration / queue of every kernel; the graph-replayed
# steps come last): tools/dbg/c4_timeline.sh [out.txt]     -- on the GPU box, from the repo root
out=$(realpath -m "${1:-gpurun_out/c4_timeline.txt}")
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl && mkdir -p /tmp/ktl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -o k -- python3 $root/tools/bench_configs.py 4 > /tmp/ktl/run.log 2>&1
f=$(find /tmp/ktl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee "$out"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts with the front kernel's tile-table launch (nco_tile_kernel); the timed run with enable_timing comes last, so take the
# step before the last three nco_tile launches (the replayed ones)
idx = [i for i, r in enumerate(rows) if "nco_tile_kernel" in r["Kernel_Name"]]
start, end = idx[-4], idx[-3]
t0 = int(rows[start]["Start_Timestamp"])
last = 0
for r in rows[start:end]:
    n = r["Kernel_Name"].replace("void ", "").replace("qh::", "")
    n = n.split("(")[0][:58]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    last = max(last, e)
    print("%-58s q%-3s start %8.1f  end %8.1f  dur %7.1f  grid %s x %s" % (n, r.get("Queue_Id", "?"), s, e, e - s, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
print("step: %.1f us from the first kernel's start to the last kernel's end; next step starts at %.1f" % (last, (int(rows[end]["Start_Timestamp"]) - t0) / 1e3))
PY
