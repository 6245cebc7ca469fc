#!/bin/bash
# The kernels of one steady and one keyed step of the config-2-with-AGC leg in order: start and duration (rocprofv3 kernel trace of
# tools/dbg/agc_leg.py).  Run on the GPU box: gpurun -- bash tools/dbg/agc_timeline.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/kst && mkdir -p /tmp/kst
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kst -o k -- python3 $GRAFT_REPO_ROOT/tools/dbg/agc_leg.py > /tmp/kst/run.log 2>&1
grep warm /tmp/kst/run.log
f=$(find /tmp/kst -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if r['Kernel_Name'].startswith('qh::agc') or 'osfir_kernel' in r['Kernel_Name']]
steps, cur = [], []
for r in sel:
    if 'osfir_kernel<double, 4096, 4' in r['Kernel_Name'] and cur: steps.append(cur); cur = []
    cur.append(r)
steps.append(cur)
for si, what in ((5, "steady"), (11, "keyed")):
    if si >= len(steps): continue
    print("step", si, what)
    t0 = int(steps[si][0]['Start_Timestamp'])
    for r in steps[si]:
        print("   %-48s start %8.3f  dur %7.3f ms" % (r['Kernel_Name'][:48], (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
P
