#!/bin/bash
# PMC passes over the three-stage half-band kernel (separate passes, no tracing domains besides the kernel trace)
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/hpmc && mkdir -p /tmp/hpmc
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/hpmc -o p -- python3 $GRAFT_REPO_ROOT/tools/dbg/hbc_only.py 3 > /tmp/hpmc/log.txt 2>&1
  f=$(find /tmp/hpmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "hb45_wave" not in r["Kernel_Name"] and "hb45_cascade" not in r["Kernel_Name"]: continue
    k = r["Counter_Name"]; agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()): print("%-26s %16.0f per launch (%d rows)" % (k, v / max(n, 1), n))
PY
done
