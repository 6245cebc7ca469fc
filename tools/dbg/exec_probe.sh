#!/bin/bash
# which step of bench.py makes a process that holds the GPU (under rocprofv3 --pmc: from its first instruction) exec another program?
root=$(pwd); cd /tmp && export TMPDIR=/tmp
n() { cat $root/gpurun_out/.graft_exec_refused 2>/dev/null | wc -l; }
rm -f $root/gpurun_out/.graft_exec_refused
cat > /tmp/a.py <<PY
import torch, torch.distributed
print(torch.cuda.is_available())
PY
cat > /tmp/b.py <<PY
import sys; sys.path.insert(0, "$root")
import torch
print(torch.cuda.is_available())
from quisk_amd import build
build.build()
PY
cat > /tmp/c.py <<PY
import sys; sys.path.insert(0, "$root")
import torch
print(torch.cuda.is_available())
import quisk_amd
quisk_amd.load()
from quisk_amd import synth, shard
PY
for s in a b c; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ep_$s -o p -- python3 /tmp/$s.py > /tmp/ep_$s.log 2>&1
  echo "$s: refused so far $(n)"
done
cat $root/gpurun_out/.graft_exec_refused 2>/dev/null
