#!/bin/bash
# usage: tools/kstats.sh <out.csv> <program> [args...]   -- rocprofv3 kernel-trace stats of one command, top kernels
# (run on the GPU box; the program goes straight after `--`, never through env/bash -c)
out=$(realpath -m "$1"); shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst && mkdir -p /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- "$@" > /tmp/kst/run.log 2>&1
tail -3 /tmp/kst/run.log
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -40 "$f" | cut -c1-700 | tee "$out"; else echo "no kernel_stats.csv"; fi
