#!/bin/bash
# first three half-band stages (one wavefront per segment) against the segment length in 512-sample steps
for st in 16 17 31 32 33 47 63 64 65 97 129; do
  echo -n "seg steps $st: "
  QH_HBC_SEG_STEPS=$st python tools/dbg/hbc_split.py 2>&1 | grep -E "^3 stages|^2 stages|^8 stages" | tr '\n' ' '
  echo
done
