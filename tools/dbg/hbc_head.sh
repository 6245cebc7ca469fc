#!/bin/bash
# eight half-band stages: where to cut the cascade into launches, and which kernel takes the first stages
for cfg in "QH_HBC_HEAD=8" "QH_HBC_HEAD=3" "QH_HBC_HEAD=4" "QH_HBC_HEAD=5" "QH_HBC_HEAD=3 QH_HBC_NOWAVE=1" "QH_HBC_HEAD=2"; do
  echo -n "$cfg: "
  env $cfg python tools/dbg/hbc_split.py 2>&1 | grep -E "^8 stages|^3 stages|^4 stages" | tr '\n' ' '
  echo
done
