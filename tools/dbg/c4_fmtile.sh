#!/bin/bash
# config 4 against the FM loop's tile length (QH_FM_TILE), long and short calls
for tl in 256 512 1024 2048; do
  echo "== QH_FM_TILE=$tl"
  QH_FM_TILE=$tl C4_LENS="${C4_LENS:-256 4096}" bash tools/dbg/c4_len.sh
done
