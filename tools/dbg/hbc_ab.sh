#!/bin/bash
# half-band cascade timing for every A/B library variant under quisk_amd/lib/ab
for lib in quisk_amd/lib/ab/libquiskhip_*.so; do
  echo "== $lib"
  QUISKHIP_LIB=$PWD/$lib python tools/dbg/hbc_split.py 2>&1 | grep -E "^3 stages|^8 stages|3 \+ 5"
done
