#!/bin/bash
# Same-box A/B of the headline (config 2) between this tree and an older tree checked out (with its library built) under abtree_<name>/:
#   tools/ab_trees.sh r02 [rounds]      -- on the GPU box, from the repo root; prints value / ms_per_step / front / band per run
name=${1:-r02}; rounds=${2:-3}
root=$(pwd)
show='import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith("{")]
j=json.loads(l[-1]); k=j.get("kernel_ms",{})
print("%-6s value %9.1f  ms_per_step %.3f  front %.3f  band %.3f  meters_off %s" % (sys.argv[1], j["value"], j["ms_per_step"], k.get("front_shift_resample",0), k.get("band_nbp",0), j.get("value_meters_off")))'
for i in $(seq $rounds); do
    (cd $root/abtree_$name && python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "$show" $name)
    (cd $root && python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-le24 --no-host-fed --no-live-traffic 2>/dev/null | python3 -c "$show" HEAD)
done
