#!/bin/bash
# same-flag A/B of the headline between abtree_<name>/ and this tree: tools/ab_trees_same.sh <name> [rounds]
name=${1:-prev}; rounds=${2:-3}
root=$(pwd)
show='import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith("{")]
j=json.loads(l[-1]); k=j.get("kernel_ms",{})
print("%-6s value %9.1f  ms_per_step %.3f  front %.3f  band %.3f" % (sys.argv[1], j["value"], j["ms_per_step"], k.get("front_shift_resample",0), k.get("band_nbp",0)))'
flags="--steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-le24 --no-host-fed --no-live-traffic"
for i in $(seq $rounds); do
    (cd $root/abtree_$name && python3 bench.py $flags 2>/dev/null | python3 -c "$show" $name)
    (cd $root && python3 bench.py $flags 2>/dev/null | python3 -c "$show" HEAD)
done
