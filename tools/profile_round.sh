#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# The round's evidence set from ONE library on ONE box, written to gpurun_out/<tag>_*: the driver's line, rocprofv3 kernel stats of the
# headline run and of every leg of `other_configs`, the PMC passes of the headline (with the traffic / VALU stamp bench.py reads) and
# of configs 3 and 5 (the half-band cascade).  Copy what is to be judged into profiles/.
tag=${1:-r06}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 300 $out/${tag}_bench.err
tools/kstats.sh $out/${tag}_kernel_stats.csv python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-le24 --no-host-fed --no-live-traffic > /dev/null
for leg in 3 4 5 2agc quisk; do
    tools/kstats.sh $out/${tag}_${leg}_kernel_stats.csv python3 $root/tools/bench_configs.py $leg > $out/${tag}_${leg}.log 2>&1
    tail -1 $out/${tag}_${leg}.log | cut -c1-400
done
python3 tools/pmc_pass.py $out/${tag}_pmc.json > /dev/null 2> $out/${tag}_pmc.err
python3 tools/pmc_pass.py $out/${tag}_c3_pmc.json $root/tools/bench_configs.py 3 > /dev/null 2> $out/${tag}_c3_pmc.err
python3 tools/pmc_pass.py $out/${tag}_c5_pmc.json $root/tools/dbg/hbc_only.py 8 > /dev/null 2> $out/${tag}_c5_pmc.err
ls -la $out | grep ${tag}_
