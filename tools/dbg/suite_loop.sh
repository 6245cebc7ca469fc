# the whole GPU suite, N times in a row on one box, the failures of every pass kept: tools/dbg/suite_loop.sh <passes> [pytest args]
cd $GRAFT_REPO_ROOT
N=${1:-4}; shift
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > gpurun_out/suite_pass_$i.log 2>&1
  echo "pass $i: $(tail -1 gpurun_out/suite_pass_$i.log)"
  grep -n "AssertionError\|^FAILED" gpurun_out/suite_pass_$i.log | cut -c1-600 | head -8
done
