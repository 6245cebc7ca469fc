#!/bin/bash
for nb in ${C4_LENS:-256 1024 4096}; do
  echo -n "nblk $nb: "
  QH_C4_NBLK=$nb python tools/bench_configs.py 4 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('%.3f ms  %.1f Gsamp/s   graph %.3f ms %.1f Gsamp/s   rerun %d  front %.3f band %.3f rest %.3f' % (r['ms'], r['Msamp_per_s']/1e3, r['ms_graph_replay'], r['Msamp_per_s_graph_replay']/1e3, r['pll_tiles_rerun'], r['front_ms'], r['band_ms'], r['rest_ms']))"
done
