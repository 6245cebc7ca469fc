cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_band_tile_8192.py -x -q -m gpu 2>&1 | tail -8
for t in 0 8192; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-le24 --no-other-configs --band-tile $t > gpurun_out/r05_band_$t.json 2> gpurun_out/r05_band_$t.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r05_band_$t.json"))
print("band-tile $t:", d["value"], d["ms_per_step"], d.get("kernel_ms"))
PY
done
QH_BAND8_FORM=2g timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-le24 --no-other-configs --band-tile 8192 > gpurun_out/r05_band_2g.json 2> gpurun_out/r05_band_2g.err
python -c "
import json
d=json.load(open('gpurun_out/r05_band_2g.json'))
print('band-tile 8192 two groups:', d['value'], d['ms_per_step'], d.get('kernel_ms'))"
