"""more seeds of tests/test_gpu_quisk_api_fuzz.py::test_random_setter_walk_with_wdsp_in_the_audio_path: api_wdsp_fuzz_sweep.py <first> <last>"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import quisk_amd as qh
import pyoracle as oracle
import test_gpu_quisk_api_fuzz as T
CASES = [(3, 192000, 48000), (3, 96000, 96000), (4, 192000, 48000), (1, 48000, 48000), (5, 192000, 96000), (2, 111111, 48000), (3, 370370, 192000),
         (4, 53333, 48000), (0, 133333, 48000), (7, 185185, 96000), (13, 96000, 48000), (9, 192000, 48000)]
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b + 1):
    mode, fs, play = CASES[seed % len(CASES)]
    try:
        T.test_random_setter_walk_with_wdsp_in_the_audio_path(qh, oracle, seed, mode, fs, play)
    except AssertionError as e:
        bad += 1
        print("seed %d %r: %s" % (seed, (mode, fs, play), str(e)[:int(os.environ.get("CHARS", "400"))].replace("\n", " ")), flush=True)
    except Exception:
        bad += 1
        print("seed %d %r: %s" % (seed, (mode, fs, play), traceback.format_exc()[-500:]), flush=True)
print("%d walks, %d bad" % (b - a + 1, bad))
