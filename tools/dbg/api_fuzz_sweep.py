"""more seeds of tests/test_gpu_quisk_api_fuzz.py (the one-receiver API) than the suite carries: bank_fuzz_sweep.py <first> <last>"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import quisk_amd as qh
import pyoracle as oracle
import test_gpu_quisk_api_fuzz as T
combos = [(3, 192000, 48000), (3, 111111, 96000), (4, 96000, 48000), (5, 192000, 48000), (3, 48000, 48000), (1, 133333, 48000), (4, 185185, 96000),
          (5, 96000, 192000), (3, 192000, 192000), (3, 370370, 48000), (1, 48000, 96000), (5, 53333, 48000), (4, 740740, 48000), (3, 96000, 96000),
          (0, 96000, 48000), (2, 192000, 48000), (7, 192000, 96000), (8, 111111, 48000), (9, 192000, 48000), (13, 96000, 48000), (10, 48000, 48000)]
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b + 1):
    mode, fs, play = combos[seed % len(combos)]
    print("walk %d (mode %d, %d -> %d)" % (seed, mode, fs, play), file=sys.stderr, flush=True)
    try:
        T.test_random_setter_walk_over_the_one_receiver_api(qh, oracle, seed, mode, fs, play)
    except AssertionError as e:
        bad += 1
        print("seed %d (mode %d, %d -> %d): %s" % (seed, mode, fs, play, str(e)[:500]), flush=True)
    except Exception:
        bad += 1
        print("seed %d (mode %d, %d -> %d): %s" % (seed, mode, fs, play, traceback.format_exc()[-500:]), flush=True)
print("%d walks, %d bad" % (b - a + 1, bad))
