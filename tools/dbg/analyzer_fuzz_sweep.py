"""more seeds of tests/test_gpu_analyzer_fuzz.py than the suite carries: analyzer_fuzz_sweep.py <first> <last>"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import quisk_amd as qh
import pyoracle as oracle
import test_gpu_analyzer_fuzz as T
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b + 1):
    try:
        T.test_random_walk_over_the_display_engine(qh, oracle, seed)
    except AssertionError as e:
        bad += 1
        print("seed %d: %s" % (seed, str(e)[:300].replace("\n", " ")), flush=True)
    except Exception:
        bad += 1
        print("seed %d: %s" % (seed, traceback.format_exc()[-400:]), flush=True)
print("%d walks, %d bad" % (b - a + 1, bad))
