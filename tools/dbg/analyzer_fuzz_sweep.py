"""more seeds of tests/test_gpu_analyzer_fuzz.py than the suite carries: analyzer_fuzz_sweep.py <first> <last> [bank]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import quisk_amd as qh
import pyoracle as oracle
import test_gpu_analyzer_fuzz as T
a, b = int(sys.argv[1]), int(sys.argv[2])
fn = T.test_random_walk_over_a_bank_of_displays if len(sys.argv) > 3 and sys.argv[3] == "bank" else T.test_random_walk_over_the_display_engine
bad = 0
for seed in range(a, b + 1):
    try:
        fn(qh, oracle, seed)
    except AssertionError as e:
        bad += 1
        print("seed %d: %s" % (seed, str(e)[:int(os.environ.get('CHARS', '300'))].replace("\n", " ")), flush=True)
    except Exception:
        bad += 1
        print("seed %d: %s" % (seed, traceback.format_exc()[-400:]), flush=True)
print("%d walks, %d bad" % (b - a + 1, bad))
