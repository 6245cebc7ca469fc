"""more seeds of the seeded setter walks of tests/test_gpu_rxa_fuzz.py than the suite carries: fuzz_sweep.py <first> <last> [wide|plain|replay]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import quisk_amd as qh
import pyoracle as oracle
import test_gpu_rxa_fuzz as F
a, b = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "wide"
bad = 0
for seed in range(a, b + 1):
    try:
        F._walk(qh, oracle, seed, kind == "replay", kind == "wide")
    except AssertionError as e:
        bad += 1
        print("seed %d: %s" % (seed, str(e)[:600]), flush=True)
    except Exception:
        bad += 1
        print("seed %d: %s" % (seed, traceback.format_exc()[-600:]), flush=True)
print("%d walks (%s), %d bad" % (b - a + 1, kind, bad))
