"""more seeds of any seeded walk that takes (qh, oracle, seed): walk_sweep.py <test module> <test function> <first> <last>   (env CHARS: how
much of a failure's message to print)"""
import importlib, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch          # before libquiskhip: one HIP runtime per process
import quisk_amd as qh
import pyoracle as oracle
fn = getattr(importlib.import_module(sys.argv[1]), sys.argv[2])
a, b = int(sys.argv[3]), int(sys.argv[4])
bad = 0
for seed in range(a, b + 1):
    try:
        fn(qh, oracle, seed)
    except AssertionError as e:
        bad += 1
        print("seed %d: %s" % (seed, str(e)[:int(os.environ.get("CHARS", "300"))].replace("\n", " ")), flush=True)
    except Exception:
        bad += 1
        print("seed %d: %s" % (seed, traceback.format_exc()[-500:]), flush=True)
print("%d walks, %d bad" % (b - a + 1, bad))
