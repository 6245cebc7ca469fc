"""Every setter of one RXA walk (all channels), in order: tools/dbg/walk_log.py <seed> [rxa|rxa_long|rxa_notch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import quisk_amd as qh
import pyoracle as oracle
import test_gpu_rxa_fuzz as T
seed = int(sys.argv[1]); fam = sys.argv[2] if len(sys.argv) > 2 else "rxa_long"
for name in ("_apply", "_apply2"):
    f = getattr(T, name)
    def wrap(*a, _f=f, _n=name, **k):
        r = _f(*a, **k)
        tg = a[1]
        print("   %s -> channel %s: %r" % (_n, tg[0][1], r), flush=True)
        return r
    setattr(T, name, wrap)
fn = {"rxa": T.test_random_setter_walk, "rxa_long": T.test_random_setter_walk_with_long_filters_agc_windows_and_long_calls,
      "rxa_notch": T.test_random_setter_walk_with_the_notch_database_the_lms_sizes_and_fm}[fam]
try:
    fn(qh, oracle, seed)
    print("walk ok")
except AssertionError as e:
    print(str(e)[:1500])
