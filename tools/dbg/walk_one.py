"""One walk of a family of walk_sweep_all.py with its whole traceback: tools/dbg/walk_one.py <family> <seed>"""
import os, sys, traceback, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fam, seed = sys.argv[1], int(sys.argv[2])
sys.argv = ["walk_sweep_all.py", "0", "-1"]
ns = runpy.run_path(os.path.join(ROOT, "tools", "dbg", "walk_sweep_all.py"))
try:
    ns["FAM"][fam](seed)
    print("passed")
except BaseException:
    print(traceback.format_exc()[-int(os.environ.get("CHARS", "2500")):])
