#!/usr/bin/env python3
"""Golden vectors for the Rx filter design: runs the REFERENCE's own Python (quisk.py MakeFilterCoef and
GetFilterCenter, pulled out of the source with `ast` because quisk.py imports wx at module level, and the
`Filters` table of filters.py) for the default filter buttons of each mode, and stores (rate, bw, center) ->
(filtI, filtQ) in tests/golden/rxfilter_golden.npz.  Runs only in the build container."""
import ast
import cmath
import math
import os
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_reference_functions():
    src = open(os.path.join(REF, "quisk.py")).read()
    tree = ast.parse(src)
    want = {"MakeFilterCoef", "GetFilterCenter"}
    funcs = [n for cls in ast.walk(tree) if isinstance(cls, ast.ClassDef) for n in cls.body
             if isinstance(n, ast.FunctionDef) and n.name in want]
    mod = ast.Module(body=funcs, type_ignores=[])
    sys.path.insert(0, REF)
    import filters as ref_filters          # pure data

    class Conf:
        cwTone = 600
    ns = {"math": math, "cmath": cmath, "Filters": ref_filters.Filters, "conf": Conf}
    exec(compile(mod, "quisk.py", "exec"), ns)
    return ns["MakeFilterCoef"], ns["GetFilterCenter"]


def main():
    make, center_of = load_reference_functions()
    cases = []
    # (mode, filter rate from get_filter_rate quisk.c:2787-2859, bandwidths of the default buttons)
    for mode, rate, bws in (("USB", 12000, (1500, 2000, 2500, 2700, 2800, 3000)), ("LSB", 12000, (2700,)),
                            ("CWU", 6000, (200, 400, 1000)), ("CWL", 6000, (500,)),
                            ("AM", 24000, (4000, 6000, 9000)), ("FM", 48000, (8000, 12000, 16000))):
        for bw in bws:
            c = center_of(None, mode, bw)
            fI, fQ = make(None, rate, None, bw, c)
            cases.append((mode, rate, bw, c, np.array(fI), np.array(fQ)))
    out = {}
    for i, (mode, rate, bw, c, fI, fQ) in enumerate(cases):
        out["case%d_meta" % i] = np.array([rate, bw, c], dtype=np.float64)
        out["case%d_mode" % i] = np.array(mode)
        out["case%d_I" % i] = fI
        out["case%d_Q" % i] = fQ
    out["ncases"] = np.array(len(cases))
    path = os.path.join(ROOT, "tests", "golden", "rxfilter_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(cases), "cases;", [(m, r, b, c, len(i)) for m, r, b, c, i, q in cases])


if __name__ == "__main__":
    main()
