"""The EMNR tables compiled into the library (quisk_amd/build.py: _emnr_tables_source -> qh_emnr_tables_gen.cpp) are WDSP's own compiled-in
fall-backs: wdsp/calculus.c (GG, GGS) and wdsp/zetahat.c, which emnr.c:207-225, 322-326 takes when `calculus` / `zetaHat.bin` are not in the
working directory."""
import os
import re
import struct

import numpy as np
import pytest


def _generated(tmp_path):
    from quisk_amd import build
    src = open(build._emnr_tables_source(str(tmp_path / "gen.cpp"))).read()
    out = {}
    for name, kind in (("kEmnrDefaultGG", "Q"), ("kEmnrDefaultGGS", "Q"), ("kEmnrDefaultZeta", "Q"), ("kEmnrDefaultRange", "Q"), ("kEmnrDefaultValid", "i")):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", src, re.S).group(1)
        toks = [t.strip() for t in body.replace("\n", " ").split(",") if t.strip()]
        if kind == "Q":
            out[name] = np.frombuffer(struct.pack("<%dQ" % len(toks), *[int(t[:-3], 16) for t in toks]), dtype="<f8")
        else:
            out[name] = np.array([int(t) for t in toks], dtype=np.int32)
    return out


def test_generated_tables_are_the_shipped_data(tmp_path):
    g = _generated(tmp_path)
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quisk_amd", "data", "wdsp_emnr_tables.npz"))
    assert np.array_equal(g["kEmnrDefaultGG"], z["GG"]) and np.array_equal(g["kEmnrDefaultGGS"], z["GGS"])
    assert np.array_equal(g["kEmnrDefaultValid"], z["zeta_valid"]) and np.array_equal(g["kEmnrDefaultRange"], z["zeta_range"])
    ok = z["zeta_valid"] > 0                                    # the cells getZeta reads (emnr.c:878-882)
    assert ok.sum() > 1000 and np.array_equal(g["kEmnrDefaultZeta"][ok], z["zeta_hat"][ok])
    none = z["zeta_hat"] == -1e300                              # the file's "no value" mark: 0 in the compiled-in table
    assert np.array_equal(g["kEmnrDefaultZeta"][~none], z["zeta_hat"][~none]) and not np.any(g["kEmnrDefaultZeta"][none]) and not np.any(ok & none)


@pytest.mark.skipif(not os.path.exists("/root/reference/wdsp/zetahat.c"), reason="the reference tree is not mounted here")
def test_generated_tables_are_the_references_compiled_in_ones(tmp_path):
    g = _generated(tmp_path)
    cal = open("/root/reference/wdsp/calculus.c").read()
    nums = np.array([float(x) for x in re.findall(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?", cal)])
    assert nums.size == 2 * 241 * 241
    assert np.array_equal(nums[:241 * 241], g["kEmnrDefaultGG"]) and np.array_equal(nums[241 * 241:], g["kEmnrDefaultGGS"])
    zh = open("/root/reference/wdsp/zetahat.c").read()
    data = re.search(r"zetaHatDefaultData\s*\[[^\]]*\]\s*=\s*\{(.*?)\};", zh, re.S).group(1)
    valid = re.search(r"zetaHatDefaultValid\s*\[[^\]]*\]\s*=\s*\{(.*?)\};", zh, re.S).group(1)
    assert np.array_equal(np.array([float(t) for t in data.replace("\n", " ").split(",")]), g["kEmnrDefaultZeta"])
    assert np.array_equal(np.array([int(t) for t in valid.replace("\n", " ").split(",")]), g["kEmnrDefaultValid"])
    rng = [float(re.search(r"zetaHatDefault%s\s*=\s*([^;]+);" % k, zh).group(1)) for k in ("Gmin", "Gmax", "Ximin", "Ximax")]
    assert rng == list(g["kEmnrDefaultRange"])
