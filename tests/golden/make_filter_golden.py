#!/usr/bin/env python3
"""Generates tests/golden/filter_golden.npz from the REFERENCE's own filter.c (oracle/_ref, built by
`make -C oracle ref` from /root/reference/filter.c where it lies).  Runs only in the build container;
the .npz (inputs are regenerated from seeds; only outputs and the small coefficient tables used as
inputs are stored) travels to the GPU box.

Cases cover every filter.c primitive on the hot path including
ragged block sizes, decimation phase carried across calls, count = 0 and count < decim.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po      # noqa: E402

SPLITS = [0, 1, 7, 333, 2, 64, 1000, 5, 0, 588]     # 2000 samples in ragged calls


def stream(seed, n, complex_=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    if complex_:
        x = x + 1j * rng.standard_normal(n)
    return x


def run_split(fn, x):
    out, pos = [], 0
    for k in SPLITS:
        out.append(fn(x[pos:pos + k]))
        pos += k
    return np.concatenate(out)


def main():
    po.build(ref=True)
    g = {}
    t98 = po.ref_table("quiskFilt48dec24Coefs", 98)         # filters.h:530 (Quisk's final /2 stage)
    t147 = po.ref_table("quiskFilt144D3Coefs", 147)         # filters.h:850
    t245 = po.ref_table("quiskFilt240D5CoefsSharp", 245)    # filters.h:478
    t36 = po.ref_table("quiskAudio24p6Coefs", 36)
    g["taps98"], g["taps147"], g["taps245"], g["taps36"] = t98, t147, t245, t36
    n = sum(SPLITS)
    xc, xr = stream(11, n), stream(12, n, False)
    for name, taps, d in (("cDecimate_98_d2", t98, 2), ("cDecimate_147_d3", t147, 3), ("cDecimate_245_d5", t245, 5),
                          ("cDecimate_98_d1", t98, 1)):
        f = po.RefFir(taps)
        g[name] = run_split(lambda v: f.cDecimate(v, d), xc)
    f = po.RefFir(t245)
    f.tune(0.0625, 1)
    g["cCDecimate_245_d5_usb"] = run_split(lambda v: f.cCDecimate(v, 5), xc)
    f = po.RefFir(t98)
    f.tune(-0.11, 0)
    g["cCDecimate_98_d2_lsb"] = run_split(lambda v: f.cCDecimate(v, 2), xc)
    f = po.RefFir(t147, is_complex=False)
    g["dDecimate_147_d3"] = run_split(lambda v: f.dDecimate(v, 3), xr)
    f = po.RefFir(t36, is_complex=False)
    g["dFilter_36"] = run_split(lambda v: f.dFilter(v), xr)
    f = po.RefFir(t36)
    g["cInterpolate_36_x2"] = run_split(lambda v: f.cInterpolate(v, 2), xc)
    f = po.RefFir(t36, is_complex=False)
    g["dInterpolate_36_x3"] = run_split(lambda v: f.dInterpolate(v, 3), xr)
    f = po.RefFir(t98)
    g["cInterpDecim_98_6_5"] = run_split(lambda v: f.cInterpDecim(v, 2, 3), xc)
    h = po.RefHB45()
    g["cDecim2HB45"] = run_split(h.cDecim2, xc)
    h = po.RefHB45()
    g["cInterp2HB45"] = run_split(h.cInterp2, xc)
    h = po.RefHB45()
    g["dInterp2HB45"] = run_split(h.dInterp2, xr)
    # cascade used by BASELINE config 5: 8 x HB45 then 245-tap /5 (a short stream)
    x = stream(13, 256 * 5 * 40)
    hs = [po.RefHB45() for _ in range(8)]
    f = po.RefFir(t245)
    y = x
    for hb in hs:
        y = hb.cDecim2(y)
    g["cascade_8hb45_d5"] = f.cDecimate(y, 5)
    g["splits"] = np.array(SPLITS)
    out = os.path.join(ROOT, "tests", "golden", "filter_golden.npz")
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out), "bytes;", len(g), "arrays")


if __name__ == "__main__":
    main()
