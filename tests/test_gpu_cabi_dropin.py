"""The drop-in boundary exercised from C: tests/cabi/dropin.c compiled with gcc -std=c99 against include/quiskhip.h, linked with
-lquiskhip, run on the GPU.  Inputs come from seeds, expectations from tests/golden/filter_golden.npz (the reference's own filter.c,
tests/golden/make_filter_golden.py) and from the CPU restatement; the program compares and reports per case."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIP32 = 2147483647.0

pytestmark = pytest.mark.gpu


def _stream(seed, n, complex_=True):         # tests/golden/make_filter_golden.py: stream()
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    if complex_:
        x = x + 1j * rng.standard_normal(n)
    return x


def test_dropin_c_program(qh, oracle, tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    lib = qh.load()
    g = np.load(os.path.join(ROOT, "tests", "golden", "filter_golden.npz"))
    splits = g["splits"].astype(np.int64)
    n = int(splits.sum())
    xc, xr = _stream(11, n), _stream(12, 64, False)
    t98 = np.ascontiguousarray(g["taps98"])
    d = tmp_path / "case"
    d.mkdir()

    def put(name, a, dt):
        np.ascontiguousarray(a, dtype=dt).tofile(str(d / name))
    put("xc.bin", xc, np.complex128); put("xr.bin", xr, np.float64); put("taps98.bin", t98, np.float64); put("splits.bin", splits, np.int64)
    put("expect_hb45.bin", g["cDecim2HB45"], np.complex128)
    put("expect_dec98.bin", g["cDecimate_98_d2"], np.complex128)
    # Quisk's 192 ksps plan on one stream, blocks of 1024 (quisk.c:1769-1833): the restatement, itself pinned by the golden vectors
    hb, f = oracle.OracleHB45(), oracle.OracleFir(t98)
    plan = [f.cDecimate(hb.cDecim2(xc[p:p + 1024]), 2) for p in range(0, n, 1024)]
    put("expect_plan192.bin", np.concatenate(plan), np.complex128)
    fc = oracle.OracleFir(t98)
    fc.tune(0.0625, 1)
    put("expect_dcout.bin", fc.cCDecimate(xr + 0j, 1), np.complex128)           # complex taps on a real stream (filter.c:83-104)
    # wdspFexchange0: samples at Quisk's scale in, 1 / CLIP32 into fexchange0, CLIP32 back (quisk_wdsp.c:48,65)
    nb = 12
    xw = _stream(13, nb * 512) * 1e8
    o = oracle.WdspChannel(512, 256, 48000, 48000, 48000)
    o.SetRXAShiftRun(1); o.SetRXAShiftFreq(-9000.0); o.RXANBPSetRun(1); o.SetRXAMode(1); o.RXASetPassband(300.0, 3000.0)
    o.SetRXAAGCMode(0); o.SetRXAAGCFixed(0.0)
    ref, errs = o.fexchange0(xw / CLIP32)
    assert errs == 0
    put("wdsp_in.bin", xw, np.complex128); put("expect_wdsp.bin", ref * CLIP32, np.complex128)

    exe = tmp_path / "dropin"
    libdir = os.path.dirname(lib._name)
    subprocess.run(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cabi", "dropin.c"), "-o", str(exe), "-L", libdir, "-lquiskhip", "-lm", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    print(r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert r.stdout.count(" ok  ") == 5
