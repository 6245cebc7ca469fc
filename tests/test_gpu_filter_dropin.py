"""filter.h drop-in exports: same names, same structs, state kept in the caller's struct in the reference's
format -- so a stream can hop between this library (GPU) and the reference's own filter.c (oracle/_ref, CPU)
from block to block.  -m gpu."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu
c_double_p = C.POINTER(C.c_double)


def stream(seed, n):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(n) + 1j * rng.standard_normal(n)


def call(fn, x, st, *args):
    buf = np.ascontiguousarray(x, dtype=np.complex128).copy()
    fn.restype = C.c_int
    n = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size), C.byref(st), *[C.c_int(a) for a in args])
    return buf[:n].copy()


@pytest.mark.parametrize("ntaps,decim", [(98, 2), (147, 3), (245, 5), (31, 1)])
def test_cdecimate_struct_state_roundtrip(qh, oracle, ntaps, decim):
    lib = qh.load()
    rng = np.random.default_rng(ntaps)
    taps = np.ascontiguousarray(rng.standard_normal(ntaps))
    x = stream(5, 3000)
    cuts = [0, 700, 701, 1500, 1507, 3000]
    st = oracle.RefCFilter()
    lib.quisk_filt_cInit(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(ntaps))
    ours = np.concatenate([call(lib.quisk_cDecimate, x[a:b], st, decim) for a, b in zip(cuts, cuts[1:])])
    want = oracle.OracleFir(taps).cDecimate(x, decim)
    assert ours.size == want.size and rel_rms(ours, want) < 1e-12
    ref = oracle.ref_filter_lib()
    if ref is None:
        pytest.skip("oracle/_ref not present")
    # alternate GPU / reference-CPU on ONE struct: the state format must be the reference's
    st2 = oracle.RefCFilter()
    ref.quisk_filt_cInit(C.byref(st2), taps.ctypes.data_as(c_double_p), C.c_int(ntaps))
    parts = []
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        fn = lib.quisk_cDecimate if i % 2 == 0 else ref.quisk_cDecimate
        parts.append(call(fn, x[a:b], st2, decim))
    mixed = np.concatenate(parts)
    assert mixed.size == want.size and rel_rms(mixed, want) < 1e-12


def test_ccdecimate_and_tune(qh, oracle):
    lib = qh.load()
    taps = np.ascontiguousarray(np.random.default_rng(1).standard_normal(245))
    x = stream(6, 4000)
    st = oracle.RefCFilter()
    lib.quisk_filt_cInit(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(245))
    lib.quisk_filt_tune.argtypes = [C.c_void_p, C.c_double, C.c_int]
    lib.quisk_filt_tune(C.byref(st), 0.0625, 1)
    ours = np.concatenate([call(lib.quisk_cCDecimate, x[:1999], st, 5), call(lib.quisk_cCDecimate, x[1999:], st, 5)])
    f = oracle.OracleFir(taps)
    f.tune(0.0625, 1)
    want = f.cCDecimate(x, 5)
    assert ours.size == want.size and rel_rms(ours, want) < 1e-12


def test_hb45_struct_state_roundtrip(qh, oracle):
    lib = qh.load()
    x = stream(7, 5001)
    cuts = [0, 1, 1000, 1003, 4000, 5001]
    st = oracle.RefCHB45()
    ours = np.concatenate([call(lib.quisk_cDecim2HB45, x[a:b], st) for a, b in zip(cuts, cuts[1:])])
    want = oracle.OracleHB45().cDecim2(x)
    assert ours.size == want.size and rel_rms(ours, want) < 1e-12
    ref = oracle.ref_filter_lib()
    if ref is None:
        pytest.skip("oracle/_ref not present")
    st2 = oracle.RefCHB45()
    parts = []
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        fn = ref.quisk_cDecim2HB45 if i % 2 == 0 else lib.quisk_cDecim2HB45
        parts.append(call(fn, x[a:b], st2))
    mixed = np.concatenate(parts)
    assert mixed.size == want.size and rel_rms(mixed, want) < 1e-12


def call_real(fn, x, st, *args, grow=1):
    buf = np.zeros(max(len(x) * grow, 1) + 8)
    buf[:len(x)] = x
    fn.restype = C.c_int
    n = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(len(x)), C.byref(st), *[C.c_int(a) for a in args])
    return buf[:n].copy()


def call_grow(fn, x, st, *args, grow=1):
    buf = np.zeros(max(len(x) * grow, 1) + 8, dtype=np.complex128)
    buf[:len(x)] = x
    fn.restype = C.c_int
    n = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(len(x)), C.byref(st), *[C.c_int(a) for a in args])
    return buf[:n].copy()


CUTS = [0, 700, 701, 1500, 1507, 3000]


def alternate(fns, run, st):
    return np.concatenate([run(fns[i % 2], a, b, st) for i, (a, b) in enumerate(zip(CUTS, CUTS[1:]))])


@pytest.mark.parametrize("interp,decim", [(2, 1), (3, 1), (6, 5), (4, 5), (2, 3)])
def test_cinterpolate_cinterpdecim_struct_state(qh, oracle, interp, decim):
    lib, ref = qh.load(), oracle.ref_filter_lib()
    taps = np.ascontiguousarray(np.random.default_rng(interp + decim).standard_normal(120))
    x = stream(8, 3000)
    f = oracle.OracleFir(taps)
    want = f.cInterpolate(x, interp) if decim == 1 else f.cInterpDecim(x, interp, decim)
    name, args = ("quisk_cInterpolate", (interp,)) if decim == 1 else ("quisk_cInterpDecim", (interp, decim))
    st = oracle.RefCFilter()
    lib.quisk_filt_cInit(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(120))
    ours = np.concatenate([call_grow(getattr(lib, name), x[a:b], st, *args, grow=interp) for a, b in zip(CUTS, CUTS[1:])])
    assert ours.size == want.size and rel_rms(ours, want) < 1e-12
    if ref is None:
        pytest.skip("oracle/_ref not present")
    st2 = oracle.RefCFilter()
    ref.quisk_filt_cInit(C.byref(st2), taps.ctypes.data_as(c_double_p), C.c_int(120))
    mixed = alternate([getattr(lib, name), getattr(ref, name)], lambda fn, a, b, s: call_grow(fn, x[a:b], s, *args, grow=interp), st2)
    assert mixed.size == want.size and rel_rms(mixed, want) < 1e-12


def test_real_primitives_struct_state(qh, oracle):
    lib, ref = qh.load(), oracle.ref_filter_lib()
    rng = np.random.default_rng(4)
    taps = np.ascontiguousarray(rng.standard_normal(147))
    x = rng.standard_normal(3000)
    for name, args, grow, want in (
            ("quisk_dDecimate", (3,), 1, oracle.OracleFir(taps, is_complex=False).dDecimate(x, 3)),
            ("quisk_dFilter", (), 1, oracle.OracleFir(taps, is_complex=False).dFilter(x)),
            ("quisk_dInterpolate", (3,), 3, oracle.OracleFir(taps, is_complex=False).dInterpolate(x, 3))):
        st = oracle.RefCFilter()
        lib.quisk_filt_dInit(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(147))
        ours = np.concatenate([call_real(getattr(lib, name), x[a:b], st, *args, grow=grow) for a, b in zip(CUTS, CUTS[1:])])
        assert ours.size == want.size and rel_rms(ours, want) < 1e-12, name
        if ref is not None:
            st2 = oracle.RefCFilter()
            ref.quisk_filt_dInit(C.byref(st2), taps.ctypes.data_as(c_double_p), C.c_int(147))
            mixed = alternate([getattr(ref, name), getattr(lib, name)], lambda fn, a, b, s: call_real(fn, x[a:b], s, *args, grow=grow), st2)
            assert mixed.size == want.size and rel_rms(mixed, want) < 1e-12, name


def test_per_sample_outputs(qh, oracle):
    """quisk_dD_out and quisk_dC_out: one sample per call (each is one GPU launch -- link compatibility only)."""
    lib = qh.load()
    rng = np.random.default_rng(9)
    taps = np.ascontiguousarray(rng.standard_normal(31))
    x = rng.standard_normal(40)
    st = oracle.RefCFilter()
    lib.quisk_filt_dInit(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(31))
    lib.quisk_dD_out.restype = C.c_double
    got = np.array([lib.quisk_dD_out(C.c_double(v), C.byref(st)) for v in x])
    want = oracle.OracleFir(taps, is_complex=False).dFilter(x)
    assert rel_rms(got, want) < 1e-12
    st = oracle.RefCFilter()
    lib.quisk_filt_dInit(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(31))
    lib.quisk_filt_tune.argtypes = [C.c_void_p, C.c_double, C.c_int]
    lib.quisk_filt_tune(C.byref(st), 0.07, 1)
    out = (C.c_double * 2)()
    got = []
    for v in x:
        lib.qh_quisk_dC_out(C.c_double(v), C.byref(st), out)
        got.append(out[0] + 1j * out[1])
    f = oracle.OracleFir(taps)
    f.tune(0.07, 1)
    want = f.cCDecimate(x + 0j, 1)                      # complex taps on a real stream
    assert rel_rms(np.array(got), want) < 1e-12


def test_dC_out_under_the_reference_name_from_c(qh, oracle, tmp_path):
    """`complex double quisk_dC_out(double, struct quisk_dFilter *)` (filter.c:83) as microphone.c:469 calls it: a C translation
    unit compiled against include/quiskhip.h and linked against the library, the value returned in registers."""
    import os
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    lib = qh.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "dc.c"
    src.write_text('#include <complex.h>\n#include "quiskhip.h"\n'
                   'void run(const double *x, int n, struct quisk_cFilter *f, double *out)\n'
                   '{ int i; for (i = 0; i < n; i++) { complex double v = quisk_dC_out(x[i], f); out[2 * i] = creal(v); out[2 * i + 1] = cimag(v); } }\n')
    so = tmp_path / "libdc.so"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-shared", "-fPIC", "-I", os.path.join(root, "include"), str(src), "-o", str(so),
                    "-L", os.path.dirname(lib._name), "-lquiskhip", "-Wl,-rpath," + os.path.dirname(lib._name)], check=True)
    shim = C.CDLL(str(so))
    rng = np.random.default_rng(19)
    taps = np.ascontiguousarray(rng.standard_normal(45))
    x = np.ascontiguousarray(rng.standard_normal(64))
    st = oracle.RefCFilter()
    lib.quisk_filt_dInit(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(45))
    lib.quisk_filt_tune.argtypes = [C.c_void_p, C.c_double, C.c_int]
    lib.quisk_filt_tune(C.byref(st), -0.11, 0)
    out = np.zeros(2 * x.size)
    shim.run(x.ctypes.data_as(c_double_p), C.c_int(x.size), C.byref(st), out.ctypes.data_as(c_double_p))
    f = oracle.OracleFir(taps)
    f.tune(-0.11, 0)
    want = f.cCDecimate(x + 0j, 1)
    assert rel_rms(out[0::2] + 1j * out[1::2], want) < 1e-12


def test_interp2hb45_struct_state(qh, oracle):
    lib, ref = qh.load(), oracle.ref_filter_lib()
    x = stream(10, 3000)
    want = oracle.OracleHB45().cInterp2(x)
    st = oracle.RefCHB45()
    ours = np.concatenate([call_grow(lib.quisk_cInterp2HB45, x[a:b], st, grow=2) for a, b in zip(CUTS, CUTS[1:])])
    assert ours.size == want.size and rel_rms(ours, want) < 1e-12
    xr = x.real.copy()
    wantr = oracle.OracleHB45().dInterp2(xr)
    st = oracle.RefDHB45()
    ours = np.concatenate([call_real(lib.quisk_dInterp2HB45, xr[a:b], st, grow=2) for a, b in zip(CUTS, CUTS[1:])])
    assert ours.size == wantr.size and rel_rms(ours, wantr) < 1e-12
    if ref is None:
        pytest.skip("oracle/_ref not present")
    st2 = oracle.RefCHB45()
    mixed = alternate([ref.quisk_cInterp2HB45, lib.quisk_cInterp2HB45], lambda fn, a, b, s: call_grow(fn, x[a:b], s, grow=2), st2)
    assert mixed.size == want.size and rel_rms(mixed, want) < 1e-12
    st3 = oracle.RefDHB45()
    mixed = alternate([lib.quisk_dInterp2HB45, ref.quisk_dInterp2HB45], lambda fn, a, b, s: call_real(fn, xr[a:b], s, grow=2), st3)
    assert mixed.size == wantr.size and rel_rms(mixed, wantr) < 1e-12


def test_differ_init_builds_the_reference_differentiator(qh, oracle):
    """quisk_filt_differInit (filter.c:35-56): taps (-1)^k / k around a zero centre; then quisk_dFilter with it, against the
    reference's own (its printf goes to stdout)."""
    lib, ref = qh.load(), oracle.ref_filter_lib()
    taps = 31
    st = oracle.RefCFilter()
    lib.quisk_filt_differInit(C.byref(st), C.c_int(taps))
    k = np.arange(taps) - (taps - 1) // 2
    want = np.where(k == 0, 0.0, (-1.0) ** np.abs(k) / np.where(k == 0, 1, k))
    got = np.array([st.dCoefs[i] for i in range(taps)])
    assert st.nTaps == taps and np.array_equal(got, want)
    x = np.random.default_rng(3).standard_normal(2000)
    ours = np.concatenate([call_real(lib.quisk_dFilter, x[a:b], st) for a, b in ((0, 700), (700, 701), (701, 2000))])
    assert rel_rms(ours, oracle.OracleFir(want, is_complex=False).dFilter(x)) < 1e-12
    if ref is not None:
        st2 = oracle.RefCFilter()
        ref.quisk_filt_differInit(C.byref(st2), C.c_int(taps))
        assert np.array_equal(np.array([st2.dCoefs[i] for i in range(taps)]), got)
        theirs = call_real(ref.quisk_dFilter, x, st2)
        assert rel_rms(ours, theirs) < 1e-12
