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
