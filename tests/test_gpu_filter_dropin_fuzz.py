"""The filter.h drop-ins against the reference's own compiled filter.c (oracle/_ref, built from /root/reference by
oracle/Makefile; the one place where parity is pinned by running the reference): a stream is cut at random points
(empty and one-sample blocks included) and every block goes, at random, to this library (GPU) or to the reference (CPU)
ON THE SAME STATE STRUCT; the result must be the reference's own run of the whole stream.  -m gpu."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu
c_double_p = C.POINTER(C.c_double)


def _call(fn, x, st, args, grow, cplx):
    buf = np.zeros(max(len(x) * grow, 1) + 8, dtype=np.complex128 if cplx else np.float64)
    buf[:len(x)] = x
    fn.restype = C.c_int
    n = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(len(x)), C.byref(st), *[C.c_int(a) for a in args])
    return buf[:n].copy()


CASES = [
    # name, init, args, grow, complex stream, taps
    ("quisk_cDecimate", "c", (2,), 1, True, 98), ("quisk_cDecimate", "c", (5,), 1, True, 245), ("quisk_cDecimate", "c", (3,), 1, True, 147),
    ("quisk_cFilter", "c", (), 1, True, 55), ("quisk_cCDecimate", "tune", (4,), 1, True, 120),
    ("quisk_cInterpolate", "c", (3,), 3, True, 90), ("quisk_cInterpDecim", "c", (6, 5), 6, True, 125), ("quisk_cInterpDecim", "c", (4, 5), 4, True, 245),
    ("quisk_dDecimate", "d", (4,), 1, False, 186), ("quisk_dFilter", "d", (), 1, False, 309), ("quisk_dInterpolate", "d", (2,), 2, False, 50),
    ("quisk_cDecim2HB45", "hbc", (), 1, True, 0), ("quisk_cInterp2HB45", "hbc", (), 2, True, 0), ("quisk_dInterp2HB45", "hbd", (), 2, False, 0),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_hops_between_gpu_and_reference(qh, oracle, case, seed):
    ref = oracle.ref_filter_lib()
    if ref is None:
        pytest.skip("oracle/_ref not present")
    lib = qh.load()
    name, init, args, grow, cplx, ntaps = CASES[case]
    rng = np.random.default_rng(1000 * case + seed)
    n = 6000
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0.0)
    if not cplx:
        x = x.real.copy()
    taps = np.ascontiguousarray(rng.standard_normal(max(ntaps, 1)))

    def fresh(L):
        if init in ("hbc", "hbd"):
            return oracle.RefCHB45() if init == "hbc" else oracle.RefDHB45()
        st = oracle.RefCFilter()
        (L.quisk_filt_dInit if init == "d" else L.quisk_filt_cInit)(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(ntaps))
        if init == "tune":
            L.quisk_filt_tune.argtypes = [C.c_void_p, C.c_double, C.c_int]
            L.quisk_filt_tune(C.byref(st), 0.0413, int(seed % 2))
        return st

    want = _call(getattr(ref, name), x, fresh(ref), args, grow, cplx)
    cuts = np.unique(np.concatenate([[0, n], rng.integers(0, n, 14)]))
    cuts = np.sort(np.concatenate([cuts, cuts[rng.integers(0, cuts.size, 3)]]))          # a few empty blocks
    extra = cuts[rng.integers(0, cuts.size - 1, 3)] + 1                                   # and one-sample ones
    cuts = np.sort(np.concatenate([cuts, extra[extra <= n]]))
    st = fresh(lib if seed % 2 else ref)
    parts, used = [], [0, 0]
    for a, b in zip(cuts[:-1], cuts[1:]):
        who = int(rng.integers(0, 2))
        used[who] += 1
        parts.append(_call(getattr(lib if who else ref, name), x[a:b], st, args, grow, cplx))
    got = np.concatenate(parts)
    assert used[0] and used[1]
    assert got.size == want.size, (name, got.size, want.size)
    assert rel_rms(got, want) < 1e-12, (name, rel_rms(got, want))
