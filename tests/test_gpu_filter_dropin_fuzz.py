"""Seeded walks over the filter.h primitives (include/quiskhip.h group 4) against the REFERENCE ITSELF: /root/reference/filter.c compiled
in place into oracle/_ref (oracle/Makefile).  One struct is handed from block to block to this library (GPU) or to the reference
(CPU) by the throw of a die -- so the state it carries must be the reference's, in the reference's ring format, after every call
and at every block length (0, 1, primes, shorter and longer than the filter) -- while a second struct goes through the reference
alone: the two output streams must agree block by block.  -m gpu."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_filter_dropin import c_double_p, call, call_grow, call_real

pytestmark = pytest.mark.gpu

# (name, complex stream?, init function, extra int arguments, output growth)
KINDS = [
    ("quisk_cDecimate", True, "quisk_filt_cInit", lambda r: (int(r.choice([1, 2, 3, 5, 8])),), 1),
    ("quisk_cCDecimate", True, "quisk_filt_cInit", lambda r: (int(r.choice([1, 2, 4, 5])),), 1),
    ("quisk_cInterpolate", True, "quisk_filt_cInit", lambda r: (int(r.choice([2, 3, 4])),), 4),
    ("quisk_cInterpDecim", True, "quisk_filt_cInit", lambda r: (int(r.choice([2, 4, 6])), int(r.choice([3, 5]))), 6),
    ("quisk_dDecimate", False, "quisk_filt_dInit", lambda r: (int(r.choice([1, 2, 3, 5])),), 1),
    ("quisk_dFilter", False, "quisk_filt_dInit", lambda r: (), 1),
    ("quisk_dInterpolate", False, "quisk_filt_dInit", lambda r: (int(r.choice([2, 3, 4])),), 4),
    ("quisk_cDecim2HB45", True, None, lambda r: (), 1),
    ("quisk_cInterp2HB45", True, None, lambda r: (), 2),
    ("quisk_dInterp2HB45", False, None, lambda r: (), 2),
    ("quisk_cFilter", True, "quisk_filt_cInit", lambda r: (), 1),
]
# walks above 2000: ONE struct handed to another primitive of its family from call to call (the ring and decim_index are all they share)
FAMILY = {True: ["quisk_cDecimate", "quisk_cInterpolate", "quisk_cInterpDecim", "quisk_cFilter"], False: ["quisk_dDecimate", "quisk_dFilter", "quisk_dInterpolate"]}


# (124, 303, 364: one tap for several phases -- the reference then puts out zeros, found by tools/dbg/walk_sweep.py; above 1000: factors
# that change in mid-stream; above 2000: the struct handed from primitive to primitive as well)
@pytest.mark.parametrize("seed", list(range(1, 31)) + [124, 303, 364] + list(range(1001, 1013)) + list(range(2001, 2013)))
def test_one_struct_between_this_library_and_the_reference(qh, oracle, seed):
    ref = oracle.ref_filter_lib()
    if ref is None:
        pytest.skip("oracle/_ref not present (the reference's filter.c was not compiled)")
    lib = qh.load()
    rng = np.random.default_rng(4200 + seed)
    name, cpx, init, mkargs, grow = KINDS[(seed - 1) % len(KINDS)]
    args = mkargs(rng)
    ntaps = int(rng.choice([1, 7, 31, 98, 147, 245, 400, 1023]))
    taps = np.ascontiguousarray(rng.standard_normal(ntaps) / np.sqrt(ntaps))
    hb = init is None
    sts = []
    for side in (lib, ref):                     # struct A: set up by this library, struct B: by the reference
        st = (oracle.RefCHB45() if cpx else oracle.RefDHB45()) if hb else oracle.RefCFilter()
        if not hb:
            getattr(side, init)(C.byref(st), taps.ctypes.data_as(c_double_p), C.c_int(ntaps))
            if name == "quisk_cCDecimate":
                side.quisk_filt_tune.argtypes = [C.c_void_p, C.c_double, C.c_int]
                side.quisk_filt_tune(C.byref(st), 0.0731, int(seed % 2))
        sts.append(st)
    def run(fn, x, st):
        if cpx:
            return call_grow(fn, x, st, *args, grow=6)
        return call_real(fn, x, st, *args, grow=4)
    by_name = {k[0]: k for k in KINDS}
    sizes = [int(rng.choice([0, 1, 2, 3, 17, 64, 257, 700, 701, 1024, 1999, 4099])) for _ in range(18)]
    total, gpu_calls = 0, 0
    for k, n in enumerate(sizes):
        if seed > 1000 and k and rng.integers(0, 4) == 0:            # (walks above 1000: another decimation / interpolation factor in mid-stream,
            args = by_name[name][3](rng)                            #  the struct's decim_index carried over from the old one)
        if seed > 2000 and k and not hb and name != "quisk_cCDecimate" and rng.integers(0, 3) == 0:
            name = str(rng.choice(FAMILY[cpx]))
            args = by_name[name][3](rng)
        x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cpx else 0.0)
        x = np.ascontiguousarray(x if cpx else x.real)
        ours = rng.integers(0, 2) == 0 or k == 0
        gpu_calls += int(ours)
        a = run(getattr(lib if ours else ref, name), x, sts[0])
        b = run(getattr(ref, name), x, sts[1])
        assert a.size == b.size, (seed, name, args, ntaps, k, n, a.size, b.size)
        total += a.size
        if a.size:
            scale = max(np.abs(b).max(), 1e-30)
            assert np.abs(a - b).max() <= 1e-11 * scale + 1e-13, (seed, name, args, ntaps, k, n, "gpu" if ours else "ref", np.abs(a - b).max() / scale)
    assert total > 0 and gpu_calls >= 3
