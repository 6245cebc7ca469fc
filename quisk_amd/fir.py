"""Python host side of the batched FIR decimator bank (include/quiskhip.h group 3).

Mirrors the calling convention of Quisk's filter.c primitives: create once with the taps
(quisk_filt_cInit, filter.c:9-20), then feed blocks of any length; the decimation phase and the
filter history carry over between calls.
"""
import ctypes as C

import numpy as np

from .lib import load, check, QuiskHipError

F64, F32 = 0, 1


def hb45_taps():
    t = np.zeros(43, dtype=np.float64)
    load().qh_hb45_taps(t.ctypes.data)
    return t


class FirBank:
    def __init__(self, nch, taps, decim, dtype=F64, device=0, stream=None):
        self._L = load()
        taps = np.asarray(taps)
        re = np.ascontiguousarray(taps.real, dtype=np.float64)
        im = np.ascontiguousarray(taps.imag, dtype=np.float64) if np.iscomplexobj(taps) else None
        self._h = self._L.qh_fir_create(device, nch, re.ctypes.data, im.ctypes.data if im is not None else None,
                                        re.size, decim, dtype, stream)
        if not self._h:
            raise QuiskHipError("qh_fir_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch, self.decim, self.dtype = nch, decim, dtype
        self.np_dtype = np.complex128 if dtype == F64 else np.complex64

    def out_count(self, n_in):
        return self._L.qh_fir_out_count(self._h, n_in)

    def process_ptr(self, d_in, in_stride, n_in, d_out, out_stride):
        n = C.c_int(0)
        check(self._L.qh_fir_process(self._h, d_in, in_stride, n_in, d_out, out_stride, C.byref(n)))
        return n.value

    def process_host(self, x):
        x = np.ascontiguousarray(x, dtype=self.np_dtype)
        if x.ndim != 2 or x.shape[0] != self.nch:
            raise ValueError("expected [nch, n]")
        nmax = max(self.out_count(x.shape[1]), 1)
        out = np.empty((self.nch, nmax), dtype=self.np_dtype)
        n = C.c_int(0)
        check(self._L.qh_fir_process_host(self._h, x.ctypes.data, x.shape[1], x.shape[1], out.ctypes.data, nmax, C.byref(n)))
        return out[:, :n.value].copy()

    def reset(self):
        check(self._L.qh_fir_reset(self._h))

    def synchronize(self):
        check(self._L.qh_fir_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_fir_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HalfBandCascade:
    """`nstage` chained quisk_cDecim2HB45 decimators (filter.c:377-417, chained at quisk.c:1772-1796) for `nch`
    streams in one pass over HBM (C ABI group 3b).  Calls take multiples of 2**nstage samples."""

    def __init__(self, nch, nstage, dtype=F64, device=0, stream=None):
        self._L = load()
        self._h = self._L.qh_hbc_create(device, nch, nstage, dtype, stream)
        if not self._h:
            raise QuiskHipError("qh_hbc_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch, self.nstage, self.decim, self.dtype = nch, nstage, 1 << nstage, dtype
        self.np_dtype = np.complex128 if dtype == F64 else np.complex64

    def process_ptr(self, d_in, in_stride, n_in, d_out, out_stride):
        check(self._L.qh_hbc_process(self._h, d_in, in_stride, n_in, d_out, out_stride))
        return n_in >> self.nstage

    def process_host(self, x):
        x = np.ascontiguousarray(x, dtype=self.np_dtype)
        if x.ndim != 2 or x.shape[0] != self.nch:
            raise ValueError("expected [nch, n]")
        n_out = x.shape[1] >> self.nstage
        out = np.empty((self.nch, max(n_out, 1)), dtype=self.np_dtype)
        check(self._L.qh_hbc_process_host(self._h, x.ctypes.data, x.shape[1], x.shape[1], out.ctypes.data, max(n_out, 1)))
        return out[:, :n_out].copy()

    def reset(self):
        check(self._L.qh_hbc_reset(self._h))

    def synchronize(self):
        check(self._L.qh_hbc_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_hbc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RationalFir:
    """quisk_cInterpDecim / quisk_cInterpolate (filter.c:287-324,131-165) for `nch` complex streams (C ABI group
    3c): interpolate by `interp`, real taps, keep every `decim`-th; gain `interp`.  Two real streams can ride as
    the real and imaginary parts (quisk_dInterpolate)."""

    def __init__(self, nch, taps, interp, decim=1, dtype=F64, device=0, stream=None):
        self._L = load()
        t = np.ascontiguousarray(taps, dtype=np.float64)
        self._h = self._L.qh_rat_create(device, nch, t.ctypes.data, t.size, interp, decim, dtype, stream)
        if not self._h:
            raise QuiskHipError("qh_rat_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch, self.interp, self.decim, self.dtype = nch, interp, decim, dtype
        self.np_dtype = np.complex128 if dtype == F64 else np.complex64

    def out_count(self, n_in):
        return self._L.qh_rat_out_count(self._h, n_in)

    @property
    def phase(self):
        return self._L.qh_rat_phase(self._h)

    def process_ptr(self, d_in, in_stride, n_in, d_out, out_stride):
        n = C.c_int(0)
        check(self._L.qh_rat_process(self._h, d_in, in_stride, n_in, d_out, out_stride, C.byref(n)))
        return n.value

    def process_host(self, x):
        x = np.ascontiguousarray(x, dtype=self.np_dtype)
        if x.ndim != 2 or x.shape[0] != self.nch:
            raise ValueError("expected [nch, n]")
        nmax = max(self.out_count(x.shape[1]), 1)
        out = np.empty((self.nch, nmax), dtype=self.np_dtype)
        n = C.c_int(0)
        check(self._L.qh_rat_process_host(self._h, x.ctypes.data, x.shape[1], x.shape[1], out.ctypes.data, nmax, C.byref(n)))
        return out[:, :n.value].copy()

    def reset(self):
        check(self._L.qh_rat_reset(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_rat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
