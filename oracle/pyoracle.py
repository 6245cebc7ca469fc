"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (quisk_amd) must never import it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_double_p = C.POINTER(C.c_double)


def build(ref=None):
    """Compile liboracle.so (and _ref/ when the reference tree is present)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    if ref is None:
        ref = os.path.exists("/root/reference/filter.c")
    if ref:
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.wo_open.restype = C.c_void_p
        L.wo_open.argtypes = [C.c_int] * 5 + [C.c_double] * 4
        L.wo_close.argtypes = [C.c_void_p]
        L.wo_fexchange0.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.wo_xrxa_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_xrxa_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        for n in ("wo_dsp_insize", "wo_dsp_outsize", "wo_out_size"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = C.c_int
        for n in ("wo_SetRXAMode", "wo_RXASetNC", "wo_SetRXAShiftRun", "wo_RXANBPSetRun", "wo_SetRXABandpassRun",
                  "wo_SetRXAAGCMode", "wo_SetRXAPanelRun", "wo_SetRXAPanelSelect", "wo_SetRXAPanelCopy",
                  "wo_SetRXAAMDSBMode", "wo_SetRXAAMDFadeLevel", "wo_SetRXACTCSSRun"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
            getattr(L, n).restype = None
        for n in ("wo_SetRXAShiftFreq", "wo_SetRXAAGCFixed", "wo_SetRXAPanelGain1", "wo_SetRXAFMDeviation",
                  "wo_SetRXACTCSSFreq"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_double]
            getattr(L, n).restype = None
        for n in ("wo_RXASetPassband", "wo_RXANBPSetFreqs", "wo_SetRXABandpassFreqs", "wo_SetRXAPanelGain2"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_double, C.c_double]
            getattr(L, n).restype = None
        L.wo_GetRXAMeter.argtypes = [C.c_void_p, C.c_int]
        L.wo_GetRXAMeter.restype = C.c_double
        L.wo_fir_bandpass.restype = C.c_void_p
        L.wo_fir_bandpass.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double]
        L.wo_calc_resample_taps.restype = C.c_void_p
        L.wo_calc_resample_taps.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_double] + [C.POINTER(C.c_int)] * 4
        L.wo_fircore_create.restype = C.c_void_p
        L.wo_fircore_create.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.wo_fircore_destroy.argtypes = [C.c_void_p]
        L.wo_fircore_exec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wo_resample_create.restype = C.c_void_p
        L.wo_resample_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_double]
        L.wo_resample_destroy.argtypes = [C.c_void_p]
        L.wo_resample_exec.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.wo_resample_exec.restype = C.c_int
        L.fo_fft.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.free_ = C.CDLL(None).free
        L.free_.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def fft(x, sign=-1):
    x = np.ascontiguousarray(x, dtype=np.complex128).copy()
    lib().fo_fft(x.ctypes.data, x.size, sign)
    return x


def fir_bandpass(N, f_low, f_high, samplerate, wintype, rtype, scale):
    L = lib()
    p = L.wo_fir_bandpass(N, f_low, f_high, samplerate, wintype, rtype, scale)
    n = N * (2 if rtype == 1 else 1)
    arr = np.ctypeslib.as_array(C.cast(p, c_double_p), shape=(n,)).copy()
    L.free_(p)
    return arr.view(np.complex128) if rtype == 1 else arr


def resample_taps(in_rate, out_rate, fc=0.0, ncoef=0, gain=1.0):
    L = lib()
    a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    p = L.wo_calc_resample_taps(in_rate, out_rate, fc, ncoef, gain, a, b, c, d)
    arr = np.ctypeslib.as_array(C.cast(p, c_double_p), shape=(c.value,)).copy()
    L.free_(p)
    return arr, a.value, b.value, c.value, d.value


class Fircore:
    def __init__(self, size, nc, impulse):
        self.size = size
        imp = np.ascontiguousarray(impulse, dtype=np.complex128)
        assert imp.size == nc
        self.h = lib().wo_fircore_create(size, nc, imp.ctypes.data)

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        assert x.size % self.size == 0
        out = np.empty_like(x)
        for b in range(x.size // self.size):
            lib().wo_fircore_exec(self.h, x[b * self.size:].ctypes.data, out[b * self.size:].ctypes.data)
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().wo_fircore_destroy(self.h)
            self.h = None


class Resample:
    def __init__(self, in_rate, out_rate, fc=0.0, ncoef=0, gain=1.0):
        self.h = lib().wo_resample_create(in_rate, out_rate, fc, ncoef, gain)
        self.ratio = out_rate / in_rate

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        out = np.empty(int(x.size * self.ratio) + 8, dtype=np.complex128)
        n = lib().wo_resample_exec(self.h, x.ctypes.data, x.size, out.ctypes.data)
        return out[:n].copy()

    def __del__(self):
        if getattr(self, "h", None):
            lib().wo_resample_destroy(self.h)
            self.h = None


class WdspChannel:
    """One oracle RXA channel; method names follow the WDSP exports (minus the channel argument)."""

    def __init__(self, in_size, dsp_size, in_rate, dsp_rate, out_rate,
                 tdelayup=0.010, tslewup=0.025, tdelaydown=0.0, tslewdown=0.010):
        self.L = lib()
        self.h = self.L.wo_open(in_size, dsp_size, in_rate, dsp_rate, out_rate, tdelayup, tslewup, tdelaydown, tslewdown)
        self.in_size = in_size
        self.out_size = self.L.wo_out_size(self.h)
        self.dsp_insize = self.L.wo_dsp_insize(self.h)
        self.dsp_outsize = self.L.wo_dsp_outsize(self.h)

    def __getattr__(self, name):
        f = getattr(lib(), "wo_" + name)

        def call(*args):
            return f(self.h, *args)
        return call

    def fexchange0(self, x):
        """x: complex128, multiple of in_size.  Returns (out, errors)."""
        x = np.ascontiguousarray(x, dtype=np.complex128)
        assert x.size % self.in_size == 0
        nb = x.size // self.in_size
        out = np.empty(nb * self.out_size, dtype=np.complex128)
        err = C.c_int(0)
        errs = 0
        for b in range(nb):
            self.L.wo_fexchange0(self.h, x[b * self.in_size:].ctypes.data, out[b * self.out_size:].ctypes.data, err)
            errs += (err.value != 0)
        return out, errs

    def xrxa(self, x):
        """Run the DSP chain block by block without iobuffs latency / slew."""
        x = np.ascontiguousarray(x, dtype=np.complex128)
        assert x.size % self.dsp_insize == 0
        nb = x.size // self.dsp_insize
        out = np.empty(nb * self.dsp_outsize, dtype=np.complex128)
        self.L.wo_xrxa_blocks(self.h, x.ctypes.data, out.ctypes.data, nb)
        return out

    def close(self):
        if self.h:
            self.L.wo_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ref_filter_lib():
    """The reference's own filter.c compiled into oracle/_ref/ (None when it was not built)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libquisk_filter_ref.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF
