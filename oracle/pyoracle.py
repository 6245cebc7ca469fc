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
        for n in ("wo_SetRXAAGCAttack", "wo_SetRXAAGCDecay", "wo_SetRXAAGCHang", "wo_SetRXAAGCSlope", "wo_SetRXAAGCHangThreshold"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
            getattr(L, n).restype = None
        for n in ("wo_SetRXAAMSQThreshold", "wo_SetRXAAMSQMaxTail", "wo_SetRXAEMNRaeZetaThresh", "wo_SetRXAEMNRaePsi",
                  "wo_SetRXAEMNRtrainZetaThresh", "wo_SetRXAEMNRtrainT2"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_double]
            getattr(L, n).restype = None
        for n in ("wo_SetRXAANFVals", "wo_SetRXAANRVals"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
            getattr(L, n).restype = None
        L.wo_SetRXAAGCTop.argtypes = [C.c_void_p, C.c_double]
        L.wo_SetRXAAGCTop.restype = None
        for n in ("wo_SetRXAMode", "wo_RXASetNC", "wo_SetRXAShiftRun", "wo_RXANBPSetRun", "wo_SetRXABandpassRun",
                  "wo_SetRXAAGCMode", "wo_SetRXAPanelRun", "wo_SetRXAPanelSelect", "wo_SetRXAPanelCopy",
                  "wo_SetRXAAMDSBMode", "wo_SetRXAAMDFadeLevel", "wo_SetRXACTCSSRun", "wo_SetRXAAMDRun", "wo_RXASetMP", "wo_SetRXAFMLimRun",
                  "wo_SetRXAEMNRRun", "wo_SetRXASNBARun", "wo_SetRXAEMNRgainMethod", "wo_SetRXAEMNRnpeMethod", "wo_SetRXAEMNRaeRun", "wo_SetRXAEMNRPosition",
                  "wo_SetRXAAMSQRun", "wo_SetRXAANFRun", "wo_SetRXAANRRun", "wo_SetRXAANFPosition", "wo_SetRXAANRPosition", "wo_SetRXASNBAovrlp"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
            getattr(L, n).restype = None
        for n in ("wo_SetRXAShiftFreq", "wo_SetRXAAGCFixed", "wo_SetRXAPanelGain1", "wo_SetRXAFMDeviation",
                  "wo_SetRXACTCSSFreq"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_double]
            getattr(L, n).restype = None
        for n in ("wo_RXASetPassband", "wo_RXANBPSetFreqs", "wo_SetRXABandpassFreqs", "wo_SetRXAPanelGain2"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_double, C.c_double]
            getattr(L, n).restype = None
        L.wo_SetRXASNBATuning.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.wo_SetRXASNBATuning.restype = None
        L.wo_GetRXAMeter.argtypes = [C.c_void_p, C.c_int]
        L.wo_SetRXAFMLimGain.argtypes = [C.c_void_p, C.c_double]
        for n in ("wo_RXANBPAddNotch", "wo_RXANBPEditNotch"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
        L.wo_RXANBPDeleteNotch.argtypes = [C.c_void_p, C.c_int]
        for n in ("wo_RXANBPSetTuneFrequency", "wo_RXANBPSetShiftFrequency"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_double]
        for n in ("wo_RXANBPSetNotchesRun", "wo_RXANBPSetWindow", "wo_RXANBPSetAutoIncrease"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
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


_EMNR = None


def emnr_tables():
    """GG / GGS / zetaHat (quisk_amd/data/wdsp_emnr_tables.npz, made by tools/extract_wdsp_emnr_tables.py) or None."""
    global _EMNR
    if _EMNR is None:
        path = os.path.join(os.path.dirname(_HERE), "quisk_amd", "data", "wdsp_emnr_tables.npz")
        if not os.path.exists(path):
            return None
        z = np.load(path)
        _EMNR = {"GG": np.ascontiguousarray(z["GG"], dtype=np.float64), "GGS": np.ascontiguousarray(z["GGS"], dtype=np.float64),
                 "zeta_hat": np.ascontiguousarray(z["zeta_hat"], dtype=np.float64),
                 "zeta_valid": np.ascontiguousarray(z["zeta_valid"], dtype=np.int32), "zeta_range": z["zeta_range"],
                 "zeta_dims": z["zeta_dims"]}
    return _EMNR


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
        t = emnr_tables()
        if t is not None:       # WDSP reads these from the files `calculus` / `zetaHat.bin` when a channel is created (emnr.c:317-334)
            self.set_emnr_tables(t)

    def set_emnr_tables(self, t):
        """Other tables than the shipped ones (a test's stand-in for a `calculus` / `zetaHat.bin` with other numbers in the working directory)."""
        self._emnr_tables = t = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in t.items()}
        self.L.wo_SetEMNRTables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_double] * 4
        self.L.wo_SetEMNRTables(self.h, t["GG"].ctypes.data, t["GGS"].ctypes.data, t["zeta_hat"].ctypes.data, t["zeta_valid"].ctypes.data,
                                *[float(v) for v in t["zeta_range"]])

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


class OracleWdspShim:
    """The restated quisk_wdsp.c shim (wo_shim_*) in front of a callable fexchange0(in_ptr, out_ptr) -> error."""

    def __init__(self, fexchange0):
        self.L = lib()
        self.L.wo_shim_create.restype = C.c_void_p
        self.L.wo_shim_destroy.argtypes = [C.c_void_p]
        self.L.wo_shim_set_parameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._fn_t = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int))
        self.L.wo_shim_fexchange0.argtypes = [C.c_void_p, self._fn_t, C.c_void_p, C.c_void_p, C.c_int]
        self.L.wo_shim_fexchange0.restype = C.c_int
        self.h = self.L.wo_shim_create()

        def tramp(ctx, pin, pout, perr):
            perr[0] = int(fexchange0(pin, pout))
        self._cb = self._fn_t(tramp)

    def set_parameter(self, in_size=-1, in_use=-1):
        self.L.wo_shim_set_parameter(self.h, in_size, in_use)

    def fexchange0(self, buf, n):
        """buf: complex128 work array, first n entries are input; returns the count written back."""
        return self.L.wo_shim_fexchange0(self.h, self._cb, None, buf.ctypes.data, n)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.wo_shim_destroy(self.h)
            self.h = None


_REF_WDSP = None


def ref_wdsp_shim_lib():
    """The reference's own quisk_wdsp.c compiled into oracle/_ref/ (None when it was not built).  Loaded with PyDLL:
    quisk_wdsp_set_parameter is a CPython function (PyArg_ParseTupleAndKeywords) and must run with the GIL held."""
    global _REF_WDSP
    if _REF_WDSP is None:
        path = os.path.join(_HERE, "_ref", "libquisk_wdsp_ref.so")
        if not os.path.exists(path):
            return None
        L = C.PyDLL(path)
        L.quisk_wdsp_set_parameter.restype = C.py_object
        L.quisk_wdsp_set_parameter.argtypes = [C.py_object, C.py_object, C.py_object]
        L.wdspFexchange0.restype = C.c_int
        L.wdspFexchange0.argtypes = [C.c_int, C.c_void_p, C.c_int]
        _REF_WDSP = L
    return _REF_WDSP


def ref_wdsp_set_parameter(channel, **kw):
    """QS.wdsp_set_parameter(channel, in_size=, fexchange0=<address>, in_use=) of the reference build."""
    return ref_wdsp_shim_lib().quisk_wdsp_set_parameter(None, (int(channel),), dict(kw))


def ref_filter_lib():
    """The reference's own filter.c compiled into oracle/_ref/ (None when it was not built)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libquisk_filter_ref.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF


# ---------------------------------------------------------------------------------------------
# Quisk filter.c primitives: the restatement (quisk_oracle.c) and the reference build (_ref)
# ---------------------------------------------------------------------------------------------
class QoFir(C.Structure):       # oracle/quisk_oracle.h: qo_fir
    _fields_ = [("taps", c_double_p), ("ctaps", c_double_p), ("ntaps", C.c_int), ("phase", C.c_int),
                ("pos", C.c_int), ("hist", c_double_p), ("is_complex", C.c_int)]


class QoHb45(C.Structure):      # oracle/quisk_oracle.h: qo_hb45
    _fields_ = [("toggle", C.c_int), ("samples", C.c_double * 44), ("center", C.c_double * 22)]


class RefCFilter(C.Structure):  # struct quisk_cFilter / quisk_dFilter, /root/reference/filter.h:1-21 (same layout)
    _fields_ = [("dCoefs", c_double_p), ("cpxCoefs", C.c_void_p), ("nBuf", C.c_int), ("nTaps", C.c_int),
                ("decim_index", C.c_int), ("samples", C.c_void_p), ("ptSamp", C.c_void_p), ("buf", C.c_void_p)]


class RefCHB45(C.Structure):    # struct quisk_cHB45Filter, filter.h:23-29
    _fields_ = [("cBuf", C.c_void_p), ("nBuf", C.c_int), ("toggle", C.c_int),
                ("samples", C.c_double * 44), ("center", C.c_double * 22)]


class RefDHB45(C.Structure):    # struct quisk_dHB45Filter, filter.h:31-37
    _fields_ = [("dBuf", C.c_void_p), ("nBuf", C.c_int), ("toggle", C.c_int),
                ("samples", C.c_double * 22), ("center", C.c_double * 11)]


class OracleFir:
    """qo_fir wrapper.  Methods take/return numpy arrays (complex128 or float64) and keep state."""

    def __init__(self, taps, is_complex=True):
        self.L = lib()
        self.taps = np.ascontiguousarray(taps, dtype=np.float64)
        self.f = QoFir()
        self.is_complex = is_complex
        self.L.qo_fir_init(C.byref(self.f), self.taps.ctypes.data_as(c_double_p), self.taps.size, 1 if is_complex else 0)

    def tune(self, freq, ssb_upper):
        self.L.qo_fir_tune.argtypes = [C.c_void_p, C.c_double, C.c_int]
        self.L.qo_fir_tune(C.byref(self.f), freq, ssb_upper)

    def _run(self, name, x, *args, grow=1):
        dt = np.complex128 if self.is_complex else np.float64
        x = np.ascontiguousarray(x, dtype=dt)
        buf = np.zeros(max(x.size * grow, 1) + 8, dtype=dt)
        buf[:x.size] = x
        fn = getattr(self.L, name)
        fn.restype = C.c_int
        n = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(x.size), C.byref(self.f), *[C.c_int(a) for a in args])
        return buf[:n].copy()

    def cDecimate(self, x, decim): return self._run("qo_cDecimate", x, decim)
    def cCDecimate(self, x, decim): return self._run("qo_cCDecimate", x, decim)
    def dDecimate(self, x, decim): return self._run("qo_dDecimate", x, decim)
    def cInterpolate(self, x, interp): return self._run("qo_cInterpolate", x, interp, grow=interp)
    def dInterpolate(self, x, interp): return self._run("qo_dInterpolate", x, interp, grow=interp)
    def cInterpDecim(self, x, interp, decim): return self._run("qo_cInterpDecim", x, interp, decim, grow=interp)
    def dFilter(self, x): return self._run("qo_dFilter", x)


class OracleHB45:
    def __init__(self):
        self.L = lib()
        self.f = QoHb45()

    def _run(self, name, x, dt, grow=1):
        x = np.ascontiguousarray(x, dtype=dt)
        buf = np.zeros(max(x.size * grow, 1) + 8, dtype=dt)
        buf[:x.size] = x
        fn = getattr(self.L, name)
        fn.restype = C.c_int
        n = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(x.size), C.byref(self.f))
        return buf[:n].copy()

    def cDecim2(self, x): return self._run("qo_cDecim2HB45", x, np.complex128)
    def cInterp2(self, x): return self._run("qo_cInterp2HB45", x, np.complex128, grow=2)
    def dInterp2(self, x): return self._run("qo_dInterp2HB45", x, np.float64, grow=2)


class RefFir:
    """The reference's own filter.c (oracle/_ref) behind the same interface as OracleFir."""

    def __init__(self, taps, is_complex=True):
        self.R = ref_filter_lib()
        if self.R is None:
            raise RuntimeError("oracle/_ref/libquisk_filter_ref.so is not built (needs /root/reference)")
        self.taps = np.ascontiguousarray(taps, dtype=np.float64)
        self.f = RefCFilter()
        self.is_complex = is_complex
        init = self.R.quisk_filt_cInit if is_complex else self.R.quisk_filt_dInit
        init(C.byref(self.f), self.taps.ctypes.data_as(c_double_p), C.c_int(self.taps.size))

    def tune(self, freq, ssb_upper):
        self.R.quisk_filt_tune.argtypes = [C.c_void_p, C.c_double, C.c_int]
        self.R.quisk_filt_tune(C.byref(self.f), freq, ssb_upper)

    def _run(self, name, x, *args, grow=1):
        dt = np.complex128 if self.is_complex else np.float64
        x = np.ascontiguousarray(x, dtype=dt)
        buf = np.zeros(max(x.size * grow, 1) + 8, dtype=dt)
        buf[:x.size] = x
        fn = getattr(self.R, name)
        fn.restype = C.c_int
        n = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(x.size), C.byref(self.f), *[C.c_int(a) for a in args])
        return buf[:n].copy()

    def cDecimate(self, x, decim): return self._run("quisk_cDecimate", x, decim)
    def cCDecimate(self, x, decim): return self._run("quisk_cCDecimate", x, decim)
    def dDecimate(self, x, decim): return self._run("quisk_dDecimate", x, decim)
    def cInterpolate(self, x, interp): return self._run("quisk_cInterpolate", x, interp, grow=interp)
    def dInterpolate(self, x, interp): return self._run("quisk_dInterpolate", x, interp, grow=interp)
    def cInterpDecim(self, x, interp, decim): return self._run("quisk_cInterpDecim", x, interp, decim, grow=interp)
    def dFilter(self, x): return self._run("quisk_dFilter", x)


class RefHB45:
    def __init__(self):
        self.R = ref_filter_lib()
        if self.R is None:
            raise RuntimeError("oracle/_ref/libquisk_filter_ref.so is not built (needs /root/reference)")
        self.fc = RefCHB45()
        self.fd = RefDHB45()

    def _run(self, name, f, x, dt, grow=1):
        x = np.ascontiguousarray(x, dtype=dt)
        buf = np.zeros(max(x.size * grow, 1) + 8, dtype=dt)
        buf[:x.size] = x
        fn = getattr(self.R, name)
        fn.restype = C.c_int
        n = fn(buf.ctypes.data_as(C.c_void_p), C.c_int(x.size), C.byref(f))
        return buf[:n].copy()

    def cDecim2(self, x): return self._run("quisk_cDecim2HB45", self.fc, x, np.complex128)
    def cInterp2(self, x): return self._run("quisk_cInterp2HB45", self.fc, x, np.complex128, grow=2)
    def dInterp2(self, x): return self._run("quisk_dInterp2HB45", self.fd, x, np.float64, grow=2)


def ref_table(name, n):
    """A coefficient table exported by the reference build (filters.h via filter.c), e.g. quiskFilt48dec24Coefs."""
    R = ref_filter_lib()
    if R is None:
        raise RuntimeError("oracle/_ref is not built")
    arr = (C.c_double * n).in_dll(R, name)
    return np.array(arr, dtype=np.float64)


class OracleGraph:
    """qo_graph wrapper: feed(x) then get(zoom, deltaf) -> (pixels, smeter_db, count) or None."""

    def __init__(self, fft_size, data_width, rate):
        L = lib()
        L.qo_graph_create.restype = C.c_void_p
        L.qo_graph_create.argtypes = [C.c_int, C.c_int, C.c_double]
        L.qo_graph_free.argtypes = [C.c_void_p]
        L.qo_graph_set_smeter_band.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.qo_graph_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.qo_graph_get.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_double)]
        self.L, self.data_width = L, data_width
        self.h = L.qo_graph_create(fft_size, data_width, rate)

    def set_smeter_band(self, f_start, bandwidth):
        self.L.qo_graph_set_smeter_band(self.h, f_start, bandwidth)

    def feed(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        return self.L.qo_graph_feed(self.h, x.ctypes.data, x.size)

    def get(self, zoom=1.0, deltaf=0.0):
        pix = np.empty(self.data_width, dtype=np.float64)
        sm = C.c_double(0)
        n = self.L.qo_graph_get(self.h, zoom, deltaf, pix.ctypes.data, C.byref(sm))
        return None if n <= 0 else (pix, sm.value, n)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.qo_graph_free(self.h)
            self.h = None


class QoRxTables(C.Structure):      # oracle/quisk_rx_oracle.h: qo_rx_tables
    _fields_ = [(n, c_double_p) for n in ("f48dec24", "f144d3", "f240d5", "audio24p4", "audio24p6", "lp48", "fmhp",
                                          "f300d5", "sdriq53", "sdriq111", "sdriq133", "sdriq167", "sdriq185")]


class OracleQuiskRx:
    """qo_rx wrapper: one Quisk-native receiver; process(x) returns the 48 ksps complex output."""

    def __init__(self, sample_rate, tables):
        L = lib()
        L.qo_rx_create.restype = C.c_void_p
        L.qo_rx_create.argtypes = [C.c_int, C.c_void_p]
        L.qo_rx_free.argtypes = [C.c_void_p]
        L.qo_rx_set_tune.argtypes = [C.c_void_p, C.c_int]
        L.qo_rx_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.qo_rx_set_filters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.qo_rx_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.qo_rx_filter_srate.argtypes = [C.c_void_p]
        self.L = L
        keys = ("quiskFilt48dec24Coefs", "quiskFilt144D3Coefs", "quiskFilt240D5CoefsSharp", "quiskAudio24p4Coefs",
                "quiskAudio24p6Coefs", "quiskLpFilt48Coefs", "quiskAudioFmHpCoefs", "quiskFilt300D5Coefs",
                "quiskFilt53D1Coefs", "quiskFilt111D2Coefs", "quiskFilt133D2Coefs", "quiskFilt167D3Coefs",
                "quiskFilt185D3Coefs")
        self._keep = [np.ascontiguousarray(tables[k], dtype=np.float64) for k in keys]
        self._t = QoRxTables(*[a.ctypes.data_as(c_double_p) for a in self._keep])
        self.h = L.qo_rx_create(sample_rate, C.byref(self._t))
        if not self.h:
            raise ValueError("qo_rx_create failed for rate %d" % sample_rate)
        L.qo_rx_set_bandwidth.argtypes = [C.c_void_p, C.c_int]
        L.qo_rx_decim_srate.argtypes = [C.c_void_p]
        self.rate = sample_rate

    def set_tune(self, f): self.L.qo_rx_set_tune(self.h, int(f))
    def set_mode(self, m): self.L.qo_rx_set_mode(self.h, int(m))
    def set_bandwidth(self, bw): self.L.qo_rx_set_bandwidth(self.h, int(bw))

    def set_ssb_squelch(self, enabled, level):
        self.L.qo_rx_set_ssb_squelch.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.L.qo_rx_set_ssb_squelch(self.h, int(enabled), int(level))

    def set_squelch(self, level):
        self.L.qo_rx_set_squelch.argtypes = [C.c_void_p, C.c_double]
        self.L.qo_rx_set_squelch(self.h, float(level))

    def set_auto_notch(self, on, rit_freq=0):
        self.L.qo_rx_set_auto_notch.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.L.qo_rx_set_auto_notch(self.h, int(on), int(rit_freq))

    def set_noise_blanker(self, level):
        self.L.qo_rx_set_noise_blanker.argtypes = [C.c_void_p, C.c_int]
        self.L.qo_rx_set_noise_blanker(self.h, int(level))

    def set_agc(self, on, release_gain=80.0):
        self.L.qo_rx_set_agc.argtypes = [C.c_void_p, C.c_int, C.c_double]
        self.L.qo_rx_set_agc(self.h, int(on), float(release_gain))

    def decim_srate(self): return self.L.qo_rx_decim_srate(self.h)
    def filter_srate(self): return self.L.qo_rx_filter_srate(self.h)

    def set_filters(self, fI, fQ):
        fI = np.ascontiguousarray(fI, dtype=np.float64)
        fQ = np.ascontiguousarray(fQ, dtype=np.float64)
        self.L.qo_rx_set_filters(self.h, fI.ctypes.data, fQ.ctypes.data, fI.size)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        buf = np.zeros(max(x.size, 16) * 2, dtype=np.complex128)
        buf[:x.size] = x
        n = self.L.qo_rx_process(self.h, buf.ctypes.data, x.size)
        return buf[:n].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.qo_rx_free(self.h)
            self.h = None


class OracleQuiskBlock:
    """qo_ps wrapper: quisk_process_samples as a whole (quisk.c:2289-2742) with the reference's setter names.
    process(x) returns the block at the playback rate."""

    _TABLE_KEYS = ("quiskFilt48dec24Coefs", "quiskFilt144D3Coefs", "quiskFilt240D5CoefsSharp", "quiskAudio24p4Coefs",
                   "quiskAudio24p6Coefs", "quiskLpFilt48Coefs", "quiskAudioFmHpCoefs", "quiskFilt300D5Coefs",
                   "quiskFilt53D1Coefs", "quiskFilt111D2Coefs", "quiskFilt133D2Coefs", "quiskFilt167D3Coefs",
                   "quiskFilt185D3Coefs")

    def __init__(self, sample_rate, playback_rate, tables):
        L = lib()
        V, I, D = C.c_void_p, C.c_int, C.c_double
        L.qo_ps_create.restype = V
        L.qo_ps_create.argtypes = [I, I, V]
        sig = {"qo_ps_free": [V], "qo_ps_set_tune": [V, I, I], "qo_ps_set_mode": [V, I], "qo_ps_set_filters": [V, V, V, I, I, I],
               "qo_ps_set_agc": [V, D], "qo_ps_set_split_rxtx": [V, I], "qo_ps_set_multirx_play_channel": [V, I],
               "qo_ps_set_multirx_play_method": [V, I], "qo_ps_set_multirx_freq": [V, I, I], "qo_ps_set_multirx_mode": [V, I, I],
               "qo_ps_set_multirx_count": [V, I], "qo_ps_set_sub_rx1_output": [V, I], "qo_ps_multirx_samples": [V, I, V, I],
               "qo_ps_set_key_state": [V, I, I, I, I], "qo_ps_set_sidetone": [V, D, I, I], "qo_ps_set_kill_audio": [V, I],
               "qo_ps_invert_spectrum": [V, I], "qo_ps_set_noise_blanker": [V, I], "qo_ps_set_auto_notch": [V, I],
               "qo_ps_set_squelch": [V, D], "qo_ps_set_ssb_squelch": [V, I, I], "qo_ps_add_tone": [V, I],
               "qo_ps_measure_frequency": [V, I], "qo_ps_set_graph": [V, V], "qo_ps_set_wdsp": [V, V, V, V],
               "qo_ps_sub_rx1_audio": [V, V, I], "qo_ps_squelch_flags": [V], "qo_ps_overrun": [V], "qo_ps_process": [V, V, I],
               "qo_ps_restart_bank": [V, I]}
        for name, args in sig.items():
            getattr(L, name).argtypes = args
        L.qo_ps_measure_frequency.restype = D
        self.L = L
        self._keep = [np.ascontiguousarray(tables[k], dtype=np.float64) for k in self._TABLE_KEYS]
        self._t = QoRxTables(*[a.ctypes.data_as(c_double_p) for a in self._keep])
        self.ratio = max(1, playback_rate // 48000)
        self.h = L.qo_ps_create(sample_rate, playback_rate, C.byref(self._t))
        self._hooks = []

    def set_tune(self, rx, tx=0): self.L.qo_ps_set_tune(self.h, int(rx), int(tx))
    def set_rx_mode(self, m): self.L.qo_ps_set_mode(self.h, int(m))

    def set_filters(self, fI, fQ, bw, nFilter=0):
        fI = np.ascontiguousarray(fI, dtype=np.float64)
        fQ = np.ascontiguousarray(fQ, dtype=np.float64)
        self.L.qo_ps_set_filters(self.h, fI.ctypes.data, fQ.ctypes.data, fI.size, int(bw), int(nFilter))

    def set_agc(self, level): self.L.qo_ps_set_agc(self.h, float(level))
    def set_split_rxtx(self, s): self.L.qo_ps_set_split_rxtx(self.h, int(s))
    def set_multirx_play_channel(self, ch): self.L.qo_ps_set_multirx_play_channel(self.h, int(ch))
    def set_multirx_play_method(self, m): self.L.qo_ps_set_multirx_play_method(self.h, int(m))
    def set_multirx_freq(self, i, f): self.L.qo_ps_set_multirx_freq(self.h, int(i), int(f))
    def set_multirx_mode(self, i, m): self.L.qo_ps_set_multirx_mode(self.h, int(i), int(m))
    def set_multirx_count(self, n): self.L.qo_ps_set_multirx_count(self.h, int(n))
    def set_sub_rx1_output(self, on): self.L.qo_ps_set_sub_rx1_output(self.h, int(on))

    def multirx_samples(self, i, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        self.L.qo_ps_multirx_samples(self.h, int(i), x.ctypes.data, x.size)

    def set_key_state(self, key_down, cw_key_down, active_sidetone, is_fdx):
        self.L.qo_ps_set_key_state(self.h, int(key_down), int(cw_key_down), int(active_sidetone), int(is_fdx))

    def set_sidetone(self, volume, rit_freq, txrx_silence_ms=-1):
        self.L.qo_ps_set_sidetone(self.h, float(volume), int(rit_freq), int(txrx_silence_ms))

    def set_kill_audio(self, k): self.L.qo_ps_set_kill_audio(self.h, int(k))
    def invert_spectrum(self, inv): self.L.qo_ps_invert_spectrum(self.h, int(inv))
    def set_noise_blanker(self, level): self.L.qo_ps_set_noise_blanker(self.h, int(level))
    def set_auto_notch(self, on): self.L.qo_ps_set_auto_notch(self.h, int(on))
    def set_squelch(self, level): self.L.qo_ps_set_squelch(self.h, float(level))
    def set_ssb_squelch(self, enabled, level): self.L.qo_ps_set_ssb_squelch(self.h, int(enabled), int(level))
    def add_tone(self, freq): self.L.qo_ps_add_tone(self.h, int(freq))
    def measure_frequency(self, mode): return self.L.qo_ps_measure_frequency(self.h, int(mode))

    def set_graph(self, graph):
        self._hooks.append(graph)
        self.L.qo_ps_set_graph(self.h, graph.h)

    def set_wdsp(self, shim, channel):
        """shim: OracleWdspShim (its own callback is not used); channel: WdspChannel whose wo_fexchange0 serves the hand-off."""
        self._hooks += [shim, channel]
        fn = C.cast(self.L.wo_fexchange0, C.c_void_p)
        self.L.qo_ps_set_wdsp(self.h, shim.h, fn, channel.h)

    def sub_rx1_audio(self):
        n = self.L.qo_ps_sub_rx1_audio(self.h, None, 0)
        out = np.zeros(max(n, 1), dtype=np.complex128)
        self.L.qo_ps_sub_rx1_audio(self.h, out.ctypes.data, n)
        return out[:n]

    def squelch_flags(self): return self.L.qo_ps_squelch_flags(self.h)
    def restart_bank(self, bank): self.L.qo_ps_restart_bank(self.h, int(bank))

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        if x.size > 66000:          # SAMP_BUFFER_SIZE (quisk.h:15): the reference's -- and the restatement's -- work arrays end there
            raise ValueError("quisk_process_samples takes at most 66000 samples per call")
        # the block buffer is Quisk's: SAMP_BUFFER_SIZE complex samples whatever the block (quisk.c:2292 works in place in it).  With WDSP in
        # the audio path a call may come back one WDSP block longer than it went in (the shim's ring, quisk_wdsp.c:52-63), interpolated like the
        # rest: an array sized for the block alone was run over by short blocks at 96 ksps playback (round 6, found under AddressSanitizer)
        buf = np.zeros(max(66000, (max(x.size, 16) + 4096) * (self.ratio + 1)), dtype=np.complex128)
        buf[:x.size] = x
        n = self.L.qo_ps_process(self.h, buf.ctypes.data, x.size)
        if self.L.qo_ps_overrun(self.h):
            raise ValueError("Buffer2Chan (quisk.c:1577) holds 12000 audio samples per call: the reference overruns its arrays beyond that")
        return buf[:n].copy() if n > 0 else buf[:0].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.qo_ps_free(self.h)
            self.h = None


def get_filter(filtI, filtQ, data_width, fft_size):
    """get_filter (quisk.c:5481-5568): the Rx filter's response in dB as the "RX Filter" screen draws it, data_width values."""
    L = lib()
    fI = np.ascontiguousarray(filtI, dtype=np.float64)
    fQ = np.ascontiguousarray(filtQ, dtype=np.float64)
    out = np.zeros(data_width, dtype=np.float64)
    L.qo_get_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.qo_get_filter.restype = None
    L.qo_get_filter(fI.ctypes.data, fQ.ctypes.data, fI.size, data_width, fft_size, out.ctypes.data)
    return out


class OracleQuiskAgc:
    """process_agc (quisk.c:2162-2287) for one stream; process(x) returns the AGC'd block (the first call only initialises)."""

    def __init__(self, sample_rate=48000, max_out=0.7, release_time=1.0):
        L = lib()
        L.qo_agc_create.restype = C.c_void_p
        L.qo_agc_create.argtypes = [C.c_int, C.c_double, C.c_double]
        L.qo_agc_free.argtypes = [C.c_void_p]
        L.qo_agc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
        self.L = L
        self.h = L.qo_agc_create(sample_rate, max_out, release_time)

    def process(self, x, is_cpx, release_gain):
        buf = np.ascontiguousarray(x, dtype=np.complex128).copy()
        self.L.qo_agc_process(self.h, buf.ctypes.data, buf.size, int(is_cpx), float(release_gain))
        return buf

    def __del__(self):
        if getattr(self, "h", None):
            self.L.qo_agc_free(self.h)
            self.h = None


class OracleNoiseBlanker:
    """NoiseBlanker (quisk.c:680-784) for one stream: process(x) returns the blanked block, delayed by `delay` samples."""

    def __init__(self, sample_rate, level=1):
        L = lib()
        L.qo_nb_create.restype = C.c_void_p
        L.qo_nb_create.argtypes = [C.c_int]
        L.qo_nb_free.argtypes = [C.c_void_p]
        L.qo_nb_set_level.argtypes = [C.c_void_p, C.c_int]
        L.qo_nb_delay.argtypes = [C.c_void_p]
        L.qo_nb_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L = L
        self.h = L.qo_nb_create(sample_rate)
        self.delay = L.qo_nb_delay(self.h)
        L.qo_nb_set_level(self.h, level)

    def set_level(self, level):
        self.L.qo_nb_set_level(self.h, level)

    def process(self, x):
        buf = np.ascontiguousarray(x, dtype=np.complex128).copy()
        self.L.qo_nb_process(self.h, buf.ctypes.data, buf.size)
        return buf

    def __del__(self):
        if getattr(self, "h", None):
            self.L.qo_nb_free(self.h)
            self.h = None


class OracleAutoNotch:
    """dAutoNotch (quisk.c:786-963) for one real audio stream; process(x, sidetone, rate) returns the filtered block."""

    def __init__(self, on=True):
        L = lib()
        L.qo_notch_create.restype = C.c_void_p
        L.qo_notch_free.argtypes = [C.c_void_p]
        L.qo_notch_set.argtypes = [C.c_void_p, C.c_int]
        L.qo_notch_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        self.L = L
        self.h = L.qo_notch_create()
        L.qo_notch_set(self.h, int(on))

    def set(self, on):
        self.L.qo_notch_set(self.h, int(on))

    def process(self, x, sidetone, rate):
        buf = np.ascontiguousarray(x, dtype=np.float64).copy()
        self.L.qo_notch_process(self.h, buf.ctypes.data, buf.size, int(sidetone), int(rate))
        return buf

    def __del__(self):
        if getattr(self, "h", None):
            self.L.qo_notch_free(self.h)
            self.h = None


def watfall_row(db, red, green, blue, y_zero, y_scale, gain, width):
    """watfall_OnGraphData (quisk.c:5372-5421): dB row -> uint8 [width, 3]."""
    L = lib()
    L.qo_watfall_row.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
    L.qo_watfall_row.restype = None
    db = np.ascontiguousarray(db, dtype=np.float64)
    pal = [np.ascontiguousarray(a, dtype=np.uint8) for a in (red, green, blue)]
    rgb = np.zeros((width, 3), dtype=np.uint8)
    L.qo_watfall_row(db.ctypes.data, db.size, width, pal[0].ctypes.data, pal[1].ctypes.data, pal[2].ctypes.data, int(y_zero), int(y_scale),
                     float(gain), rgb.ctypes.data)
    return rgb


class OracleBandscope:
    """get_bandscope (quisk.c:4957-5011): block(x) per `size` real samples, get(clock, zoom, deltaf) -> (pixels, adc, count)."""

    def __init__(self, size, graph_width):
        L = lib()
        L.qo_bscope_create.restype = C.c_void_p
        L.qo_bscope_create.argtypes = [C.c_int, C.c_int]
        L.qo_bscope_free.argtypes = [C.c_void_p]
        L.qo_bscope_block.argtypes = [C.c_void_p, C.c_void_p]
        L.qo_bscope_get.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_double)]
        self.L, self.size, self.width = L, size, graph_width
        self.h = L.qo_bscope_create(size, graph_width)

    def block(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.size
        self.L.qo_bscope_block(self.h, x.ctypes.data)

    def get(self, clock, zoom=1.0, deltaf=0.0):
        pix = np.empty(self.width, dtype=np.float64)
        adc = C.c_double(0)
        n = self.L.qo_bscope_get(self.h, int(clock), zoom, deltaf, pix.ctypes.data, C.byref(adc))
        return None if n <= 0 else (pix, adc.value, n)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.qo_bscope_free(self.h)
            self.h = None


class OracleAnalyzer:
    """oracle/analyzer_oracle.c: one WDSP display (wdsp/analyzer.c), synchronous."""

    def __init__(self, max_size, max_stitch=1):
        L = lib()
        L.ao_create.restype = C.c_void_p
        L.ao_create.argtypes = [C.c_int, C.c_int]
        L.ao_destroy.argtypes = [C.c_void_p]
        L.ao_set_analyzer.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_double] + [C.c_int] * 2 + [C.c_double] * 2 + [C.c_int] * 3 + [C.c_double] * 2 + [C.c_int]
        L.ao_set_calibration.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ao_spectrum0.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ao_spectrum.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ao_get_pixels.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for n in ("ao_set_detector_mode", "ao_set_average_mode", "ao_set_num_average", "ao_set_norm_onehz"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_int]
            getattr(L, n).restype = None
        L.ao_set_av_backmult.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.ao_set_av_backmult.restype = None
        L.ao_set_sample_rate.argtypes = [C.c_void_p, C.c_int]
        L.ao_set_sample_rate.restype = None
        L.ao_get_enb.argtypes = [C.c_void_p]
        L.ao_get_enb.restype = C.c_double
        L.ao_frames.argtypes = [C.c_void_p]
        L.ao_frames.restype = C.c_long
        L.ao_window_ptr.argtypes = [C.c_void_p]
        L.ao_window_ptr.restype = C.POINTER(C.c_double)
        L.ao_cd_ptr.argtypes = [C.c_void_p]
        L.ao_cd_ptr.restype = C.POINTER(C.c_double)
        self.L = L
        self.h = L.ao_create(max_size, max_stitch)
        self.num_pixels = 0
        self.buff_size = 0
        self.size = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ao_destroy(self.h)
            self.h = None

    def SetAnalyzer(self, n_pixout, n_fft, typ, flp, sz, bf_sz, win_type, pi, ovrlp, clp, fscLin, fscHin, n_pix, n_stch, calset, fmin, fmax,
                    max_w):
        assert n_fft == 1
        self.L.ao_set_analyzer(self.h, n_pixout, typ, int(flp[0]), sz, bf_sz, win_type, pi, ovrlp, clp, fscLin, fscHin, n_pix, n_stch, calset,
                               fmin, fmax, max_w)
        self.num_pixels, self.buff_size, self.size = n_pix, bf_sz, sz

    def SetCalibration(self, set_num, table):
        t = np.ascontiguousarray(table, dtype=np.float64).copy()
        self.L.ao_set_calibration(self.h, set_num, t.shape[0], t.ctypes.data)

    def Spectrum0(self, run, ss, LO, pbuff):
        b = np.ascontiguousarray(pbuff, dtype=np.float64)
        assert b.size == 2 * self.buff_size
        self.L.ao_spectrum0(self.h, run, ss, b.ctypes.data)

    def Spectrum(self, ss, LO, pI, pQ):
        i = np.ascontiguousarray(pI, dtype=np.float32)
        q = np.ascontiguousarray(pQ, dtype=np.float32)
        self.L.ao_spectrum(self.h, ss, i.ctypes.data, q.ctypes.data)

    def GetPixels(self, pixout):
        pix = np.zeros(self.num_pixels, dtype=np.float32)
        flag = self.L.ao_get_pixels(self.h, pixout, pix.ctypes.data)
        return pix, flag

    def SnapSpectrum(self, ss):
        """Arms the snap (analyzer.c:1337-1346); the returned array (2 * size doubles) is filled by the next frame of sub-span ss."""
        buf = np.zeros(2 * self.size, dtype=np.float64)
        self.L.ao_snap.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.L.ao_snap.restype = None
        self.L.ao_snap(self.h, int(ss), buf.ctypes.data)
        self._snap = buf
        return buf

    def snap_taken(self):
        self.L.ao_snap_taken.argtypes = [C.c_void_p]
        return self.L.ao_snap_taken(self.h)

    def ResetPixelBuffers(self):
        self.L.ao_reset_pixel_buffers.argtypes = [C.c_void_p]
        self.L.ao_reset_pixel_buffers(self.h)

    def SetDisplayDetectorMode(self, pixout, mode): self.L.ao_set_detector_mode(self.h, pixout, mode)
    def SetDisplayAverageMode(self, pixout, mode): self.L.ao_set_average_mode(self.h, pixout, mode)
    def SetDisplayNumAverage(self, pixout, num): self.L.ao_set_num_average(self.h, pixout, num)
    def SetDisplayAvBackmult(self, pixout, mult): self.L.ao_set_av_backmult(self.h, pixout, mult)
    def SetDisplaySampleRate(self, rate): self.L.ao_set_sample_rate(self.h, rate)
    def SetDisplayNormOneHz(self, pixout, norm): self.L.ao_set_norm_onehz(self.h, pixout, norm)
    def GetDisplayENB(self): return self.L.ao_get_enb(self.h)
    def frames(self): return self.L.ao_frames(self.h)
    def window(self): return np.ctypeslib.as_array(self.L.ao_window_ptr(self.h), shape=(self.size,)).copy()
    def cd(self): return np.ctypeslib.as_array(self.L.ao_cd_ptr(self.h), shape=(self.num_pixels,)).copy()


def analyzer_detector(det_type, bins, num_pixels, pix_per_bin, bin_per_pix, inv_enb, fsclipL, fsclipH, det_offset, pixels=None):
    L = lib()
    L.ao_detector.argtypes = [C.c_int] * 3 + [C.c_double] * 2 + [C.c_void_p] * 2 + [C.c_double] * 4
    b = np.ascontiguousarray(bins, dtype=np.float64)
    out = np.zeros(num_pixels) if pixels is None else np.ascontiguousarray(pixels, dtype=np.float64).copy()
    L.ao_detector(det_type, b.size, num_pixels, pix_per_bin, bin_per_pix, b.ctypes.data, out.ctypes.data, inv_enb, fsclipL, fsclipH, det_offset)
    return out
