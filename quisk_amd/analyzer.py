"""WDSP's display engine (wdsp/analyzer.c) on the GPU: a bank of displays that share one configuration.

Method names and argument lists are WDSP's (SetAnalyzer, SetDisplayDetectorMode, ..., analyzer.c:999-1017,1582-1676);
samples are complex (I + jQ).  The WDSP-named C entry points themselves (XCreateAnalyzer, Spectrum0, GetPixels ...) are
exported by the library for C callers; `WdspDisplay` below drives them the way a ctypes binding of libwdsp would."""
import ctypes as C

import numpy as np

from .lib import load, QuiskHipError


def _check(L, rc):
    if rc != 0:
        raise QuiskHipError(L.qh_last_error().decode(errors="replace"))


class AnalyzerBank:
    def __init__(self, ndisp, max_size, max_stitch=1, device=0, stream=None):
        self._L = load()
        self._h = self._L.qh_ana_create(device, ndisp, max_size, max_stitch, stream)
        if not self._h:
            raise QuiskHipError("qh_ana_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.ndisp = ndisp
        self.num_pixels = 0
        self.buff_size = 0

    def SetAnalyzer(self, n_pixout, n_fft, typ, flp, sz, bf_sz, win_type, pi, ovrlp, clp, fscLin, fscHin, n_pix, n_stch, calset, fmin, fmax,
                    max_w):
        f = (C.c_int * max(1, len(flp)))(*[int(v) for v in flp])
        _check(self._L, self._L.qh_ana_set_analyzer(self._h, n_pixout, n_fft, typ, f, sz, bf_sz, win_type, pi, ovrlp, clp, fscLin, fscHin, n_pix,
                                                     n_stch, calset, fmin, fmax, max_w))
        self.num_pixels, self.buff_size = n_pix, bf_sz

    def SetCalibration(self, set_num, table):
        t = np.ascontiguousarray(table, dtype=np.float64)
        _check(self._L, self._L.qh_ana_set_calibration(self._h, set_num, t.shape[0], t.ctypes.data))

    def SetDisplayDetectorMode(self, pixout, mode): _check(self._L, self._L.qh_ana_set_detector_mode(self._h, pixout, mode))
    def SetDisplayAverageMode(self, pixout, mode): _check(self._L, self._L.qh_ana_set_average_mode(self._h, pixout, mode))
    def SetDisplayNumAverage(self, pixout, num): _check(self._L, self._L.qh_ana_set_num_average(self._h, pixout, num))
    def SetDisplayAvBackmult(self, pixout, mult): _check(self._L, self._L.qh_ana_set_av_backmult(self._h, pixout, mult))
    def SetDisplaySampleRate(self, rate): _check(self._L, self._L.qh_ana_set_sample_rate(self._h, rate))
    def SetDisplayNormOneHz(self, pixout, norm): _check(self._L, self._L.qh_ana_set_norm_onehz(self._h, pixout, norm))
    def GetDisplayENB(self): return self._L.qh_ana_get_enb(self._h)
    def ResetPixelBuffers(self): _check(self._L, self._L.qh_ana_reset_pixel_buffers(self._h))

    def SnapSpectrum_arm(self, disp, ss):
        """SnapSpectrum (analyzer.c:1337) in two steps: ask for the next frame's transform of (display, sub-span) ..."""
        _check(self._L, self._L.qh_ana_snap_arm(self._h, int(disp), int(ss)))

    def SnapSpectrum_take(self, size):
        """... and take it once a feed call has made that frame: `size` complex values, fft-shifted, or None."""
        buf = np.zeros(2 * size, dtype=np.float64)
        flag = C.c_int(0)
        self._L.qh_ana_snap_take.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        _check(self._L, self._L.qh_ana_snap_take(self._h, buf.ctypes.data, C.byref(flag)))
        return buf.view(np.complex128) if flag.value else None

    def feed_ptr(self, ss, d_iq, disp_stride, n):
        """n complex128 samples per display on the device; returns the number of pixel rows the call published."""
        frames = C.c_int(0)
        _check(self._L, self._L.qh_ana_feed(self._h, ss, d_iq, disp_stride, n, C.byref(frames)))
        return frames.value

    def feed_host(self, ss, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        if x.ndim != 2 or x.shape[0] != self.ndisp:
            raise ValueError("expected [ndisp, n] complex128")
        frames = C.c_int(0)
        _check(self._L, self._L.qh_ana_feed_host(self._h, ss, x.ctypes.data, x.shape[1], x.shape[1], 0, C.byref(frames)))
        return frames.value

    def rows_host(self, pixout, max_frames=4096):
        """[ndisp, frames, num_pixels] float32: every row of the last feed call."""
        frames = C.c_int(0)
        dptr, npx = C.c_void_p(), C.c_int(0)
        _check(self._L, self._L.qh_ana_rows(self._h, pixout, C.byref(dptr), C.byref(frames), C.byref(npx)))
        out = np.empty((self.ndisp, frames.value, self.num_pixels), dtype=np.float32)
        _check(self._L, self._L.qh_ana_rows_host(self._h, pixout, out.ctypes.data, max(frames.value, 1), C.byref(frames)))
        return out

    def GetPixels(self, disp, pixout):
        pix = np.zeros(self.num_pixels, dtype=np.float32)
        flag = C.c_int(0)
        _check(self._L, self._L.qh_ana_get_pixels(self._h, disp, pixout, pix.ctypes.data, C.byref(flag)))
        return pix, flag.value

    def frames(self):
        return self._L.qh_ana_frames(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_ana_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WdspDisplay:
    """One display through WDSP's own exported names, as a ctypes binding of libwdsp.so would call them."""

    def __init__(self, disp, max_size, max_stitch=1):
        self._L = load()
        self.disp = disp
        ok = C.c_int(-1)
        self._L.XCreateAnalyzer(disp, C.byref(ok), max_size, 1, max_stitch, b"")
        if ok.value != 0:
            raise QuiskHipError("XCreateAnalyzer failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.num_pixels = 0
        self.buff_size = 0

    def SetAnalyzer(self, n_pixout, n_fft, typ, flp, sz, bf_sz, win_type, pi, ovrlp, clp, fscLin, fscHin, n_pix, n_stch, calset, fmin, fmax,
                    max_w):
        f = (C.c_int * max(1, len(flp)))(*[int(v) for v in flp])
        self._L.SetAnalyzer(self.disp, n_pixout, n_fft, typ, f, sz, bf_sz, win_type, pi, ovrlp, clp, fscLin, fscHin, n_pix, n_stch, calset, fmin,
                            fmax, max_w)
        self.num_pixels, self.buff_size = n_pix, bf_sz

    def Spectrum0(self, run, ss, LO, pbuff):
        b = np.ascontiguousarray(pbuff, dtype=np.float64)
        assert b.size == 2 * self.buff_size
        self._L.Spectrum0(run, self.disp, ss, LO, b.ctypes.data)

    def Spectrum2(self, run, ss, LO, pbuff):
        b = np.ascontiguousarray(pbuff, dtype=np.float32)
        assert b.size == 2 * self.buff_size
        self._L.Spectrum2(run, self.disp, ss, LO, b.ctypes.data)

    def Spectrum(self, ss, LO, pI, pQ):
        i = np.ascontiguousarray(pI, dtype=np.float32)
        q = np.ascontiguousarray(pQ, dtype=np.float32)
        self._L.Spectrum(self.disp, ss, LO, i.ctypes.data, q.ctypes.data)

    def OpenCloseBuffer(self, ss, LO, pI, pQ):
        ip, qp = C.c_void_p(), C.c_void_p()
        self._L.OpenBuffer(self.disp, ss, LO, C.byref(ip), C.byref(qp))
        C.memmove(ip, np.ascontiguousarray(pI, dtype=np.float32).ctypes.data, 4 * self.buff_size)
        C.memmove(qp, np.ascontiguousarray(pQ, dtype=np.float32).ctypes.data, 4 * self.buff_size)
        self._L.CloseBuffer(self.disp, ss, LO)

    def GetPixels(self, pixout):
        pix = np.zeros(self.num_pixels, dtype=np.float32)
        flag = C.c_int(0)
        self._L.GetPixels(self.disp, pixout, pix.ctypes.data, C.byref(flag))
        return pix, flag.value

    def SetCalibration(self, set_num, table):
        t = np.ascontiguousarray(table, dtype=np.float64)
        self._L.SetCalibration(self.disp, set_num, t.shape[0], t.ctypes.data)

    def SetDisplayDetectorMode(self, pixout, mode): self._L.SetDisplayDetectorMode(self.disp, pixout, mode)
    def SetDisplayAverageMode(self, pixout, mode): self._L.SetDisplayAverageMode(self.disp, pixout, mode)
    def SetDisplayNumAverage(self, pixout, num): self._L.SetDisplayNumAverage(self.disp, pixout, num)
    def SetDisplayAvBackmult(self, pixout, mult): self._L.SetDisplayAvBackmult(self.disp, pixout, mult)
    def SetDisplaySampleRate(self, rate): self._L.SetDisplaySampleRate(self.disp, rate)
    def SetDisplayNormOneHz(self, pixout, norm): self._L.SetDisplayNormOneHz(self.disp, pixout, norm)
    def GetDisplayENB(self): return self._L.GetDisplayENB(self.disp)
    def ResetPixelBuffers(self): self._L.ResetPixelBuffers(self.disp)

    def close(self):
        if self.disp is not None:
            self._L.DestroyAnalyzer(self.disp)
            self.disp = None
