"""Python host side of the batched panadapter (include/quiskhip.h group 5); mirrors QS.get_graph()."""
import ctypes as C

import numpy as np

from .lib import load, check, QuiskHipError


class Panadapter:
    def __init__(self, nch, fft_size, data_width, sample_rate, device=0, stream=None):
        self._L = load()
        self._h = self._L.qh_pan_create(device, nch, fft_size, data_width, float(sample_rate), stream)
        if not self._h:
            raise QuiskHipError("qh_pan_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch, self.fft_size, self.data_width = nch, fft_size, data_width

    def set_smeter_band(self, ch, f_start, bandwidth):
        check(self._L.qh_pan_set_smeter_band(self._h, ch, float(f_start), float(bandwidth)))

    def feed_ptr(self, d_in, in_stride, n):
        check(self._L.qh_pan_feed(self._h, d_in, in_stride, n))

    def attach_fir(self, taps, decim):
        """A decimating FIR on the panadapter's own read (fft_size 16384, decimation 32, <= 1024 real taps)."""
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        check(self._L.qh_pan_attach_fir(self._h, taps.ctypes.data, taps.size, int(decim)))

    def feed_decimate_ptr(self, d_in, in_stride, n, d_out, out_stride):
        got = C.c_int(0)
        check(self._L.qh_pan_feed_decimate(self._h, d_in, in_stride, n, d_out, out_stride, C.byref(got)))
        return got.value

    def feed_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        if x.ndim != 2 or x.shape[0] != self.nch:
            raise ValueError("expected [nch, n] complex128")
        check(self._L.qh_pan_feed_host(self._h, x.ctypes.data, x.shape[1], x.shape[1]))

    def count(self):
        return self._L.qh_pan_count(self._h)

    def get_graph(self, zoom=1.0, deltaf=0.0):
        """Returns (pixels [nch, data_width] dB, smeter [nch] dB, count) or None when no FFT has completed."""
        pix = np.empty((self.nch, self.data_width), dtype=np.float64)
        sm = np.empty(self.nch, dtype=np.float64)
        cnt = C.c_int(0)
        check(self._L.qh_pan_graph(self._h, float(zoom), float(deltaf), pix.ctypes.data, sm.ctypes.data, C.byref(cnt)))
        if cnt.value <= 0:
            return None
        return pix, sm, cnt.value

    def waterfall_row(self, red, green, blue, y_zero, y_scale, gain, width, zoom=1.0, deltaf=0.0):
        """get_graph + watfall_OnGraphData (quisk.c:5372): (rgb uint8 [nch, width, 3], smeter, count) or None."""
        pal = [np.ascontiguousarray(a, dtype=np.uint8) for a in (red, green, blue)]
        if any(a.size != 256 for a in pal):
            raise ValueError("red, green, blue: 256 bytes each")
        rgb = np.zeros((self.nch, width, 3), dtype=np.uint8)
        sm = np.empty(self.nch, dtype=np.float64)
        cnt = C.c_int(0)
        check(self._L.qh_pan_waterfall(self._h, float(zoom), float(deltaf), pal[0].ctypes.data, pal[1].ctypes.data, pal[2].ctypes.data,
                                       int(y_zero), int(y_scale), float(gain), int(width), rgb.ctypes.data, sm.ctypes.data, C.byref(cnt)))
        if cnt.value <= 0:
            return None
        return rgb, sm, cnt.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_pan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Bandscope:
    """QS.get_bandscope (quisk.c:4957-5011) for `nch` ADC streams of real samples."""

    def __init__(self, nch, bandscope_size, graph_width, device=0, stream=None):
        self._L = load()
        self._h = self._L.qh_bscope_create(device, nch, bandscope_size, graph_width, stream)
        if not self._h:
            raise QuiskHipError("qh_bscope_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch, self.size, self.graph_width = nch, bandscope_size, graph_width

    def feed_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[0] != self.nch:
            raise ValueError("expected [nch, n] float64")
        check(self._L.qh_bscope_feed_host(self._h, x.ctypes.data, x.shape[1], x.shape[1]))

    def feed_ptr(self, d_in, in_stride, n):
        check(self._L.qh_bscope_feed(self._h, d_in, in_stride, n))

    def count(self):
        return self._L.qh_bscope_count(self._h)

    def get_bandscope(self, clock, zoom=1.0, deltaf=0.0):
        """(pixels [nch, graph_width] dB, adc_level [nch], count) or None when no block has completed."""
        pix = np.empty((self.nch, self.graph_width), dtype=np.float64)
        adc = np.empty(self.nch, dtype=np.float64)
        cnt = C.c_int(0)
        check(self._L.qh_bscope_graph(self._h, int(clock), float(zoom), float(deltaf), pix.ctypes.data, adc.ctypes.data, C.byref(cnt)))
        return None if cnt.value <= 0 else (pix, adc, cnt.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_bscope_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def waterfall_rows(db, red, green, blue, y_zero, y_scale, gain, width, device=0):
    """watfall_OnGraphData (quisk.c:5372-5421) for dB rows [nrows, ncols] -> uint8 [nrows, width, 3]."""
    from .lib import load
    db = np.ascontiguousarray(np.atleast_2d(db), dtype=np.float64)
    pal = [np.ascontiguousarray(a, dtype=np.uint8) for a in (red, green, blue)]
    rgb = np.zeros((db.shape[0], width, 3), dtype=np.uint8)
    check(load().qh_watfall_rows_host(device, db.ctypes.data, db.shape[0], db.shape[1], pal[0].ctypes.data, pal[1].ctypes.data,
                                      pal[2].ctypes.data, int(y_zero), int(y_scale), float(gain), int(width), rgb.ctypes.data))
    return rgb
