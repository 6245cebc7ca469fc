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

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_pan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
