"""Python host side of the Quisk-native receiver bank (include/quiskhip.h group 6).

Method names follow the _quisk calls the GUI makes: set_tune (quisk.c:4702), set_filters (quisk.c:4551),
get_filter_rate (quisk.c:2787).  Rx filter taps come from quisk_amd.rxfilter.make_filter_coef, the restatement of
quisk.py's MakeFilterCoef.
"""
import ctypes as C

import numpy as np

from . import rxfilter
from .lib import load, check, QuiskHipError


_TABLE_KEYS = ("quiskFilt48dec24Coefs", "quiskFilt144D3Coefs", "quiskFilt240D5CoefsSharp", "quiskAudio24p4Coefs",
               "quiskAudio24p6Coefs", "quiskLpFilt48Coefs", "quiskAudioFmHpCoefs", "quiskFilt300D5Coefs",
               "quiskFilt53D1Coefs", "quiskFilt111D2Coefs", "quiskFilt133D2Coefs", "quiskFilt167D3Coefs",
               "quiskFilt185D3Coefs")


class _Tables(C.Structure):         # include/quiskhip.h: qh_qrx_tables
    _fields_ = [(k, C.c_void_p) for k in _TABLE_KEYS]


class QuiskRxBank:
    """`bandwidth` is what the GUI passes as set_filters' third argument (quisk.c:4581); it decides the filter
    rate of the DGT / FDV modes and whether DGT-IQ is filtered at all."""

    def __init__(self, nch, sample_rate, mode, bandwidth=2700, device=0, stream=None):
        self._L = load()
        t = rxfilter.coefficient_tables()
        self._tabs = [np.ascontiguousarray(t[k], dtype=np.float64) for k in _TABLE_KEYS]
        self._tstruct = _Tables(*[a.ctypes.data for a in self._tabs])
        self._h = self._L.qh_qrx_create_ex(device, nch, sample_rate, mode, bandwidth, C.byref(self._tstruct), stream)
        if not self._h:
            raise QuiskHipError("qh_qrx_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch, self.sample_rate, self.mode, self.bandwidth = nch, sample_rate, mode, bandwidth

    def get_decim_rate(self):
        return self._L.qh_qrx_decim_rate(self._h)

    def get_filter_rate(self):
        return self._L.qh_qrx_filter_rate(self._h)

    def set_tune(self, ch, rx_tune_freq):
        check(self._L.qh_qrx_set_tune(self._h, ch, int(rx_tune_freq)))

    def set_tune_all(self, rx_tune_freqs):
        """every receiver's set_tune (quisk.c:4702) in one launch per table: rx_tune_freqs[nch] Hz"""
        f = np.ascontiguousarray(rx_tune_freqs, dtype=np.int32)
        assert f.size == self.nch
        check(self._L.qh_qrx_set_tune_all(self._h, f.ctypes.data))

    def set_filters(self, ch, filtI, filtQ):
        fI = np.ascontiguousarray(filtI, dtype=np.float64)
        fQ = np.ascontiguousarray(filtQ, dtype=np.float64)
        if fI.size != fQ.size:
            raise ValueError("The size of filters I and Q must be equal")
        check(self._L.qh_qrx_set_filters(self._h, ch, fI.ctypes.data, fQ.ctypes.data, fI.size))

    def set_agc(self, on, release_gain=80.0):
        """process_agc on the output like quisk_process_samples; release_gain is QS.set_agc's argument (quisk.c:4543)."""
        check(self._L.qh_qrx_set_agc(self._h, 1 if on else 0, float(release_gain)))

    def set_auto_notch(self, on, rit_freq=0):
        """QS.set_auto_notch (quisk.c:4596): dAutoNotch on the demodulated audio; rit_freq keeps CW's sidetone."""
        check(self._L.qh_qrx_set_auto_notch(self._h, 1 if on else 0, int(rit_freq)))

    def set_noise_blanker(self, level):
        """QS.set_noise_blanker (quisk.c:4605): NoiseBlanker on the raw samples ahead of the tune; 0 = off."""
        check(self._L.qh_qrx_set_noise_blanker(self._h, int(level)))

    def set_ssb_squelch(self, enabled, level):
        """QS.set_ssb_squelch (quisk.c:4729): CW / SSB / AM."""
        check(self._L.qh_qrx_set_ssb_squelch(self._h, 1 if enabled else 0, int(level)))

    def set_squelch(self, ch, level):
        """FM squelch threshold, QS.set_squelch (quisk.c:4721)."""
        check(self._L.qh_qrx_set_squelch(self._h, ch, float(level)))

    def out_count(self, n_in):
        return self._L.qh_qrx_out_count(self._h, n_in)

    def process_ptr(self, d_in, in_stride, n_in, d_out, out_stride):
        n = C.c_int(0)
        check(self._L.qh_qrx_process(self._h, d_in, in_stride, n_in, d_out, out_stride, C.byref(n)))
        return n.value

    def process_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        if x.ndim != 2 or x.shape[0] != self.nch:
            raise ValueError("expected [nch, n] complex128")
        cap = max(self.out_count(x.shape[1]), 1)
        out = np.empty((self.nch, cap), dtype=np.complex128)
        n = C.c_int(0)
        check(self._L.qh_qrx_process_host(self._h, x.ctypes.data, x.shape[1], x.shape[1], out.ctypes.data, cap, C.byref(n)))
        return out[:, :n.value].copy()

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_qrx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QuiskProcessBank:
    """quisk_process_samples (quisk.c:2289-2742) for a bank of receivers (include/quiskhip.h group 9b): test tone / inversion,
    NoiseBlanker, the panadapter's feed, tune + decimate + demodulate, cFracDecim, interpolation to the playback rate, process_agc
    (always on, as in the reference) and the squelches.  Method names follow the _quisk calls."""

    def __init__(self, nch, sample_rate, mode, bandwidth=2700, playback_rate=48000, fft_size=0, data_width=0, device=0, stream=None):
        self._L = load()
        t = rxfilter.coefficient_tables()
        self._tabs = [np.ascontiguousarray(t[k], dtype=np.float64) for k in _TABLE_KEYS]
        self._tstruct = _Tables(*[a.ctypes.data for a in self._tabs])
        self._h = self._L.qh_qps_create(device, nch, sample_rate, playback_rate, mode, bandwidth, C.byref(self._tstruct), fft_size, data_width, stream)
        if not self._h:
            raise QuiskHipError("qh_qps_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch, self.sample_rate, self.mode, self.data_width = nch, sample_rate, mode, data_width

    def get_filter_rate(self):
        return self._L.qh_qps_filter_rate(self._h)

    def get_decim_rate(self):
        return self._L.qh_qps_decim_rate(self._h)

    def set_tune(self, ch, rx_tune_freq):
        check(self._L.qh_qps_set_tune(self._h, ch, int(rx_tune_freq)))

    def set_tune_all(self, rx_tune_freqs):
        """every receiver's set_tune in one launch per table: rx_tune_freqs[nch] Hz"""
        f = np.ascontiguousarray(rx_tune_freqs, dtype=np.int32)
        assert f.size == self.nch
        check(self._L.qh_qps_set_tune_all(self._h, f.ctypes.data))

    def set_filters(self, ch, filtI, filtQ):
        fI = np.ascontiguousarray(filtI, dtype=np.float64)
        fQ = np.ascontiguousarray(filtQ, dtype=np.float64)
        if fI.size != fQ.size:
            raise ValueError("The size of filters I and Q must be equal")
        check(self._L.qh_qps_set_filters(self._h, ch, fI.ctypes.data, fQ.ctypes.data, fI.size))

    def set_agc(self, level):
        check(self._L.qh_qps_set_agc(self._h, float(level)))

    def set_noise_blanker(self, level):
        check(self._L.qh_qps_set_noise_blanker(self._h, int(level)))

    def set_auto_notch(self, on, rit_freq=0):
        check(self._L.qh_qps_set_auto_notch(self._h, 1 if on else 0, int(rit_freq)))

    def invert_spectrum(self, invert):
        check(self._L.qh_qps_invert_spectrum(self._h, 1 if invert else 0))

    def set_kill_audio(self, kill):
        check(self._L.qh_qps_set_kill_audio(self._h, 1 if kill else 0))

    def add_tone(self, freq):
        check(self._L.qh_qps_add_tone(self._h, int(freq)))

    def set_squelch(self, ch, level):
        check(self._L.qh_qps_set_squelch(self._h, ch, float(level)))

    def set_ssb_squelch(self, enabled, level):
        check(self._L.qh_qps_set_ssb_squelch(self._h, 1 if enabled else 0, int(level)))

    def set_pieces(self, pieces):
        check(self._L.qh_qps_set_pieces(self._h, int(pieces)))

    def set_pipelined(self, on):
        """a call returns with its AGC still running; outputs are complete after synchronize()"""
        check(self._L.qh_qps_set_pipelined(self._h, 1 if on else 0))

    def out_capacity(self, n_in):
        return self._L.qh_qps_out_capacity(self._h, n_in)

    def process_ptr(self, d_in, in_stride, n, d_out, out_stride):
        got = C.c_int(0)
        check(self._L.qh_qps_process(self._h, d_in, in_stride, n, d_out, out_stride, C.byref(got)))
        return got.value

    def process_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        if x.ndim != 2 or x.shape[0] != self.nch:
            raise ValueError("expected [nch, n] complex128")
        cap = max(self.out_capacity(x.shape[1]), 1)
        out = np.empty((self.nch, cap), dtype=np.complex128)
        got = C.c_int(0)
        check(self._L.qh_qps_process_host(self._h, x.ctypes.data, x.shape[1], x.shape[1], out.ctypes.data, cap, C.byref(got)))
        return out[:, :got.value].copy()

    def squelch_flags(self):
        f = np.zeros(self.nch, dtype=np.int32)
        check(self._L.qh_qps_squelch_flags(self._h, f.ctypes.data))
        return f

    def get_graph(self, zoom=1.0, deltaf=0.0):
        pix = np.empty((self.nch, self.data_width), dtype=np.float64)
        sm = np.empty(self.nch, dtype=np.float64)
        cnt = C.c_int(0)
        check(self._L.qh_qps_get_graph(self._h, float(zoom), float(deltaf), pix.ctypes.data, sm.ctypes.data, C.byref(cnt)))
        return None if cnt.value <= 0 else (pix, sm, cnt.value)

    def synchronize(self):
        check(self._L.qh_qps_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_qps_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QuiskAgc:
    """process_agc (quisk.c:2162-2287) for `nch` streams; the first process call only initialises, as in the reference."""

    def __init__(self, nch, sample_rate=48000, max_out=0.7, release_time=1.0, is_cpx=False, device=0, stream=None):
        self._L = load()
        self._h = self._L.qh_qagc_create(device, nch, sample_rate, max_out, release_time, 1 if is_cpx else 0, stream)
        if not self._h:
            raise QuiskHipError("qh_qagc_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch = nch

    def set_agc(self, ch, release_gain):
        check(self._L.qh_qagc_set_gain(self._h, ch, float(release_gain)))

    def process_ptr(self, d_buf, stride, n):
        check(self._L.qh_qagc_process(self._h, d_buf, stride, n))

    def process2_ptr(self, d_src, src_stride, d_dst, dst_stride, n):
        """from one device buffer into another"""
        check(self._L.qh_qagc_process2(self._h, d_src, src_stride, d_dst, dst_stride, n))

    def debug_form(self, form):
        """diagnostics: 0 = the two regimes as instruction chains (default), 1 = the whole machine sample by sample (bit-identical)"""
        check(self._L.qh_qagc_debug_form(self._h, int(form)))

    def process_host(self, x):
        buf = np.ascontiguousarray(x, dtype=np.complex128).copy()
        if buf.ndim != 2 or buf.shape[0] != self.nch:
            raise ValueError("expected [nch, n] complex128")
        check(self._L.qh_qagc_process_host(self._h, buf.ctypes.data, buf.shape[1], buf.shape[1]))
        return buf

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_qagc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NoiseBlanker:
    """NoiseBlanker (quisk.c:680-784) for `nch` streams at the receiver's input rate; level 0..3 as set_noise_blanker."""

    def __init__(self, nch, sample_rate, level=1, device=0, stream=None):
        self._L = load()
        self._h = self._L.qh_nb_create(device, nch, sample_rate, stream)
        if not self._h:
            raise QuiskHipError("qh_nb_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch = nch
        self.delay = self._L.qh_nb_delay(self._h)
        self.set_level(level)

    def set_level(self, level):
        check(self._L.qh_nb_set_level(self._h, int(level)))

    def reset(self):
        check(self._L.qh_nb_reset(self._h))

    def process_ptr(self, d_in, in_stride, d_out, out_stride, n):
        """Device pointers, strides in complex samples; d_in != d_out.  Asynchronous on the blanker's stream."""
        check(self._L.qh_nb_process(self._h, d_in, in_stride, d_out, out_stride, n))

    def process_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        if x.ndim != 2 or x.shape[0] != self.nch:
            raise ValueError("expected [nch, n] complex128")
        out = np.empty_like(x)
        check(self._L.qh_nb_process_host(self._h, x.ctypes.data, x.shape[1], out.ctypes.data, x.shape[1], x.shape[1]))
        return out

    def synchronize(self):
        check(self._L.qh_nb_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_nb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
