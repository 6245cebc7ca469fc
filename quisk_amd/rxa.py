"""Python host side of the batched RXA engine (include/quiskhip.h group 1).

Method names are the WDSP export names the reference binds through ctypes
(quisk_wdsp.py:79-91, quisk.py:6017-6052); the first argument is the channel index inside the
batch, or -1 for all channels.
"""
import ctypes as C

import numpy as np

from .lib import load, check, QuiskHipError

_SETTERS = ("SetRXAMode", "RXASetNC", "SetRXAShiftRun", "RXANBPSetRun", "SetRXABandpassRun", "SetRXAAGCMode",
            "SetRXAPanelSelect", "SetRXAPanelCopy", "SetRXAShiftFreq", "SetRXAAGCFixed", "SetRXAPanelGain1",
            "RXASetPassband", "RXANBPSetFreqs", "SetRXABandpassFreqs", "SetRXAPanelGain2", "SetRXAAMDSBMode",
            "SetRXAAMDFadeLevel", "SetRXAFMDeviation", "SetRXACTCSSFreq", "SetRXACTCSSRun", "SetRXAAGCAttack",
            "SetRXAAGCDecay", "SetRXAAGCHang", "SetRXAAGCTop", "SetRXAAGCSlope", "SetRXAAGCHangThreshold", "RXASetMP",
            "SetRXAAMDRun", "RXANBPSetNotchesRun", "RXANBPSetWindow", "RXANBPSetAutoIncrease", "RXANBPSetTuneFrequency",
            "RXANBPSetShiftFrequency", "SetRXAFMLimRun", "SetRXAFMLimGain",
            "SetRXASNBARun", "SetRXASNBAOutputBandwidth", "SetRXASNBAasize", "SetRXASNBAnpasses", "SetRXASNBAk1", "SetRXASNBAk2", "SetRXASNBAbridge", "SetRXASNBApresamps", "SetRXASNBApostsamps", "SetRXASNBApmultmin", "SetRXASNBAovrlp", "SetRXAEMNRRun", "SetRXAEMNRgainMethod", "SetRXAEMNRnpeMethod", "SetRXAEMNRaeRun", "SetRXAEMNRPosition", "SetRXAEMNRaeZetaThresh", "SetRXAEMNRaePsi", "SetRXAEMNRtrainZetaThresh", "SetRXAEMNRtrainT2",
            "SetRXAAMSQRun", "SetRXAAMSQThreshold", "SetRXAAMSQMaxTail", "SetRXAANFRun", "SetRXAANFTaps", "SetRXAANFDelay", "SetRXAANFPosition", "SetRXAANFGain", "SetRXAANFLeakage", "SetRXAANFVals",
            "SetRXAANRRun", "SetRXAANRTaps", "SetRXAANRDelay", "SetRXAANRPosition", "SetRXAANRGain", "SetRXAANRLeakage", "SetRXAANRVals")


class AudioFormat(C.Structure):
    """qh_audio_format (include/quiskhip.h 7b): the narrowing of Quisk's sound back ends (sound_alsa.c:344-390,
    sound_pulseaudio.c:684-695).  kind: "i16", "i24", "i32", "f32"."""
    _fields_ = [("kind", C.c_int), ("num_channels", C.c_int), ("channel_I", C.c_int), ("channel_Q", C.c_int),
                ("volume", C.c_double), ("prescale", C.c_double)]
    KINDS = {"i16": 1, "i24": 2, "i32": 3, "f32": 4}
    NP = {1: np.int16, 2: np.uint8, 3: np.int32, 4: np.float32}

    def __init__(self, kind="i16", volume=1.0, prescale=1.0, num_channels=2, channel_I=0, channel_Q=1):
        super().__init__(self.KINDS[kind], num_channels, channel_I, channel_Q, volume, prescale)

    @property
    def frame_bytes(self):
        return self.num_channels * (2 if self.kind == 1 else 3 if self.kind == 2 else 4)


class RxaEngine:
    """nch independent WDSP-RXA receiver channels on one MI355X."""

    def __init__(self, nch, dsp_size=256, in_rate=192000, dsp_rate=48000, out_rate=48000, device=0, stream=None):
        self._L = load()
        self._h = self._L.qh_rxa_create(device, nch, dsp_size, in_rate, dsp_rate, out_rate, stream)
        if not self._h:
            raise QuiskHipError("qh_rxa_create failed: %s" % self._L.qh_last_error().decode(errors="replace"))
        self.nch = nch
        self.dsp_insize = self._L.qh_rxa_dsp_insize(self._h)
        self.dsp_outsize = self._L.qh_rxa_dsp_outsize(self._h)
        self._emnr = None

    def load_emnr_tables(self, path=None):
        """EMNR's gain tables (WDSP's `calculus` / `zetaHat.bin` data, extracted by tools/extract_wdsp_emnr_tables.py)."""
        import os
        path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "wdsp_emnr_tables.npz")
        z = np.load(path)
        t = [np.ascontiguousarray(z["GG"], dtype=np.float64), np.ascontiguousarray(z["GGS"], dtype=np.float64),
             np.ascontiguousarray(z["zeta_hat"], dtype=np.float64), np.ascontiguousarray(z["zeta_valid"], dtype=np.int32)]
        r = [float(v) for v in z["zeta_range"]]
        check(self._L.qh_rxa_SetEMNRTables(self._h, t[0].ctypes.data, t[1].ctypes.data, t[2].ctypes.data, t[3].ctypes.data, *r))
        self._emnr = t

    def __getattr__(self, name):
        if name in _SETTERS:
            f = getattr(self._L, "qh_rxa_" + name)

            def call(ch, *args):
                check(f(self._h, ch, *args))
            return call
        raise AttributeError(name)

    # the notch database (wdsp/nbp.c:358-469): results are the reference's return values (0 / -1)
    def RXANBPAddNotch(self, ch, notch, fcenter, fwidth, active):
        r = C.c_int(-1)
        check(self._L.qh_rxa_RXANBPAddNotch(self._h, ch, notch, fcenter, fwidth, active, C.byref(r)))
        return r.value

    def RXANBPEditNotch(self, ch, notch, fcenter, fwidth, active):
        r = C.c_int(-1)
        check(self._L.qh_rxa_RXANBPEditNotch(self._h, ch, notch, fcenter, fwidth, active, C.byref(r)))
        return r.value

    def RXANBPDeleteNotch(self, ch, notch):
        r = C.c_int(-1)
        check(self._L.qh_rxa_RXANBPDeleteNotch(self._h, ch, notch, C.byref(r)))
        return r.value

    def RXANBPGetNotch(self, ch, notch):
        f, w, a, r = C.c_double(0), C.c_double(0), C.c_int(0), C.c_int(-1)
        check(self._L.qh_rxa_RXANBPGetNotch(self._h, ch, notch, C.byref(f), C.byref(w), C.byref(a), C.byref(r)))
        return r.value, f.value, w.value, a.value

    def RXANBPGetNumNotches(self, ch):
        n = C.c_int(0)
        check(self._L.qh_rxa_RXANBPGetNumNotches(self._h, ch, C.byref(n)))
        return n.value

    def RXANBPGetMinNotchWidth(self, ch):
        w = C.c_double(0)
        check(self._L.qh_rxa_RXANBPGetMinNotchWidth(self._h, ch, C.byref(w)))
        return w.value

    def process_ptr(self, d_in, in_stride, d_out, out_stride, nblk):
        """Device pointers (ints), strides in complex samples.  Asynchronous on the engine's stream."""
        check(self._L.qh_rxa_process(self._h, d_in, in_stride, d_out, out_stride, nblk))

    def process_host(self, x):
        """x: complex128 [nch, nblk*dsp_insize] on the host -> complex128 [nch, nblk*dsp_outsize]."""
        x = np.ascontiguousarray(x, dtype=np.complex128)
        if x.ndim != 2 or x.shape[0] != self.nch or x.shape[1] % self.dsp_insize:
            raise ValueError("expected [nch, k*dsp_insize] complex128")
        nblk = x.shape[1] // self.dsp_insize
        out = np.empty((self.nch, nblk * self.dsp_outsize), dtype=np.complex128)
        check(self._L.qh_rxa_process_host(self._h, x.ctypes.data, x.shape[1], out.ctypes.data, out.shape[1], nblk))
        return out

    def process_audio_ptr(self, d_in, in_stride, d_out, out_stride_bytes, nblk, fmt):
        """Like process_ptr, the output as audio frames (AudioFormat) narrowed in the last kernel's store."""
        check(self._L.qh_rxa_process_audio(self._h, d_in, in_stride, d_out, out_stride_bytes, nblk, C.byref(fmt)))

    def process_packed_ptr(self, d_src, src_bytes, fmt, chan_stride, d_out, out_stride, nblk):
        """Wire-format input (quisk_amd.ingest.IqFormat) decoded inside the first kernel's load."""
        check(self._L.qh_rxa_process_packed(self._h, d_src, src_bytes, C.byref(fmt), chan_stride, d_out, out_stride, nblk))

    def process_packed_host(self, buf, fmt, chan_stride, nblk):
        raw = np.frombuffer(bytes(buf), dtype=np.uint8)
        out = np.empty((self.nch, nblk * self.dsp_outsize), dtype=np.complex128)
        check(self._L.qh_rxa_process_packed_host(self._h, raw.ctypes.data, raw.size, C.byref(fmt), chan_stride, out.ctypes.data,
                                                 out.shape[1], nblk))
        return out

    def enable_meters(self, on=True):
        check(self._L.qh_rxa_enable_meters(self._h, 1 if on else 0))

    def GetRXAMeter(self, ch, mt):
        v = C.c_double(0)
        check(self._L.qh_rxa_GetRXAMeter(self._h, ch, mt, C.byref(v)))
        return v.value

    def pll_repairs(self):
        return self._L.qh_rxa_pll_repairs(self._h)

    def debug_agc(self, form):
        """diagnostics: 0 = time tiles for long calls (default), 1 = the sample-by-sample form of the wcpAGC loop, 2 = 64 samples per step"""
        check(self._L.qh_rxa_debug_agc(self._h, int(form)))

    def agc_repairs(self):
        return self._L.qh_rxa_agc_repairs(self._h)

    def agc_segments_rerun(self):
        return self._L.qh_rxa_agc_segments_rerun(self._h)

    def agc_tiled_channels(self):
        return self._L.qh_rxa_agc_tiled_channels(self._h)

    def synchronize(self):
        check(self._L.qh_rxa_synchronize(self._h))

    def flush(self):
        """flush_rxa (wdsp/RXA.c:537-559): every stage's state back to its start"""
        check(self._L.qh_rxa_flush(self._h))

    def set_graph_replay(self, on=True):
        """Block-at-a-time callers: replay the launch sequence of process_ptr from hipGraphs while nothing changes."""
        check(self._L.qh_rxa_set_graph_replay(self._h, 1 if on else 0))

    def set_band_tile(self, nfft=0):
        """Tile of the band-pass stages: 4096 (0 = default) or 8192 (two lane groups per tile; measured slower)."""
        check(self._L.qh_rxa_set_band_tile(self._h, int(nfft)))

    def band_tile(self):
        return self._L.qh_rxa_band_tile(self._h)

    def graph_launches(self):
        return self._L.qh_rxa_graph_launches(self._h)

    def enable_timing(self, on=True):
        check(self._L.qh_rxa_enable_timing(self._h, 1 if on else 0))

    def timing_ms(self):
        buf = (C.c_double * 3)()
        n = self._L.qh_rxa_timing(self._h, buf, 3)
        return [buf[k] for k in range(n)]

    def device_bytes(self):
        return self._L.qh_rxa_device_bytes(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.qh_rxa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
