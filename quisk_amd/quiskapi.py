"""The Quisk native block API for one receiver (include/quiskhip.h group 9), named like the `_quisk` calls the GUI
makes (QS.set_tune, QS.set_rx_mode, QS.set_filters, QS.set_agc, QS.get_filter_rate, QS.get_graph) around
quisk_process_samples (quisk.c:2289)."""
import ctypes as C

import numpy as np

from . import rxfilter
from .lib import load, check
from .qrx import _TABLE_KEYS, _Tables

_keep = {}


def open(sample_rate, fft_size=0, data_width=0, playback_rate=48000):
    """record_app's fft_size / data_width and open_sound's sample and playback rates (quisk.c:5946,4106)."""
    L = load()
    t = rxfilter.coefficient_tables()
    tabs = [np.ascontiguousarray(t[k], dtype=np.float64) for k in _TABLE_KEYS]
    st = _Tables(*[a.ctypes.data for a in tabs])
    _keep["tabs"] = (tabs, st)
    _keep["data_width"] = data_width
    _keep["ratio"] = max(1, playback_rate // 48000)
    check(L.qh_quisk_open(sample_rate, playback_rate, C.byref(st), fft_size, data_width))


def close():
    load().qh_quisk_close()


def set_tune(rx_tune_freq):
    load().qh_quisk_set_tune(int(rx_tune_freq))


def set_rx_mode(mode):
    load().qh_quisk_set_rx_mode(int(mode))


def set_filters(filtI, filtQ, bandwidth, nFilter=0):
    """QS.set_filters(I, Q, bandwidth, start_offset, nFilter), quisk.c:4551."""
    fI = np.ascontiguousarray(filtI, dtype=np.float64)
    fQ = np.ascontiguousarray(filtQ, dtype=np.float64)
    if fI.size != fQ.size:
        raise ValueError("The size of filters I and Q must be equal")
    check(load().qh_quisk_set_filters_n(fI.ctypes.data, fQ.ctypes.data, fI.size, int(bandwidth), int(nFilter)))


def set_tune2(rx_tune_freq, tx_tune_freq):
    """QS.set_tune(rx, tx), quisk.c:4702."""
    load().qh_quisk_set_tune(int(rx_tune_freq))
    load().qh_quisk_set_tx_tune(int(tx_tune_freq))


def set_split_rxtx(split):
    load().qh_quisk_set_split_rxtx(int(split))


def set_multirx_play_channel(ch):
    load().qh_quisk_set_multirx_play_channel(int(ch))


def set_multirx_play_method(method):
    load().qh_quisk_set_multirx_play_method(int(method))


def set_multirx_freq(index, freq):
    load().qh_quisk_set_multirx_freq(int(index), int(freq))


def set_multirx_mode(index, mode):
    load().qh_quisk_set_multirx_mode(int(index), int(mode))


def multirx_samples(index, x):
    """The played sub-receiver's block (multirx_cSamples[index]) for the next process_samples call."""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    check(load().qh_quisk_multirx_samples(int(index), x.ctypes.data, x.size))


def set_filters2(filtI, filtQ, bandwidth):
    """QS.set_filters(I, Q, bandwidth, offset, 1): the filter of the played sub-receiver."""
    fI = np.ascontiguousarray(filtI, dtype=np.float64)
    fQ = np.ascontiguousarray(filtQ, dtype=np.float64)
    check(load().qh_quisk_set_filters2(fI.ctypes.data, fQ.ctypes.data, fI.size, int(bandwidth)))


def set_key_state(key_down, cw_key_down=0, active_sidetone=0, is_fdx=0):
    load().qh_quisk_set_key_state(int(key_down), int(cw_key_down), int(active_sidetone), int(is_fdx))


def set_sidetone(volume, rit_freq, playback_rate=48000, txrx_silence_msec=50):
    load().qh_quisk_set_sidetone(C.c_double(volume), int(rit_freq), int(playback_rate), int(txrx_silence_msec))


def set_kill_audio(kill):
    load().qh_quisk_set_kill_audio(int(kill))


def invert_spectrum(invert):
    load().qh_quisk_invert_spectrum(int(invert))


def set_agc(level):
    load().qh_quisk_set_agc(C.c_double(level))


def set_auto_notch(on, rit_freq=0):
    """QS.set_auto_notch(on), quisk.c:4596.  (rit_freq is ignored: the RIT is set_sidetone's, as in the reference.)"""
    load().qh_quisk_set_auto_notch(int(on), 0)


def set_noise_blanker(level):
    load().qh_quisk_set_noise_blanker(int(level))


def get_filter_rate():
    return load().qh_quisk_get_filter_rate()


def process_samples(buf, n):
    """In place on a complex128 buffer with room for the output; returns the output count."""
    return load().qh_quisk_process_samples(buf.ctypes.data, int(n))


def get_graph(zoom=1.0, deltaf=0.0):
    pix = np.empty(_keep["data_width"], dtype=np.float64)
    sm = C.c_double(0)
    cnt = load().qh_quisk_get_graph(C.c_double(zoom), C.c_double(deltaf), pix.ctypes.data, C.byref(sm))
    return None if cnt == 0 else (pix, sm.value, cnt)


def get_filter():
    """QS.get_filter() (quisk.c:5481): the Rx filter's response in dB, data_width values."""
    from .lib import QuiskHipError
    L = load()
    w = _keep.get("data_width", 0)
    out = np.zeros(max(w, 1), dtype=np.float64)
    L.qh_quisk_get_filter.argtypes = [C.c_void_p]
    n = L.qh_quisk_get_filter(out.ctypes.data)
    if n <= 0:
        raise QuiskHipError(L.qh_last_error().decode(errors="replace"))
    return out[:n]


def set_squelch(level):
    load().qh_quisk_set_squelch(C.c_double(level))


def set_ssb_squelch(enabled, level):
    load().qh_quisk_set_ssb_squelch(int(enabled), int(level))


def squelch_flags():
    return load().qh_quisk_squelch_flags()


def add_tone(freq):
    load().qh_quisk_add_tone(int(freq))


def measure_frequency(mode):
    return load().qh_quisk_measure_frequency(int(mode))


def set_multirx_count(n):
    load().qh_quisk_set_multirx_count(int(n))


def set_sub_rx1_output(on):
    load().qh_quisk_set_sub_rx1_output(int(on))


def sub_rx1_audio():
    n = load().qh_quisk_sub_rx1_audio(None, 0)
    out = np.zeros(max(n, 1), dtype=np.complex128)
    load().qh_quisk_sub_rx1_audio(out.ctypes.data, n)
    return out[:n]


def process(x):
    """process_samples on a copy with room for the playback-rate output; returns the output block.  With WDSP in the audio path a call may
    hand back a block more than it was given (the shim's ring held one that was waiting, quisk_wdsp.c:52-63), interpolated like the rest:
    Quisk's own buffer is SAMP_BUFFER_SIZE = 66000 samples whatever the block; here the room follows the block, the shim's block size
    and the interpolation ratio.  (Round 6: sized for the block alone, short blocks at 96 ksps playback behind a 512-sample WDSP block ran
    over the array -- heap corruption in the caller, found by a seed sweep of the api_wdsp walks.)"""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    ratio = _keep.get("ratio", 1)
    wdsp_block = max(0, int(load().qh_wdsp_shim_in_size(0)))
    buf = np.zeros((max(x.size, 16) + wdsp_block + 16) * (ratio + 1), dtype=np.complex128)
    buf[:x.size] = x
    n = process_samples(buf, x.size)
    return buf[:max(n, 0)].copy()
