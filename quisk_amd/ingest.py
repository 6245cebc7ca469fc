"""Wire-format IQ samples (include/quiskhip.h group 7): the byte layouts of Quisk's sample sources, read on the
GPU as they arrived.  quisk_read_rx_udp (quisk.c:3378-3392), add_rx_samples (quisk.c:2923-2952) and the Hermes
frames of read_rx_udp10 (quisk.c:3745-3760)."""
import ctypes as C

import numpy as np

from .lib import load, check

F64, F32 = 0, 1


class IqFormat(C.Structure):        # include/quiskhip.h: qh_iq_format
    _fields_ = [("sample_bytes", C.c_int), ("big_endian", C.c_int), ("q_first", C.c_int), ("records_per_frame", C.c_int),
                ("first_offset", C.c_longlong), ("record_stride", C.c_longlong), ("frame_stride", C.c_longlong),
                ("gain", C.c_double)]

    @classmethod
    def le24(cls, gain=1.0):
        f = cls()
        load().qh_iq_format_le24(C.byref(f), C.c_double(gain))
        return f

    @classmethod
    def hermes(cls, nrx=1, gain=1.0):
        f = cls()
        load().qh_iq_format_hermes(C.byref(f), nrx, C.c_double(gain))
        return f

    @classmethod
    def plain(cls, sample_bytes, big_endian, gain=1.0):
        """add_rx_samples(py_sample_rx_bytes, py_sample_rx_endian): I then Q, back to back."""
        return cls(sample_bytes, 1 if big_endian else 0, 0, 0, 0, 2 * sample_bytes, 0, gain)


def unpack_ptr(d_src, src_bytes, fmt, nch, chan_stride, n, d_dst, dst_stride, dtype=F64, device=0, stream=None):
    check(load().qh_unpack_iq(device, stream, d_src, src_bytes, C.byref(fmt), nch, chan_stride, n, d_dst, dst_stride, dtype))


def unpack_host(buf, fmt, nch, chan_stride, n, dtype=F64, device=0):
    """buf: bytes-like; returns complex [nch, n]."""
    raw = np.frombuffer(bytes(buf), dtype=np.uint8)
    out = np.empty((nch, max(n, 1)), dtype=np.complex128 if dtype == F64 else np.complex64)
    check(load().qh_unpack_iq_host(device, raw.ctypes.data, raw.size, C.byref(fmt), nch, chan_stride, n, out.ctypes.data,
                                   out.shape[1], dtype))
    return out[:, :n]


def unpack_udp17_host(buf, packet_bytes=1442, gain=1.0, invert_spectrum=False, dc=0j, device=0):
    """read_rx_udp17's sample loop (quisk.c:3917-3996) on whole packets: returns (channel 0 samples, channel 1 samples with `dc`
    removed, slots of channel 1 where a scan's first block starts, packets with the overrange bit, sum of the raw channel 1
    samples)."""
    raw = np.frombuffer(bytes(buf), dtype=np.uint8)
    npk = raw.size // packet_bytes
    nrec = npk * ((packet_bytes - 2) // 6)
    ch0 = np.empty(max(nrec, 1), dtype=np.complex128)
    ch1 = np.empty(max(nrec, 1), dtype=np.complex128)
    marks = np.empty(max(nrec, 1), dtype=np.int32)
    counts = np.zeros(4, dtype=np.int64)
    dcs = np.zeros(2, dtype=np.float64)
    check(load().qh_unpack_udp17_host(device, raw.ctypes.data, npk, packet_bytes, float(gain), 1 if invert_spectrum else 0, float(dc.real),
                                      float(dc.imag), ch0.ctypes.data, ch1.ctypes.data, marks.ctypes.data, counts.ctypes.data, dcs.ctypes.data))
    return ch0[:counts[0]].copy(), ch1[:counts[1]].copy(), marks[:counts[2]].copy(), int(counts[3]), complex(dcs[0], dcs[1])
