"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the integer sample unpacking in front of the receive path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.  Each function follows the
reference loop it cites byte for byte (the reference builds the int32 with memcpy / shifts; numpy does the same
with views), so the results are exact integers times one double multiply.  PARITY UNPINNED: quisk.c cannot be
built here (needs <fftw3.h>, quisk.c:6) and the reference ships no test vectors for these loops; the known
answers in tests/test_oracle_ingest.py are hand-computed from the code.
"""
import numpy as np


def _left_justified_le(parts, sample_bytes):
    """memcpy(ptxr + (4 - sample_bytes), buf + index, sample_bytes) into a zeroed little-endian int (quisk.c:3380-3381)."""
    w = np.zeros(parts.shape[:-1] + (4,), dtype=np.uint8)
    w[..., 4 - sample_bytes:] = parts
    return w.view("<i4")[..., 0]


def read_rx_udp_le(buf, sample_bytes=3, gain=1.0):
    """quisk_read_rx_udp, little-endian host branch, quisk.c:3378-3392: I then Q, `sample_bytes` each;
    samp = (xr + xi * I) * rx_udp_gain_correct."""
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    n = b.size // (2 * sample_bytes)
    parts = b[:n * 2 * sample_bytes].reshape(n, 2, sample_bytes)
    v = _left_justified_le(parts, sample_bytes).astype(np.float64)
    return (v[:, 0] + 1j * v[:, 1]) * gain


def add_rx_samples(buf, sample_bytes, big_endian):
    """add_rx_samples, quisk.c:2923-2952: little-endian bytes land in the top of the int (pt_ii = &ii + 4 - bytes);
    big-endian bytes are stored from byte 3 downwards (*pt_ii-- = *buf++)."""
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    n = b.size // (2 * sample_bytes)
    parts = b[:n * 2 * sample_bytes].reshape(n, 2, sample_bytes)
    if big_endian:
        parts = parts[..., ::-1]            # first wire byte -> byte 3, next -> byte 2, ...
    v = _left_justified_le(parts, sample_bytes).astype(np.float64)
    return v[:, 0] + 1j * v[:, 1]


def hermes_frames(buf, nrx=1):
    """read_rx_udp10, quisk.c:3745-3760, on a sequence of 512-byte frames: after 3 sync + 5 control bytes,
    504 / (6 nrx + 2) records; per receiver xi = b0<<24 | b1<<16 | b2<<8 then xr likewise, sample = xr + xi * I;
    2 microphone bytes close the record.  Returns [nrx, n]."""
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    nframes = b.size // 512
    rec = 6 * nrx + 2
    per = 504 // rec
    out = np.zeros((nrx, nframes * per), dtype=np.complex128)
    for f in range(nframes):
        base = f * 512 + 8
        for r in range(per):
            for j in range(nrx):
                p = base + r * rec + 6 * j
                t = b[p:p + 6].astype(np.int64)
                xi = np.int64((t[0] << 24 | t[1] << 16 | t[2] << 8) & 0xffffffff).astype(np.int64)
                xr = np.int64((t[3] << 24 | t[4] << 16 | t[5] << 8) & 0xffffffff).astype(np.int64)
                xi = xi - (1 << 32) if xi >= (1 << 31) else xi      # the C ints are 32 bits wide
                xr = xr - (1 << 32) if xr >= (1 << 31) else xr
                out[j, f * per + r] = float(xr) + 1j * float(xi)
    return out


def read_rx_udp17(buf, packet_bytes=1442, gain=1.0, invert_spectrum=False, dc=0j):
    """read_rx_udp17's sample loop, quisk.c:3917-3996, little-endian host branch, packet after packet: 2 header bytes, then
    records of I, Q (3 bytes each, memcpy to bytes 1..3 of a zeroed int).  `xr & 0x100` (the LSB of I) selects channel 1, where
    the sample is conjugated when the spectrum is inverted, added to dc_sum and has dc_average removed; `!(xi & 0x100)` there marks
    the start of the first block.  Returns (channel 0, channel 1, mark slots, overrange packets, dc_sum)."""
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    ch0, ch1, marks = [], [], []
    over, dc_sum = 0, 0j
    for p in range(b.size // packet_bytes):
        pk = b[p * packet_bytes:(p + 1) * packet_bytes]
        if pk[1] & 0x02:
            over += 1
        index = 2
        while index < packet_bytes:
            xr = int(_left_justified_le(pk[index:index + 3][None, :], 3)[0]); index += 3
            xi = int(_left_justified_le(pk[index:index + 3][None, :], 3)[0]); index += 3
            sample = complex(float(xr), float(xi)) * gain
            if xr & 0x100:
                if invert_spectrum:
                    sample = sample.conjugate()
                dc_sum += sample
                if not (xi & 0x100):
                    marks.append(len(ch1))
                ch1.append(sample - dc)
            else:
                ch0.append(sample)
    return np.array(ch0, dtype=np.complex128), np.array(ch1, dtype=np.complex128), np.array(marks, dtype=np.int32), over, dc_sum
