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
